#!/bin/bash
# Runs in the BUILD container after tools/make_profiles.sh <tag> and tools/make_pipeline_profile.sh <tag> came back
# through gpurun_out/: copies the summaries that should be judged into profiles/ (tracked), named per round.
#   bash tools/collect_profiles.sh r03
set -e
TAG=${1:-r03}
cd "$(dirname "$0")/.."
P=gpurun_out/profiles_${TAG}; Q=gpurun_out/pipeline_${TAG}
cp $P/kernel_stats.csv profiles/${TAG}_kernel_stats.csv
cp $P/kernel_summary.md profiles/${TAG}_kernel_summary.md
cp $P/bench.json profiles/${TAG}_bench.json
cp $P/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
cp $P/pmc_summary.md profiles/${TAG}_pmc_summary.md
cp $P/pmc_traffic.json profiles/pmc_traffic.json
cp $Q/pipeline_summary.md profiles/${TAG}_pipeline_summary.md
cp $Q/config2.json profiles/${TAG}_config2_pipeline.json
cp $Q/config3.json profiles/${TAG}_config3_pipeline.json
[ -f $Q/config2_all_pairs.json ] && cp $Q/config2_all_pairs.json profiles/${TAG}_config2_all_pairs_pipeline.json
cp $Q/pmc_counters.json profiles/pmc_counters.json
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
sha = json.load(open("profiles/pmc_counters.json")).get("csrc_sha")
tr = json.load(open("profiles/pmc_traffic.json"))
print("pmc_counters csrc_sha", sha, "| pmc_traffic", {k: v.get("csrc_sha") for k, v in tr.items()})
sys.path.insert(0, ".")
from drone_path_planning_python_amd._lib import csrc_sha
print("this tree's csrc_sha", csrc_sha(), "(bench.py quotes the counters only when they match)")
PY

# evidence of the separate runs (final bench line with the counters of this build, sizes, parts, PMC of the order-9 kernel,
# the 2-rank rehearsal), when present under gpurun_out/
[ -f gpurun_out/${TAG}_bench_final.json ] && cp gpurun_out/${TAG}_bench_final.json profiles/${TAG}_bench.json
if [ -f gpurun_out/${TAG}_order_sizes.txt ]; then
  { echo "# tools/order_sizes.py: the solve over batch sizes, each kernel family under a hipGraph of 50 launches, one MI355X."
    echo "# Per size three lines: the launcher's choice, with no_twist (small batches off the two-sided latency kernel), with no_twist + no_twin"
    echo "# (solve_kernel_reg).  Order 9 (configs[4] = 65536 x 10; sharded over 8 GPUs: 8192 per rank), then order 7, then odd / longer paths."
    cat gpurun_out/${TAG}_order_sizes.txt; } > profiles/${TAG}_order_sizes.txt
fi
if [ -f gpurun_out/${TAG}_cull_sweep.txt ]; then
  { echo "# tools/cull_sweep.sh: the pairwise pass behind its exact broad phase.  Fixture = the configs[2] pipeline (us per pipeline, stage"
    echo "# times); MODE 0 = evaluator chosen per pass on the device, 1 = surviving 128 x 8 shares, 2 = surviving 8 x 8 group pairs; then"
    echo "# tools/collide_tune.py on dense (every sample anywhere in a 100 m cube: nothing can be culled) and sparse (SPREAD=400: drones 3 m"
    echo "# apart over 400 x 400 m) synthetic swarms, and with the broad phase off (NO_CULL)."
    cat gpurun_out/${TAG}_cull_sweep.txt; } > profiles/${TAG}_cull_sweep.txt
fi
if [ -f gpurun_out/${TAG}_collide_sizes.jsonl ]; then
  { echo "# tools/collide_tune.py <N> 91 (WPC=0: the launcher's own geometry; 25 warm + 20 timed launches): transposition + span + merge of one"
    echo "# msnap_formation_collide over the whole swarm, one MI355X; then tools/collide_parts.py: the pass in parts (every unordered pair on"
    echo "# exactly one rank), every rank of P = 2, 4, 8 emulated on the one GPU (part_us: msnap_formation_collide_part of that rank, eager"
    echo "# launches timed with events -- three kernels per call, so the small parts include ~10 us of launch overhead; kernel times under"
    echo "# rocprofv3 are in ${TAG}_collide_parts_kernels.md), folded result compared with the single launch bit for bit"
    cat gpurun_out/${TAG}_collide_sizes.jsonl; } > profiles/${TAG}_collide_sizes.txt
fi
[ -f gpurun_out/${TAG}_collide_parts_kernels.md ] && cp gpurun_out/${TAG}_collide_parts_kernels.md profiles/${TAG}_collide_parts_kernels.md
if [ -f gpurun_out/${TAG}_twin_pmc.txt ]; then
  { echo "# bash tools/pmc_kernel.sh solve_kernel_twin \"tools/order9_once.py 65536 10 0 20\" <counter sets>: solve_kernel_twin<5, 10> at configs[4]"
    echo "# (means per launch; SQ_* cycle counters in units of 4 cycles summed over waves / SIMDs, GRBM_GUI_ACTIVE in cycles summed over 8 XCDs)"
    grep -v "amdgpu\|^$" gpurun_out/${TAG}_twin_pmc.txt; } > profiles/${TAG}_twin_pmc.txt
fi
if [ -f gpurun_out/${TAG}_bench_n2.json ]; then
  grep "^{" gpurun_out/${TAG}_bench_n2.json | tail -1 > profiles/${TAG}_bench_n2_gloo_rehearsal.json
fi
ls profiles | grep "^${TAG}_"
