#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the configs[2] / configs[3] pipelines on the pinned
# formation fixtures (a8 -> solve -> sample -> pairwise pass -> sweep against the reference's STL scene)
# and of the streaming kernels -> gpurun_out/pipeline_<tag>/
set -o pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pipeline_${TAG}
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -- python3 tools/formation_pipeline.py --config 2 --reps 30 > $OUT/config2.json 2> $OUT/c2.err || { tail -3 $OUT/c2.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 tools/formation_pipeline.py --config 3 --reps 30 > $OUT/config3.json 2> $OUT/c3.err || { tail -3 $OUT/c3.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/aux -- python3 tools/aux_bench.py > $OUT/aux_bench.txt 2> $OUT/aux.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p49 -- python3 tools/node_path49.py > $OUT/node_path49.txt 2> $OUT/p49.err || { tail -3 $OUT/p49.err; exit 1; }
# vector-instruction counts of the two collision kernels on the same workload (bench.py's mesh roofline)
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc3 -- python3 tools/formation_pipeline.py --config 3 --reps 10 > /dev/null 2> $OUT/pmc3.err || exit 1
# the all-pairs kernel of the pairwise pass on the same workload (broad phase off: dense swarms, shards, parts)
export MSNAP_COLLIDE_NO_CULL=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2all -- python3 tools/formation_pipeline.py --config 2 --reps 30 > $OUT/config2_all_pairs.json 2> $OUT/c2all.err || { tail -3 $OUT/c2all.err; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc3all -- python3 tools/formation_pipeline.py --config 3 --reps 10 > /dev/null 2> $OUT/pmc3all.err || exit 1
unset MSNAP_COLLIDE_NO_CULL
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.getcwd())
from drone_path_planning_python_amd._lib import csrc_sha
out = sys.argv[1]
with open(out + "/pipeline_summary.md", "w") as f:
    for tag, title in (("c2", "configs[2]: 4096 drones x 10 segments (formation fixture), 91 samples, pairwise pass"),
                       ("c3", "configs[3]: 4096 drones x 20 segments (formation fixture), 96 samples, pairwise pass + 68-triangle STL scene"),
                       ("c2all", "configs[2] with the pairwise pass's broad phase off (MSNAP_COLLIDE_NO_CULL=1): the all-pairs kernel"),
                       ("aux", "streaming kernels at 2^18 drones x 10 segments (tools/aux_bench.py)"),
                       ("p49", "the reference's live shape: 50-pose rigid-body path -> transform -> paths_to_pols, 2 drones x 49 segments (tools/node_path49.py)")):
        path = glob.glob(f"{out}/{tag}/**/*kernel_stats.csv", recursive=True)[0]
        f.write(f"### {title}\n\n| kernel | calls | mean us | min us | max us |\n|---|---|---|---|---|\n")
        for r in csv.DictReader(open(path)):
            if "msnap::" not in r["Name"]:
                continue
            name = r["Name"].split("(")[0].replace("void ", "")
            f.write("| `%s` | %s | %.1f | %.1f | %.1f |\n" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                           float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
        f.write("\n")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f"{out}/pmc3/**/*counter_collection.csv", recursive=True) + glob.glob(f"{out}/pmc3all/**/*counter_collection.csv", recursive=True):
        allp = "/pmc3all/" in fn
        for r in csv.DictReader(open(fn)):
            for k in (("collide_span_kernel",) if allp else ("mesh_sweep_kernel", "collide_eval_groups_kernel", "collide_gather_kernel", "sample_kernel", "grid_sample_kernel")):
                if "::" + k in r["Kernel_Name"]:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
    counters["csrc_sha"] = csrc_sha()      # bench.py quotes these counters only for the sources they were taken from
    json.dump(counters, open(out + "/pmc_counters.json", "w"), indent=1)
    f.write("### PMC counters per launch, configs[3] workload (mean)\n\n| kernel | counter | value |\n|---|---|---|\n")
    for k, cs in sorted(counters.items()):
        if not isinstance(cs, dict):
            continue
        for c, v in sorted(cs.items()):
            f.write(f"| `{k}` | {c} | {v:.6g} |\n")
PY
cat $OUT/pipeline_summary.md
rm -rf $OUT/c2 $OUT/c3 $OUT/c2all $OUT/aux $OUT/pmc3 $OUT/pmc3all $OUT/p49
