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
