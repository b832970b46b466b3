#!/bin/bash
# One GPU iteration: parity tests, then the bench under rocprofv3 --kernel-trace.
# usage (on the GPU box via gpurun): bash tools/gpu_iter.sh <tag> [extra bench args]
set -o pipefail
TAG=${1:-iter}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python3 -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; prc=$?
tail -3 gpurun_out/${TAG}_pytest.log
[ $prc -ne 0 ] && exit $prc
rm -rf gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --no-graph --steps 300 --warmup 30 --no-cpu-baseline "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python3 tools/prof_summary.py gpurun_out/prof_${TAG} gpurun_out/${TAG}_prof.md | cut -c1-220 | head -8
python3 - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print("headline: %.3g traj/s  %.2f us/step  roofline frac %.4f" % (d["value"], d["ms_per_step"]*1e3, d["roofline"]["frac"]))
s=d.get("saturated")
if s: print("saturated: %.4g traj/s  %.3f ms/launch  %.1f GB/s  frac %.3f" % (s["value"], s["ms_per_launch"], s["roofline"]["achieved"], s["roofline"]["frac"]))
PY
