#!/bin/bash
# GPU box: the broad-phase pass after a change -- parity tests, randomised check, fixture pipeline per evaluator, kernel trace.
#   bash tools/gpu_cull_check.sh <tag> [stress seconds]
set -o pipefail
TAG=${1:-x}; SEC=${2:-60}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/cull_$TAG; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_formation_full.py tests/test_aux_gpu.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
timeout -k 10 $((SEC + 120)) python tools/cull_stress.py $SEC 4 > $O/stress.log 2>&1 || { tail -20 $O/stress.log; exit 1; }
tail -1 $O/stress.log
for mode in 0 1 2; do
  echo "== fixture, mode $mode" >> $O/pipe.log
  MSNAP_COLLIDE_CULL_MODE=$mode python tools/formation_pipeline.py --config 2 --reps 40 2>&1 | grep -oE '"us_per_pipeline": [0-9.]+, "stage_us": \{[^}]*\}' >> $O/pipe.log
done
echo "== config 3" >> $O/pipe.log
python tools/formation_pipeline.py --config 3 --reps 40 2>&1 | grep -oE '"us_per_pipeline": [0-9.]+, "stage_us": \{[^}]*\}' >> $O/pipe.log
cat $O/pipe.log
rocprofv3 --kernel-trace -d $O/trace -- python3 tools/formation_pipeline.py --config 2 --reps 60 > /dev/null 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
python3 tools/kernel_medians.py $(dirname $(find $O/trace -name "*.db" | head -1)) sample_kernel > $O/kernel_medians.txt 2>&1
cat $O/kernel_medians.txt | head -40
rm -rf $O/trace
