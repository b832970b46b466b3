#!/bin/bash
# headline step time for several library builds: bash tools/ab_libs.sh [segments] lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT" || exit 1
M=$1; shift
for L in "$@"; do
  for rep in 1 2; do
  T=$(MSNAP_LIB_PATH=$GRAFT_REPO_ROOT/$L python3 bench.py --segments $M --no-cpu-baseline --no-saturated --no-shared-grid --steps 1000 --warmup 100 | python3 -c "import json,sys; print('%.3f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  echo "M=$M $L $T us"
  done
done
