#!/bin/bash
# small-batch step time over segment counts, committed library (tools/libmsnap_old.so) vs working tree
cd "$GRAFT_REPO_ROOT" || exit 1
for M in 3 4 6 8 10 12 13 16 18 20 22 24; do
  A=$(MSNAP_LIB_PATH=$GRAFT_REPO_ROOT/tools/libmsnap_old.so python3 bench.py --segments $M --no-cpu-baseline --no-saturated --no-shared-grid --steps 500 --warmup 50 | python3 -c "import json,sys; print('%.3f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  B=$(python3 bench.py --segments $M --no-cpu-baseline --no-saturated --no-shared-grid --steps 500 --warmup 50 | python3 -c "import json,sys; print('%.3f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  echo "M=$M old $A us  new $B us"
done
