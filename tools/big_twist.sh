#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for N in 65536 1048576; do
  A=$(MSNAP_TWIST_MAX_DRONES=100000000 python3 bench.py --drones $N --no-cpu-baseline --no-saturated --no-shared-grid --steps 50 --warmup 5 | python3 -c "import json,sys; print('%.2f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  B=$(python3 bench.py --drones $N --no-cpu-baseline --no-saturated --no-shared-grid --steps 50 --warmup 5 | python3 -c "import json,sys; print('%.2f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  echo "N=$N two-sided $A us  one-sided $B us"
done
