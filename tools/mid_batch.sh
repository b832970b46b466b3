#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for O in 7 9; do for N in 4096 6144 8192 12288; do
  A=$(MSNAP_TWIST_MAX_DRONES=1000000 python3 bench.py --order $O --drones $N --no-cpu-baseline --no-saturated --no-shared-grid --steps 300 --warmup 30 | python3 -c "import json,sys; print('%.2f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  B=$(MSNAP_NO_TWIST=1 python3 bench.py --order $O --drones $N --no-cpu-baseline --no-saturated --no-shared-grid --steps 300 --warmup 30 | python3 -c "import json,sys; print('%.2f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  echo "order $O N=$N two-sided $A us  one-sided $B us"
done; done
