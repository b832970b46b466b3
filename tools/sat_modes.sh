#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for mode in "" "--no-graph"; do for rep in 1 2; do
  python3 bench.py --drones 1048576 $mode --no-cpu-baseline --no-saturated --no-shared-grid --steps 20 --warmup 3 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('main path $mode: %.1f us/step' % (d['ms_per_step']*1e3))"
done; done
python3 tools/sat_probe.py probe
python3 bench.py --no-cpu-baseline --no-shared-grid | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('saturated leg: %.1f us' % (d['saturated']['ms_per_launch']*1e3))"
