#!/bin/bash
# order-9 step times over batch sizes (BASELINE.json configs[4] is 65536 drones, 8192 per GPU on 8)
cd "$GRAFT_REPO_ROOT" || exit 1
for N in 256 4096 8192 65536 524288; do
  python3 bench.py --order 9 --drones $N --no-cpu-baseline --no-saturated --no-shared-grid --steps 200 --warmup 20 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=$N order 9: %.2f us/step  %.3e traj/s  roofline frac %.3f  kernel %s' % (d['ms_per_step']*1e3, d['value'], d['roofline']['frac'], d['roofline']['kernel']))"
done
