#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for K in 20 200 2000 20000; do for rep in 1 2; do
  python3 bench.py --steps $K --warmup 5 --no-cpu-baseline --no-saturated --no-shared-grid | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $K: %.3f us/step' % (d['ms_per_step']*1e3))"
done; done
