#!/bin/bash
# order-9 solve: small-batch two-sided kernel vs the register-resident persistent kernel over batch sizes
for T in 0 1000000; do
  for N in 8192 16384 32768 65536 131072; do
    MSNAP_TWIST_MAX_DRONES=$T python3 bench.py --order 9 --drones $N --steps 60 --warmup 20 --no-graph --no-cpu-baseline \
      --no-saturated --no-shared-grid --no-configs --no-end-to-end 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('twist_max', $T, 'N', $N, r['kernel'], round(r['avg_launch_us'],1), 'us', round(r['frac'],3))"
  done
done
