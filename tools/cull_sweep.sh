#!/bin/bash
# A/B of the exact broad phase of the pairwise pass: the configs[2] fixture pipeline per evaluator (MODES), then
# dense (nothing can be culled) and sparse synthetic swarms with and without the broad phase.  GPU box only; output
# under gpurun_out/.
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/cull_sweep.log
: > $out
for mode in ${MODES:-0 1 2}; do
  echo "== fixture, MSNAP_COLLIDE_CULL_MODE=$mode (0 chosen per pass, 1 shares, 2 group pairs)" >> $out
  MSNAP_COLLIDE_CULL_MODE=$mode python tools/formation_pipeline.py --config 2 --reps 30 2>&1 | grep -oE '"us_per_pipeline": [0-9.]+, "stage_us": \{[^}]*\}' >> $out
done
echo "== fixture, no cull" >> $out
MSNAP_COLLIDE_NO_CULL=1 python tools/formation_pipeline.py --config 2 --reps 30 2>&1 | grep -oE '"us_per_pipeline": [0-9.]+, "stage_us": \{[^}]*\}' >> $out
for n in ${NS:-1024 2048 4096 8192}; do
  for spread in "" 400; do
    for mode in ${MODES:-0 1 2}; do
      echo "== N=$n SPREAD=$spread MODE=$mode" >> $out
      SPREAD=$spread MSNAP_COLLIDE_CULL_MODE=$mode WPC=0 python tools/collide_tune.py $n 91 2>&1 | grep '^{' | tail -1 >> $out
    done
    echo "== N=$n SPREAD=$spread NO_CULL" >> $out
    SPREAD=$spread MSNAP_COLLIDE_NO_CULL=1 WPC=0 python tools/collide_tune.py $n 91 2>&1 | grep '^{' | tail -1 >> $out
  done
done
