#!/bin/bash
# GPU-box loop for the small-batch kernel: parity tests that exercise it, phase timeline, headline bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python3 -m pytest tests/test_solve_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
MSNAP_LIB_PATH=$GRAFT_REPO_ROOT/tools/libmsnap_tl.so timeout -k 10 120 python3 tools/twist_timeline.py || exit 1
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-saturated --no-shared-grid | cut -c1-330
