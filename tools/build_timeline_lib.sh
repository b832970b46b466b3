#!/bin/bash
# Builds tools/libmsnap_tl.so: the library with -DMSNAP_TOOLS_TIMELINE phase probes in the
# small-batch solve kernel, the shared-grid GEMM and the fused solve + sampler kernel
# (tools/twist_timeline.py, tools/grid_timeline.py, tools/grid_sample_timeline.py).
set -e
cd "$(dirname "$0")/../drone_path_planning_python_amd/csrc"
make -s
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-bitwise-instead-of-logical -DMSNAP_TOOLS_TIMELINE"
/opt/rocm/bin/hipcc $FLAGS -c msnap_solve.hip -o /tmp/tl_msnap_solve.o
/opt/rocm/bin/hipcc $FLAGS -c msnap_grid.hip -o /tmp/tl_msnap_grid.o
/opt/rocm/bin/hipcc $FLAGS -c msnap_aux.hip -o /tmp/tl_msnap_aux.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libmsnap_tl.so msnap_api.o /tmp/tl_msnap_solve.o /tmp/tl_msnap_aux.o /tmp/tl_msnap_grid.o
echo "built tools/libmsnap_tl.so  (run: MSNAP_LIB_PATH=\$PWD/tools/libmsnap_tl.so python3 tools/twist_timeline.py)"
