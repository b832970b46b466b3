#!/bin/bash
# Runs ON THE GPU BOX after tools/collect_profiles.sh has put the counter files of this build under profiles/: the
# bench line that quotes them, the solve over batch sizes, the pairwise pass over sizes and in parts (+ kernel trace),
# the PMC passes of the order-9 throughput kernel and the 2-rank gloo rehearsal -> gpurun_out/<tag>_*
#   bash tools/final_evidence.sh r03
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python3 bench.py > gpurun_out/${TAG}_bench_final.json 2> gpurun_out/${TAG}_bench_final.err || { tail -3 gpurun_out/${TAG}_bench_final.err; exit 1; }
(python3 tools/order_sizes.py 10 2048 4096 8192 16384 32768 65536; PROBE_ORDER=7 python3 tools/order_sizes.py 10 2048 4096 8192 16384 32768 65536
 PROBE_ORDER=7 python3 tools/order_sizes.py 9 4096 16384; python3 tools/order_sizes.py 9 8192 65536
 PROBE_ORDER=7 python3 tools/order_sizes.py 20 4096 16384 65536; python3 tools/order_sizes.py 12 8192 65536) 2>&1 | grep "^order" > gpurun_out/${TAG}_order_sizes.txt
(for n in 256 512 1024 2048 3072 4096 6144 8192; do WPC=0 python3 tools/collide_tune.py $n 91; done
 WPC=0 python3 tools/collide_tune.py 8192 96; WPC=0 python3 tools/collide_tune.py 4096 96
 python3 tools/collide_parts.py 4096 91 2 4 8; python3 tools/collide_parts.py 16384 91 8) 2>&1 | grep "^{" > gpurun_out/${TAG}_collide_sizes.jsonl
NS="2048 4096 8192" bash tools/cull_sweep.sh && cp gpurun_out/cull_sweep.log gpurun_out/${TAG}_cull_sweep.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_parts -- python3 tools/collide_parts.py 4096 91 2 4 8 > /dev/null 2>&1
python3 tools/prof_summary.py gpurun_out/prof_${TAG}_parts gpurun_out/${TAG}_collide_parts_kernels.md > /dev/null
rm -rf gpurun_out/prof_${TAG}_parts
bash tools/pmc_kernel.sh solve_kernel_twin "tools/order9_once.py 65536 10 0 20" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
  "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
  > gpurun_out/${TAG}_twin_pmc.txt 2>&1
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 \
  --steps 200 --warmup 20 --backend gloo --no-cpu-baseline --no-saturated --no-shared-grid --no-end-to-end \
  > gpurun_out/${TAG}_bench_n2.json 2> gpurun_out/${TAG}_bench_n2.err
tail -c 300 gpurun_out/${TAG}_bench_final.json; echo; echo "final evidence done"
