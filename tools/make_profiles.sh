#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): collects the rocprofv3 evidence for one round.
#   bash tools/make_profiles.sh r01
# Writes gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/.
set -o pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/profiles_${TAG}
rm -rf "$OUT" && mkdir -p "$OUT"

echo "== 1. kernel trace + stats of the default bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || { tail -5 $OUT/bench_under_rocprof.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 tools/prof_summary.py $OUT/trace $OUT/kernel_summary.md > /dev/null

echo "== 2. un-profiled bench line (the number to quote)"
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }

echo "== 3. HBM traffic counters, one pass each (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2), per workload"
HEAD="--no-graph --steps 100 --warmup 10 --no-cpu-baseline --no-saturated --no-shared-grid --no-configs --no-end-to-end"
SAT="--no-graph --drones 1048576 --steps 5 --warmup 2 --no-cpu-baseline --no-saturated --no-shared-grid --no-configs --no-end-to-end"
K2="--no-graph --drones 1048576 --saturated-drones 1048576 --steps 5 --warmup 2 --no-cpu-baseline --no-saturated --no-configs --no-end-to-end"
C4="--no-graph --order 9 --drones 65536 --steps 20 --warmup 5 --no-cpu-baseline --no-saturated --no-shared-grid --no-configs --no-end-to-end"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_65536x10o9_$C -- python3 bench.py $C4 > /dev/null 2> $OUT/pmc_c4_$C.err || { tail -5 $OUT/pmc_c4_$C.err; exit 1; }
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_256x10o7_$C -- python3 bench.py $HEAD > /dev/null 2> $OUT/pmc_h_$C.err || { tail -5 $OUT/pmc_h_$C.err; exit 1; }
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_1048576x10o7_$C -- python3 bench.py $SAT > /dev/null 2> $OUT/pmc_s_$C.err || { tail -5 $OUT/pmc_s_$C.err; exit 1; }
  # the shared-grid GEMM (K2) at the same size: every grid_gemm launch of this command is 2^20 drones
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_grid1048576x10o7_$C -- python3 bench.py $K2 > /dev/null 2> $OUT/pmc_g_$C.err || { tail -5 $OUT/pmc_g_$C.err; exit 1; }
done
echo "== 4. SQ counters (all kernels of the default command: solve variants + the K2 MFMA GEMM)"
ALL="--no-graph --steps 100 --warmup 10 --no-cpu-baseline --no-end-to-end"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_all_sq1 -- python3 bench.py $ALL > /dev/null 2> $OUT/pmc_sq1.err || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_all_sq2 -- python3 bench.py $ALL > /dev/null 2> $OUT/pmc_sq2.err || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr --kernel-trace --output-format csv -d $OUT/pmc_all_mem -- python3 bench.py $ALL > /dev/null 2> $OUT/pmc_mem.err || exit 1
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.md
cat $OUT/pmc_summary.md | tail -25
# keep the merged-back payload small
rm -rf $OUT/trace $OUT/pmc_*_FETCH_SIZE $OUT/pmc_*_WRITE_SIZE $OUT/pmc_all_sq1 $OUT/pmc_all_sq2 $OUT/pmc_all_mem
tail -c 1200 $OUT/bench.json
