#!/bin/bash
# PMC passes for the bench command (run on the GPU box through gpurun): one counter group per rocprofv3 run
# (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2 — MI355X_MICROARCH.md; TA/TCP groups larger than two
# counters are refused by the hardware), --kernel-trace only beside --pmc.  Then tools/make_counters.py <TAG>.
# The last run is the plain kernel trace (--stats) of the same command: profiles/<TAG>_kernel_stats.csv.
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --no-extras}"
TAG=${TAG:-r02}
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_TA_BUSY_sum TA_BUSY_avr" "TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_$i.json 2> $R/gpurun_out/pmc_${TAG}_$i.err || { echo "pmc pass $i ($grp) failed"; tail -3 $R/gpurun_out/pmc_${TAG}_$i.err; }
  echo "pass $i done: $grp"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_${TAG} -- python3 $R/bench.py $ARGS > $R/gpurun_out/trace_${TAG}.json 2> $R/gpurun_out/trace_${TAG}.err || { echo "kernel trace failed"; tail -3 $R/gpurun_out/trace_${TAG}.err; }
echo "kernel trace done"
