#!/bin/bash
# one SQ counter pass + lane utilisation of the bench frame (BENCH_ARGS, TAG): the group make_counters.py needs for
# valu_lane_utilization / instruction counts only
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --no-extras}"
TAG=${TAG:-r03sq}
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_${TAG}_1 -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_1.json 2> $R/gpurun_out/pmc_${TAG}_1.err || { echo "pmc pass failed"; tail -3 $R/gpurun_out/pmc_${TAG}_1.err; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${TAG}_2 -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_2.json 2> $R/gpurun_out/pmc_${TAG}_2.err || { echo "pmc pass 2 failed"; tail -3 $R/gpurun_out/pmc_${TAG}_2.err; }
echo done
