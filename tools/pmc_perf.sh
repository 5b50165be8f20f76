#!/bin/bash
# SQ counter passes on tools/gpu_perf.py (MODES env), output gpurun_out/pmcp_<TAG>_N
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-a}
MODES=${MODES:-es0s0}
SPP=${SPP:-32}
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcp_${TAG}_$i -- python3 $R/tools/gpu_perf.py sponza260k 1920 1080 $SPP $MODES 1 > $R/gpurun_out/pmcp_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmcp_${TAG}_$i.log; }
  echo "pass $i done"
done
