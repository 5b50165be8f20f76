#!/bin/bash
# texture-addresser / L1 (TCP) / texture-data utilisation passes (two counters per pass: larger TA/TCP groups are
# refused by the hardware ("exceeds the capabilities") and the aborted profiler then sits until the timeout) on tools/gpu_perf.py, output gpurun_out/pmct_<TAG>_N
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-a}
SPP=${SPP:-64}
cd /tmp
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_ACCESSES_sum TCP_TAGRAM0_REQ_sum"; do
  i=$((i+1))
  timeout -k 5 60 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmct_${TAG}_$i -- python3 $R/tools/gpu_perf.py sponza260k 1920 1080 $SPP es0s0 1 > $R/gpurun_out/pmct_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmct_${TAG}_$i.log; }
  echo "pass $i done"
done
