#!/bin/bash
# scalar data cache passes on tools/gpu_perf.py (camera-ray kernel: its wave-uniform steps fetch node records through it)
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-s}
SPP=${SPP:-256}
cd /tmp
i=0
for grp in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" "SQC_DCACHE_BUSY_CYCLES SQC_TC_DATA_READ_REQ SQC_TC_STALL SQ_BUSY_CYCLES" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcs_${TAG}_$i -- python3 $R/tools/gpu_perf.py sponza260k 1920 1080 $SPP es0s0 1 > $R/gpurun_out/pmcs_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmcs_${TAG}_$i.log; }
  echo "pass $i done"
done
