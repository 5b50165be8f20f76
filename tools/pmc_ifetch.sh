#!/bin/bash
# instruction-fetch / scalar-cache counter passes on tools/gpu_perf.py, output gpurun_out/pmci_<TAG>_N
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-a}
SPP=${SPP:-64}
cd /tmp
i=0
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmci_${TAG}_$i -- python3 $R/tools/gpu_perf.py sponza260k 1920 1080 $SPP es0s0 1 > $R/gpurun_out/pmci_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmci_${TAG}_$i.log; }
  echo "pass $i done"
done
