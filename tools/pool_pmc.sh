#!/bin/bash
# SQ / LDS counter passes of the phase-pure probe (k_trace_pool, A/B library) and of k_trace_w<1> on the same rays:
# what profiles/r04_state_pool.txt quotes.  One counter group per rocprofv3 run (never with trace domains other than
# --kernel-trace).  TAG=r04pool [VMX_AB_POOL_SLOTS=.. VMX_AB_POOL_LEVELS=..] bash tools/pool_pmc.sh
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r04pool}
export VMX_LIB=$R/build/libvermilion_hip_ab.so
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/tools/pool_probe.py 256 > $R/gpurun_out/pmc_${TAG}_$i.txt 2> $R/gpurun_out/pmc_${TAG}_$i.err || { echo "pmc pass $i failed"; tail -3 $R/gpurun_out/pmc_${TAG}_$i.err; }
  echo "pass $i done: $grp"
done
python3 $R/tools/pool_pmc_show.py $TAG
