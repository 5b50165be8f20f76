#!/bin/bash
# Bounce-kernel time against resident blocks per CU (A/B library: VMX_AB_BOUNCE_BLOCKS caps k_trace_w<1>'s blocks):
# how much of the kernel's time is latency hiding (profiles/r04_state_pool.txt).
for b in "$@"; do
  echo "== bounce blocks per CU <= $b"
  VMX_LIB=build/libvermilion_hip_ab.so VMX_AB_BOUNCE_BLOCKS=$b python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step'])"
done
