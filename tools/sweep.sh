#!/bin/bash
R=$GRAFT_REPO_ROOT
for h in 1048576 4194304 16777216 67108864; do
  echo "== HYBRID=$h"
  VMX_HYBRID=$h timeout -k 10 120 python $R/tools/gpu_perf.py sponza260k 1920 1080 256 es1s0 2 2>&1 | grep rep1 | cut -c1-230
done
