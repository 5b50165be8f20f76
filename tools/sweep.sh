#!/bin/bash
R=$GRAFT_REPO_ROOT
for cfg in "10" "8" "7"; do
  echo "== LDSE=$cfg"
  LDSE=$cfg timeout -k 10 200 python $R/tools/gpu_perf.py sponza260k 1920 1080 256 es0s0,es1s0 2 2>&1 | grep -E "rep1|rror" | cut -c1-250
done
