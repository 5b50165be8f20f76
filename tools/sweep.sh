#!/bin/bash
R=$GRAFT_REPO_ROOT
for mp in 134217728 268435456 600000000; do
  echo "== MAXP=$mp"
  MAXP=$mp timeout -k 10 200 python $R/tools/gpu_perf.py sponza260k 1920 1080 256 es0s0 2 2>&1 | grep -E "rep1|rror" | cut -c1-250
done
