#!/bin/bash
R=$GRAFT_REPO_ROOT
for mp in 16777216 33554432 67108864 134217728; do
  echo "== MAXP=$mp"
  MAXP=$mp timeout -k 10 120 python $R/tools/gpu_perf.py sponza260k 1920 1080 256 es0s0 2 2>&1 | grep rep1
done
