#!/bin/bash
R=$GRAFT_REPO_ROOT
for cfg in "1048576 16" "4194304 32" "4194304 64" "16777216 64" "16777216 128"; do
  set -- $cfg
  echo "== SPEC=$1 CAP=$2"
  VMX_SPEC=$1 VMX_SPEC_CAP=$2 timeout -k 10 120 python $R/tools/gpu_perf.py sponza260k 1920 1080 256 es1s0 2 2>&1 | grep rep1 | cut -c1-250
done
