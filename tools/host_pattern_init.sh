#!/bin/bash
# The library's host code built with -ftrivial-auto-var-init=pattern (every automatic variable starts as 0xAA...), then the
# GPU tests against it: a frame that depends on an uninitialised host variable (round 3: FrameDev::lead) fails here
# whatever the stack happened to hold.   hipcc line here, tests on the GPU box:
#   bash tools/host_pattern_init.sh && /usr/local/graft/bin/gpurun -- 'VMX_LIB=build/libvmx_pattern.so python -m pytest tests -m gpu -x -q'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/vermilion_amd/csrc
mkdir -p $R/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function \
  -DVMX_TRACE_WAVES_PER_SIMD=7 -DVMX_TRACE_SGPRS=80 -Xarch_host -ftrivial-auto-var-init=pattern -I$R/include -shared \
  -o $R/build/libvmx_pattern.so vmx_kernels.hip lbvh_build.hip path_compact.hip -x hip vmx_api.cpp bvh_build.cpp
echo "built $R/build/libvmx_pattern.so"
