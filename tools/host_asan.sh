#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the library (device code untouched: -fsanitize only after -Xarch_host) and a
# run of the C++ examples against it on the GPU box.   /usr/local/graft/bin/gpurun -- 'bash tools/host_asan.sh'
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/vermilion_amd/csrc
mkdir -p $R/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O1 -g -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function \
  -DVMX_TRACE_WAVES_PER_SIMD=7 -DVMX_TRACE_SGPRS=80 -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
  -I$R/include -shared -o $R/build/libvermilion_hip_asan.so vmx_kernels.hip lbvh_build.hip path_compact.hip -x hip vmx_api.cpp bvh_build.cpp
cd $R
for ex in render_cornell render_multi; do
  /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -g -fsanitize=address -fsanitize=undefined -I include examples/$ex.cpp build/libvermilion_hip_asan.so \
    -Wl,-rpath,$R/build -o build/${ex}_asan
done
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
./build/render_cornell_asan && ./build/render_multi_asan
echo "host sanitizers: clean"
