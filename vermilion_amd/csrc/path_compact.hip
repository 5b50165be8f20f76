// path_compact.hip — order-preserving compaction of the camera paths that have to be traced (VMX_SAMPLING_ELIDE_DEAD).
//
// k_raygen<1> decides per path whether its radiance is provably zero (vmx_kernels.hip: step_is_dead) and leaves,
// per wave of 64 consecutive path ids, one word of live bits and its popcount.  Here: exclusive scan of the popcounts
// (hipcub), then every live path id is written to its place — neighbours stay neighbours, so a traversal wave still
// holds samples of one pixel (or of a few neighbouring ones).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "vmx_kernels.h"

namespace vmx {
namespace {

__global__ void __launch_bounds__(256)
k_live_scatter(const unsigned long long *__restrict__ live_mask, const unsigned int *__restrict__ offs, uint32_t nwords,
               unsigned int *__restrict__ ids, unsigned int *__restrict__ count) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); w < nwords; w += waves) {
        const unsigned long long bits = live_mask[w];
        const uint32_t off = offs[w];
        if ((bits >> lane) & 1ull) ids[off + (uint32_t)__popcll(bits & ((1ull << lane) - 1ull))] = w * 64u + lane;
        if (w + 1 == nwords && lane == 0) *count = off + (uint32_t)__popcll(bits);
    }
}

}  // namespace

size_t live_compact_tmp_bytes(uint32_t nwords) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (unsigned int *)nullptr, (unsigned int *)nullptr, (int)nwords, nullptr);
    return bytes;
}

int launch_live_compact(const unsigned long long *live_mask, const unsigned int *live_cnt, uint32_t nwords,
                        unsigned int *offs, unsigned int *ids, unsigned int *count, void *tmp, size_t tmp_bytes, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (nwords == 0) return (int)hipMemsetAsync(count, 0, 4, s);
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, live_cnt, offs, (int)nwords, s);
    if (e != hipSuccess) return (int)e;
    const uint32_t grid = std::min<uint32_t>((nwords + 3) / 4, 256u * 16u);
    hipLaunchKernelGGL(k_live_scatter, dim3(grid), dim3(256), 0, s, live_mask, offs, nwords, ids, count);
    return (int)hipGetLastError();
}

}  // namespace vmx
