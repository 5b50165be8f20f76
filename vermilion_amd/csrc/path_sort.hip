// path_sort.hip — reordering of the live-path id queue between two bounce generations.
//
// The ids k_shade appends are in pixel order: neighbouring entries start on neighbouring surface points
// but leave them in unrelated directions, so the 64 rays of a bounce wave part ways at the first nodes.
// Here every queued path gets a key from its next ray (origin cell in the scene's bounds, direction cell
// on an octahedral map) and each sub-queue is radix-sorted by it (hipcub, key-value pairs), so that the
// rays a wave picks up share origin region and direction.  The order in which a ray performs its own
// tests (bvh.cpp:61-133) does not depend on its neighbours: frames stay bit-identical.
//
// MEASURED AND NOT IN THE PRODUCT (profiles/r03_bounce_sort.txt): whatever the key, the bounce
// kernel's time, step mix and lane utilisation stay where they were (34.9 ms, 98 % of wave-steps hold inner-node
// and leaf lanes, 0.51 -> 0.52), and the sort costs 3-6 ms on top.  This file is compiled into the A/B library
// only (make ab); tools/sort_probe.py reproduces the measurement.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <algorithm>

#include "vmx_kernels.h"

namespace vmx {
namespace {

__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ uint32_t spread2(uint32_t v) {  // 16 bits -> every second bit
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
__device__ __forceinline__ uint32_t quant(float x, uint32_t bits) {  // x in [0,1) -> cell; NaN -> 0
    const float top = (float)((1u << bits) - 1u);
    return (uint32_t)fminf(fmaxf(x * (float)(1u << bits), 0.f), top);
}

__global__ void __launch_bounds__(256)
k_ray_keys(IdQueue q, const float4 *__restrict__ state, SortKeyCfg cfg, unsigned int *__restrict__ keys) {
    const uint32_t sub = blockIdx.y;
    const uint32_t n = min(q.counts[sub * 32], q.sub_capacity);
    for (uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x; pos < n; pos += gridDim.x * blockDim.x) {
        const size_t at = (size_t)sub * q.sub_capacity + pos;
        const uint32_t pid = q.ids[at];
        const float4 a = state[(size_t)pid * 4], b = state[(size_t)pid * 4 + 1];
        uint32_t cell = 0, dir = 0;
        if (cfg.obits) {
            cell = spread3(quant((a.x - cfg.lo[0]) * cfg.inv[0], cfg.obits)) |
                   (spread3(quant((a.y - cfg.lo[1]) * cfg.inv[1], cfg.obits)) << 1) |
                   (spread3(quant((a.z - cfg.lo[2]) * cfg.inv[2], cfg.obits)) << 2);
        }
        if (cfg.dbits) {
            // octahedral map of the direction onto [0,1)^2
            const float dx = a.w, dy = b.x, dz = b.y;
            const float s = 1.0f / (fabsf(dx) + fabsf(dy) + fabsf(dz));
            float u = dx * s, v = dy * s;
            if (dz < 0.f) {
                const float uu = (1.f - fabsf(v)) * (u >= 0.f ? 1.f : -1.f);
                const float vv = (1.f - fabsf(u)) * (v >= 0.f ? 1.f : -1.f);
                u = uu, v = vv;
            }
            dir = spread2(quant(u * 0.5f + 0.5f, cfg.dbits)) | (spread2(quant(v * 0.5f + 0.5f, cfg.dbits)) << 1);
        }
        uint32_t key;
        if (cfg.chunk_log2) key = ((pos >> cfg.chunk_log2) << (2 * cfg.dbits)) | dir;  // direction order inside chunks of the queue
        else if (cfg.dir_major) key = (dir << (3 * cfg.obits)) | cell;
        else key = (cell << (2 * cfg.dbits)) | dir;
        keys[at] = key;
    }
}

}  // namespace

size_t path_sort_tmp_bytes(uint32_t max_n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (unsigned int *)nullptr, (unsigned int *)nullptr,
                                             (unsigned int *)nullptr, (unsigned int *)nullptr, (int)max_n, 0, 32, nullptr);
    return bytes;
}

int path_sort_ids(const IdQueue &q, const uint32_t *h_counts, const void *state, const SortKeyCfg &cfg,
                  unsigned int *keys_a, unsigned int *keys_b, unsigned int *ids_out, void *tmp, size_t tmp_bytes,
                  void *stream) {
    hipStream_t s = (hipStream_t)stream;
    uint32_t largest = 0;
    for (uint32_t i = 0; i < kSubQueues; ++i) largest = h_counts[i] > largest ? h_counts[i] : largest;
    if (largest == 0) return 0;
    const uint32_t gx = std::min<uint32_t>((largest + 255) / 256, 2048);
    hipLaunchKernelGGL(k_ray_keys, dim3(gx, kSubQueues), dim3(256), 0, s, q, (const float4 *)state, cfg, keys_a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    uint32_t bits = cfg.chunk_log2 ? 32u : 3 * cfg.obits + 2 * cfg.dbits;
    if (cfg.chunk_log2) {
        uint32_t top = largest >> cfg.chunk_log2, tb = 0;
        while (top) ++tb, top >>= 1;
        bits = std::min(32u, 2 * cfg.dbits + tb);
    }
    for (uint32_t i = 0; i < kSubQueues; ++i) {
        if (!h_counts[i]) continue;
        const size_t off = (size_t)i * q.sub_capacity;
        size_t tb = tmp_bytes;
        e = hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys_a + off, keys_b + off, q.ids + off, ids_out + off,
                                               (int)h_counts[i], 0, (int)bits, s);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // namespace vmx
