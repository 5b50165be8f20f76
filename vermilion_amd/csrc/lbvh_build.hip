// lbvh_build.hip — BVH construction on the GPU: a linear BVH.
//
//   k_lbvh_keys       63-bit Morton code of each triangle's centroid (21 bits per axis, over the
//                     centroid bounds) — unique after the triangle index breaks ties in the sort
//   hipcub radix sort (key, triangle id) pairs
//   k_lbvh_hierarchy  one thread per internal node: its key range and split from common-prefix
//                     lengths (Karras 2012, "Maximizing parallelism in the construction of BVHs,
//                     octrees, and k-d trees"), child / parent links
//   k_lbvh_fit        bottom-up: every leaf climbs towards the root, the second thread to reach a
//                     node merges its children's boxes (exact: min/max are order-independent) and
//                     subtree height
//
// The host then walks the tree once, depth first, into the reference's flat layout (bvh.h:11-14:
// pre-order, left child = i + 1, leaf <=> rightOffset == 0), turning every subtree of at most
// `leaf_size` triangles into one leaf, and hands it to the same flattening as the other builders
// (bvh_build.cpp).  Like the SAH builder this is NOT the reference's topology: for throughput runs
// and scenes that change per frame, not for triangle-ID parity runs.  Its parity bar is the same:
// the reference's traversal run over the exported tree gives bit-identical hits (tests).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "bvh_build.h"

namespace vmx {
namespace {

#define LB_TRY(expr)                                                            \
    do {                                                                        \
        const hipError_t e_ = (expr);                                           \
        if (e_ != hipSuccess) {                                                 \
            err = std::string("LBVH builder: ") + #expr + ": " + hipGetErrorString(e_); \
            return false;                                                       \
        }                                                                       \
    } while (0)

template <typename T>
struct Buf {
    T *p = nullptr;
    ~Buf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)); }
};

__device__ __forceinline__ unsigned long long spread21(uint32_t v) {
    // 21 bits -> every third bit of 63
    unsigned long long x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void k_lbvh_keys(const float *__restrict__ pos, uint32_t n, float lox, float loy, float loz, float sx,
                            float sy, float sz, unsigned long long *__restrict__ keys, uint32_t *__restrict__ ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = pos + (size_t)i * 9;
    const float cx = ((p[0] + p[3]) + p[6]) * (1.f / 3.f), cy = ((p[1] + p[4]) + p[7]) * (1.f / 3.f),
                cz = ((p[2] + p[5]) + p[8]) * (1.f / 3.f);
    const uint32_t qx = (uint32_t)fminf(fmaxf((cx - lox) * sx, 0.f), 2097151.f);
    const uint32_t qy = (uint32_t)fminf(fmaxf((cy - loy) * sy, 0.f), 2097151.f);
    const uint32_t qz = (uint32_t)fminf(fmaxf((cz - loz) * sz, 0.f), 2097151.f);
    keys[i] = spread21(qx) << 2 | spread21(qy) << 1 | spread21(qz);
    ids[i] = i;
}

// common-prefix length of sorted entries i and j (-1 outside the array); equal codes are told
// apart by their position, so every pair of entries differs
__device__ __forceinline__ int prefix_len(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

// child encoding: bit 31 set = leaf (sorted position), else internal node index
__global__ void k_lbvh_hierarchy(const unsigned long long *__restrict__ keys, int n, uint32_t *__restrict__ left,
                                 uint32_t *__restrict__ right, uint32_t *__restrict__ first,
                                 uint32_t *__restrict__ last, uint32_t *__restrict__ parent_of_internal,
                                 uint32_t *__restrict__ parent_of_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (prefix_len(keys, n, i, i + 1) - prefix_len(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = prefix_len(keys, n, i, i - d);
    int lmax = 2;
    while (prefix_len(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (prefix_len(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = prefix_len(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (prefix_len(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const uint32_t lc = lo == gamma ? (0x80000000u | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t rc = hi == gamma + 1 ? (0x80000000u | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    left[i] = lc, right[i] = rc;
    first[i] = (uint32_t)lo, last[i] = (uint32_t)hi;
    if (lc & 0x80000000u) parent_of_leaf[gamma] = (uint32_t)i;
    else parent_of_internal[gamma] = (uint32_t)i;
    if (rc & 0x80000000u) parent_of_leaf[gamma + 1] = (uint32_t)i;
    else parent_of_internal[gamma + 1] = (uint32_t)i;
    if (i == 0) parent_of_internal[0] = 0xFFFFFFFFu;
}

__global__ void k_lbvh_fit(const float *__restrict__ pos, const uint32_t *__restrict__ ids, int n,
                           const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                           const uint32_t *__restrict__ parent_of_internal,
                           const uint32_t *__restrict__ parent_of_leaf, float *__restrict__ leaf_box,
                           float *__restrict__ node_box, uint32_t *__restrict__ height,
                           unsigned int *__restrict__ arrivals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = pos + (size_t)ids[i] * 9;
    float b[6];
    for (int a = 0; a < 3; ++a) {
        b[a] = fminf(fminf(p[a], p[3 + a]), p[6 + a]);
        b[3 + a] = fmaxf(fmaxf(p[a], p[3 + a]), p[6 + a]);
    }
    for (int a = 0; a < 6; ++a) leaf_box[(size_t)i * 6 + a] = b[a];
    if (n == 1) return;
    uint32_t node = parent_of_leaf[i];
    while (node != 0xFFFFFFFFu) {
        __threadfence();
        if (atomicAdd(&arrivals[node], 1u) == 0u) return;  // the sibling subtree is not done yet
        __threadfence();
        const uint32_t lc = left[node], rc = right[node];
        // (volatile: the boxes were written by other CUs; read them past this CU's L1)
        const volatile float *lb = (lc & 0x80000000u) ? leaf_box + (size_t)(lc & 0x7FFFFFFFu) * 6 : node_box + (size_t)lc * 6;
        const volatile float *rb = (rc & 0x80000000u) ? leaf_box + (size_t)(rc & 0x7FFFFFFFu) * 6 : node_box + (size_t)rc * 6;
        const volatile uint32_t *vh = height;
        const uint32_t lh = (lc & 0x80000000u) ? 0u : vh[lc], rh = (rc & 0x80000000u) ? 0u : vh[rc];
        for (int a = 0; a < 3; ++a) {
            node_box[(size_t)node * 6 + a] = fminf(lb[a], rb[a]);
            node_box[(size_t)node * 6 + 3 + a] = fmaxf(lb[3 + a], rb[3 + a]);
        }
        height[node] = max(lh, rh) + 1u;
        node = parent_of_internal[node];
    }
}

}  // namespace

bool build_bvh_lbvh(const float *pos, const float *nrm, const float *uv, uint32_t ntris, uint32_t leaf_size,
                    int device, HostBvh &out, std::string &err) {
    if (!check_bvh_input(pos, nrm, ntris, leaf_size, err)) return false;
    const int n = (int)ntris;
    LB_TRY(hipSetDevice(device));
    // centroid bounds on the host (one pass over data the host already holds)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t t = 0; t < ntris; ++t) {
        const float *p = pos + (size_t)t * 9;
        for (int a = 0; a < 3; ++a) {
            const float c = ((p[a] + p[3 + a]) + p[6 + a]) * (1.f / 3.f);
            lo[a] = fminf(lo[a], c), hi[a] = fmaxf(hi[a], c);
        }
    }
    float sc[3];
    for (int a = 0; a < 3; ++a) sc[a] = hi[a] > lo[a] ? 2097151.f / (hi[a] - lo[a]) : 0.f;

    Buf<float> d_pos, d_leaf_box, d_node_box;
    Buf<unsigned long long> d_keys, d_keys2;
    Buf<uint32_t> d_ids, d_ids2, d_left, d_right, d_first, d_last, d_pint, d_pleaf, d_height;
    Buf<unsigned int> d_arr;
    Buf<unsigned char> d_tmp;
    const size_t ni = n > 1 ? (size_t)n - 1 : 1;
    LB_TRY(d_pos.alloc((size_t)n * 9));
    LB_TRY(d_keys.alloc(n));
    LB_TRY(d_keys2.alloc(n));
    LB_TRY(d_ids.alloc(n));
    LB_TRY(d_ids2.alloc(n));
    LB_TRY(d_left.alloc(ni));
    LB_TRY(d_right.alloc(ni));
    LB_TRY(d_first.alloc(ni));
    LB_TRY(d_last.alloc(ni));
    LB_TRY(d_pint.alloc(ni));
    LB_TRY(d_pleaf.alloc(n));
    LB_TRY(d_height.alloc(ni));
    LB_TRY(d_arr.alloc(ni));
    LB_TRY(d_leaf_box.alloc((size_t)n * 6));
    LB_TRY(d_node_box.alloc(ni * 6));
    LB_TRY(hipMemcpy(d_pos.p, pos, (size_t)n * 36, hipMemcpyHostToDevice));
    LB_TRY(hipMemset(d_arr.p, 0, ni * 4));
    LB_TRY(hipMemset(d_height.p, 0, ni * 4));

    const dim3 blk(256), grd((n + 255) / 256);
    hipLaunchKernelGGL(k_lbvh_keys, grd, blk, 0, 0, d_pos.p, ntris, lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], d_keys.p,
                       d_ids.p);
    LB_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys.p, d_keys2.p, d_ids.p, d_ids2.p, n, 0, 63));
    LB_TRY(d_tmp.alloc(tmp_bytes));
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_keys.p, d_keys2.p, d_ids.p, d_ids2.p, n, 0, 63));
    if (n > 1) {
        hipLaunchKernelGGL(k_lbvh_hierarchy, dim3((n - 1 + 255) / 256), blk, 0, 0, d_keys2.p, n, d_left.p, d_right.p,
                           d_first.p, d_last.p, d_pint.p, d_pleaf.p);
        LB_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_lbvh_fit, grd, blk, 0, 0, d_pos.p, d_ids2.p, n, d_left.p, d_right.p, d_pint.p, d_pleaf.p,
                       d_leaf_box.p, d_node_box.p, d_height.p, d_arr.p);
    LB_TRY(hipGetLastError());
    LB_TRY(hipDeviceSynchronize());

    std::vector<uint32_t> left(ni), right(ni), first(ni), last(ni);
    std::vector<float> node_box(ni * 6), leaf_box((size_t)n * 6);
    out.prim_order.resize(n);
    LB_TRY(hipMemcpy(out.prim_order.data(), d_ids2.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    LB_TRY(hipMemcpy(leaf_box.data(), d_leaf_box.p, (size_t)n * 24, hipMemcpyDeviceToHost));
    if (n > 1) {
        LB_TRY(hipMemcpy(left.data(), d_left.p, ni * 4, hipMemcpyDeviceToHost));
        LB_TRY(hipMemcpy(right.data(), d_right.p, ni * 4, hipMemcpyDeviceToHost));
        LB_TRY(hipMemcpy(first.data(), d_first.p, ni * 4, hipMemcpyDeviceToHost));
        LB_TRY(hipMemcpy(last.data(), d_last.p, ni * 4, hipMemcpyDeviceToHost));
        LB_TRY(hipMemcpy(node_box.data(), d_node_box.p, ni * 24, hipMemcpyDeviceToHost));
    }

    // depth-first walk into the reference's flat layout; subtrees of <= leaf_size triangles become leaves
    out.start.clear(), out.nprims.clear(), out.right_offset.clear(), out.bbox.clear();
    out.n_leaves = 0, out.max_depth = 0;
    struct Item {
        uint32_t child, parent, depth;
        bool is_right;
    };
    std::vector<Item> work;
    work.push_back({n > 1 ? 0u : 0x80000000u, 0u, 0u, false});
    while (!work.empty()) {
        const Item it = work.back();
        work.pop_back();
        const uint32_t me = (uint32_t)out.start.size();
        const bool is_prim = (it.child & 0x80000000u) != 0;
        const uint32_t idx = it.child & 0x7FFFFFFFu;
        const uint32_t begin = is_prim ? idx : first[idx], end = is_prim ? idx + 1 : last[idx] + 1;
        const float *box = is_prim ? &leaf_box[(size_t)idx * 6] : &node_box[(size_t)idx * 6];
        const bool leaf = end - begin <= leaf_size;
        out.start.push_back(begin);
        out.nprims.push_back(end - begin);
        out.right_offset.push_back(leaf ? 0u : 0xffffffffu);
        for (int a = 0; a < 6; ++a) out.bbox.push_back(box[a]);
        if (it.depth > out.max_depth) out.max_depth = it.depth;
        if (it.is_right) out.right_offset[it.parent] = me - it.parent;
        if (leaf) {
            out.n_leaves++;
            continue;
        }
        work.push_back({right[idx], me, it.depth + 1, true});  // popped after the whole left subtree
        work.push_back({left[idx], me, it.depth + 1, false});
    }
    return flatten_bvh(pos, nrm, uv, ntris, out, err);
}

}  // namespace vmx
