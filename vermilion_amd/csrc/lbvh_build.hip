// lbvh_build.hip — BVH construction on the GPU: a linear BVH, and a quality tree by parallel
// locally-ordered clustering (k_ploc_*, below) that shares its Morton sort, record emission and flat export.
//
//   k_lbvh_bounds     centroid bounds (wave min/max + integer atomics on order-preserving float keys)
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
//   k_lbvh_emit_*     the device records the traversal kernels read (InnerRecord / TriRecord /
//                     AttrRecord), written straight from the hierarchy: every subtree of at most
//                     `leaf_size` triangles becomes one leaf reference; inner record i is Karras node i
//
// Nothing returns to the host during a build (only the root's height, for the stack depth), so scene
// creation with this builder is a per-frame operation.  The reference's flat layout (bvh.h:11-14) is
// produced on demand (lbvh_export_flat: vmx_scene_bvh / vmx_scene_describe) by one depth-first host
// walk over the downloaded hierarchy.  Like the SAH builder this is NOT the reference's topology: for
// throughput runs and scenes that change per frame, not for triangle-ID parity runs.  Its parity bar is
// the same: the reference's traversal run over the exported tree gives bit-identical hits (tests).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <algorithm>
#include "bvh_build.h"

namespace vmx {
namespace {

#ifndef VMX_PLOC_RADIUS
#define VMX_PLOC_RADIUS 16
#endif
constexpr int kPlocRadius = VMX_PLOC_RADIUS;  // neighbours searched on each side (PLOC)

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


// ---- device records straight from the hierarchy (no host flatten) ------------------------------------
// Inner record i IS Karras node i (records of nodes inside a collapsed subtree stay unused: the array is
// ntris - 1 slots, references are indices, so no compaction or renumbering pass is needed); a subtree of at
// most `leaf_size` triangles becomes a leaf reference (first sorted position, count) in its parent.
__device__ __forceinline__ uint32_t lbvh_child_ref(uint32_t c, uint32_t leaf_size, const uint32_t *first,
                                                   const uint32_t *last, const float *leaf_box, const float *node_box,
                                                   const float *&box) {
    if (c & 0x80000000u) {
        const uint32_t p = c & 0x7FFFFFFFu;
        box = leaf_box + (size_t)p * 6;
        return kLeafBit | (1u << kLeafCountShift) | p;
    }
    box = node_box + (size_t)c * 6;
    const uint32_t cnt = last[c] - first[c] + 1u;
    return cnt <= leaf_size ? (kLeafBit | (cnt << kLeafCountShift) | first[c]) : c;
}

__global__ void k_lbvh_emit_inner(int n, uint32_t leaf_size, const uint32_t *__restrict__ left,
                                  const uint32_t *__restrict__ right, const uint32_t *__restrict__ first,
                                  const uint32_t *__restrict__ last, const float *__restrict__ leaf_box,
                                  const float *__restrict__ node_box, InnerRecord *__restrict__ inner) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    InnerRecord r;
    if (last[i] - first[i] + 1u <= leaf_size) {  // inside (or the top of) a collapsed subtree: never referenced
        for (int a = 0; a < 3; ++a) r.lmin[a] = r.lmax[a] = r.rmin[a] = r.rmax[a] = 0.f;
        r.left = r.right = r.pad0 = r.pad1 = 0;
        inner[i] = r;
        return;
    }
    const float *lb, *rb;
    r.left = lbvh_child_ref(left[i], leaf_size, first, last, leaf_box, node_box, lb);
    r.right = lbvh_child_ref(right[i], leaf_size, first, last, leaf_box, node_box, rb);
    for (int a = 0; a < 3; ++a) r.lmin[a] = lb[a], r.lmax[a] = lb[3 + a], r.rmin[a] = rb[a], r.rmax[a] = rb[3 + a];
    r.pad0 = r.pad1 = 0;
    inner[i] = r;
}

// TriRecord / AttrRecord of sorted position `slot` (leaf order), as bvh_build.cpp's flatten writes them
__global__ void k_lbvh_emit_tris(int n, const uint32_t *__restrict__ ids, const float *__restrict__ pos,
                                 const float *__restrict__ nrm, const float *__restrict__ uv, TriRecord *__restrict__ tris,
                                 AttrRecord *__restrict__ attrs) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    const uint32_t t = ids[slot];
    const float *p = pos + (size_t)t * 9, *q = nrm + (size_t)t * 9;
    TriRecord tr;
    for (int a = 0; a < 3; ++a) {
        tr.v0[a] = p[a];
        tr.e1[a] = p[3 + a] - p[a];  // triangle.cpp:12
        tr.e2[a] = p[6 + a] - p[a];  // triangle.cpp:13
    }
    tr.id = t, tr.pad[0] = tr.pad[1] = 0;
    tris[slot] = tr;
    AttrRecord ar;
    for (int a = 0; a < 3; ++a) ar.n0[a] = q[a], ar.n1[a] = q[3 + a], ar.n2[a] = q[6 + a];
    for (int a = 0; a < 2; ++a) {
        ar.uv0[a] = uv ? uv[(size_t)t * 6 + a] : 0.f;
        ar.uv1[a] = uv ? uv[(size_t)t * 6 + 2 + a] : 0.f;
        ar.uv2[a] = uv ? uv[(size_t)t * 6 + 4 + a] : 0.f;
    }
    ar.pad = 0.f;
    attrs[slot] = ar;
}

// centroid bounds on the device: one block-level min/max + 6 float atomics per block (values are
// order-independent, so the result is deterministic)
__global__ void k_lbvh_bounds(const float *__restrict__ pos, uint32_t n, int *__restrict__ bounds /* keys of lo[3], hi[3] */) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *p = pos + (size_t)i * 9;
        for (int a = 0; a < 3; ++a) {
            const float c = ((p[a] + p[3 + a]) + p[6 + a]) * (1.f / 3.f);
            lo[a] = fminf(lo[a], c), hi[a] = fmaxf(hi[a], c);
        }
    }
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
        }
    }
    if ((threadIdx.x & 63u) == 0) {
        for (int a = 0; a < 3; ++a) {
            // float min/max through integer atomics on the order-preserving key (centroids are finite)
            atomicMin(&bounds[a], __float_as_int(lo[a]) >= 0 ? __float_as_int(lo[a]) : (int)(0x80000000u - (uint32_t)__float_as_int(lo[a])));
            atomicMax(&bounds[3 + a], __float_as_int(hi[a]) >= 0 ? __float_as_int(hi[a]) : (int)(0x80000000u - (uint32_t)__float_as_int(hi[a])));
        }
    }
}

__global__ void k_lbvh_keys(const float *__restrict__ pos, uint32_t n, const int *__restrict__ bounds_key,
                                unsigned long long *__restrict__ keys, uint32_t *__restrict__ ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float lo[3], sc[3];
    for (int a = 0; a < 3; ++a) {
        const int kl = bounds_key[a], kh = bounds_key[3 + a];
        const float l = __int_as_float(kl >= 0 ? kl : (int)(0x80000000u - (uint32_t)kl));
        const float h = __int_as_float(kh >= 0 ? kh : (int)(0x80000000u - (uint32_t)kh));
        lo[a] = l;
        sc[a] = h > l ? 2097151.f / (h - l) : 0.f;
    }
    const float *p = pos + (size_t)i * 9;
    const float cx = ((p[0] + p[3]) + p[6]) * (1.f / 3.f), cy = ((p[1] + p[4]) + p[7]) * (1.f / 3.f),
                cz = ((p[2] + p[5]) + p[8]) * (1.f / 3.f);
    const uint32_t qx = (uint32_t)fminf(fmaxf((cx - lo[0]) * sc[0], 0.f), 2097151.f);
    const uint32_t qy = (uint32_t)fminf(fmaxf((cy - lo[1]) * sc[1], 0.f), 2097151.f);
    const uint32_t qz = (uint32_t)fminf(fmaxf((cz - lo[2]) * sc[2], 0.f), 2097151.f);
    keys[i] = spread21(qx) << 2 | spread21(qy) << 1 | spread21(qz);
    ids[i] = i;
}


// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) -----------------------------------
// A quality tree built bottom-up on the GPU: the clusters (at first the triangles in Morton order) each look
// for the neighbour within `radius` positions whose union box has the smallest surface area; pairs that chose
// each other merge into a new node, the cluster array is compacted, and the rounds repeat until one cluster
// is left.  Everything that orders the result is a scan or a strict total order on pairs (area, lower
// position, higher position), so the tree is the same on every run.  Node indices are handed out from the top
// (the last merge gets index 0): the emission kernels take inner record 0 as the root, as for the Karras tree.
__device__ __forceinline__ float ploc_half_area(const float *a, const float *b) {
    const float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]), dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]),
                dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return dx * dy + dy * dz + dz * dx;
}

__global__ void k_ploc_leaf_boxes(const float *__restrict__ pos, const uint32_t *__restrict__ ids_m, int n,
                                  float *__restrict__ box_m, uint32_t *__restrict__ c_node, float *__restrict__ c_box) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = pos + (size_t)ids_m[i] * 9;
    for (int a = 0; a < 3; ++a) {
        const float lo = fminf(fminf(p[a], p[3 + a]), p[6 + a]), hi = fmaxf(fmaxf(p[a], p[3 + a]), p[6 + a]);
        box_m[(size_t)i * 6 + a] = lo, box_m[(size_t)i * 6 + 3 + a] = hi;
        c_box[(size_t)i * 6 + a] = lo, c_box[(size_t)i * 6 + 3 + a] = hi;
    }
    c_node[i] = 0x80000000u | (uint32_t)i;
}

__global__ void k_ploc_nearest(int nc, int radius, const float *__restrict__ c_box, uint32_t *__restrict__ nn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    float mine[6];
    for (int a = 0; a < 6; ++a) mine[a] = c_box[(size_t)i * 6 + a];
    float best = INFINITY;
    int bj = -1;
    const int lo = max(i - radius, 0), hi = min(i + radius, nc - 1);
    for (int j = lo; j <= hi; ++j) {
        if (j == i) continue;
        const float a = ploc_half_area(mine, c_box + (size_t)j * 6);
        // strict total order on pairs: (area, lower position, higher position)
        bool better = bj < 0 || a < best;
        if (!better && a == best) {
            const int m0 = min(i, j), m1 = max(i, j), b0 = min(i, bj), b1 = max(i, bj);
            better = m0 < b0 || (m0 == b0 && m1 < b1);
        }
        if (better) best = a, bj = j;
    }
    nn[i] = (uint32_t)bj;
}

// survivor (bit 0) and merge-leader (bit 32) flags of cluster i
__global__ void k_ploc_flags(int nc, const uint32_t *__restrict__ nn, unsigned long long *__restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const uint32_t j = nn[i];
    const bool mutual = nn[j] == (uint32_t)i;
    const bool leader = mutual && (uint32_t)i < j;
    const bool absorbed = mutual && (uint32_t)i > j;
    flags[i] = (absorbed ? 0ull : 1ull) | (leader ? (1ull << 32) : 0ull);
}

__global__ void k_ploc_commit(int nc, uint32_t base, const uint32_t *__restrict__ nn,
                              const unsigned long long *__restrict__ flags, const unsigned long long *__restrict__ scan,
                              const uint32_t *__restrict__ c_node, const float *__restrict__ c_box,
                              uint32_t *__restrict__ n_node, float *__restrict__ n_box, uint32_t *__restrict__ left,
                              uint32_t *__restrict__ right, uint32_t *__restrict__ parent_of_internal,
                              uint32_t *__restrict__ parent_of_leaf, float *__restrict__ node_box,
                              uint32_t *__restrict__ height, uint32_t *__restrict__ size) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const unsigned long long f = flags[i];
    if (!(f & 1ull)) return;  // absorbed by its partner
    const uint32_t pos = (uint32_t)scan[i];
    if (!(f >> 32)) {
        n_node[pos] = c_node[i];
        for (int a = 0; a < 6; ++a) n_box[(size_t)pos * 6 + a] = c_box[(size_t)i * 6 + a];
        return;
    }
    const uint32_t j = nn[i], idx = base - 1u - (uint32_t)(scan[i] >> 32);
    const uint32_t lc = c_node[i], rc = c_node[j];
    left[idx] = lc, right[idx] = rc;
    float b[6];
    for (int a = 0; a < 3; ++a) {
        b[a] = fminf(c_box[(size_t)i * 6 + a], c_box[(size_t)j * 6 + a]);
        b[3 + a] = fmaxf(c_box[(size_t)i * 6 + 3 + a], c_box[(size_t)j * 6 + 3 + a]);
    }
    for (int a = 0; a < 6; ++a) node_box[(size_t)idx * 6 + a] = b[a], n_box[(size_t)pos * 6 + a] = b[a];
    const uint32_t lh = (lc & 0x80000000u) ? 0u : height[lc], rh = (rc & 0x80000000u) ? 0u : height[rc];
    const uint32_t ls = (lc & 0x80000000u) ? 1u : size[lc], rs = (rc & 0x80000000u) ? 1u : size[rc];
    height[idx] = max(lh, rh) + 1u, size[idx] = ls + rs;
    if (lc & 0x80000000u) parent_of_leaf[lc & 0x7FFFFFFFu] = idx;
    else parent_of_internal[lc] = idx;
    if (rc & 0x80000000u) parent_of_leaf[rc & 0x7FFFFFFFu] = idx;
    else parent_of_internal[rc] = idx;
    parent_of_internal[idx] = 0xFFFFFFFFu;  // until a later round gives it a parent
    n_node[pos] = idx;
}

// depth-first position of every triangle (and first / last of every node): the triangles of a subtree must be
// contiguous in leaf order, and merging clusters that are not neighbours breaks the Morton order's contiguity.
// A node's first position = the triangles to the left of it = the sizes of the left siblings on its way up.
__device__ __forceinline__ uint32_t ploc_first(uint32_t self, uint32_t node, const uint32_t *left, const uint32_t *right,
                                               const uint32_t *parent_of_internal, const uint32_t *size) {
    uint32_t pos = 0;
    while (node != 0xFFFFFFFFu) {
        if (right[node] == self) {
            const uint32_t l = left[node];
            pos += (l & 0x80000000u) ? 1u : size[l];
        }
        self = node;
        node = parent_of_internal[node];
    }
    return pos;
}
__global__ void k_ploc_order(int n, const uint32_t *__restrict__ ids_m, const float *__restrict__ box_m,
                             const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                             const uint32_t *__restrict__ parent_of_internal, const uint32_t *__restrict__ parent_of_leaf,
                             const uint32_t *__restrict__ size, uint32_t *__restrict__ dfs_of_slot, uint32_t *__restrict__ ids,
                             float *__restrict__ leaf_box, uint32_t *__restrict__ first, uint32_t *__restrict__ last) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t pos = n > 1 ? ploc_first(0x80000000u | (uint32_t)i, parent_of_leaf[i], left, right, parent_of_internal, size) : 0u;
        dfs_of_slot[i] = pos;
        ids[pos] = ids_m[i];
        for (int a = 0; a < 6; ++a) leaf_box[(size_t)pos * 6 + a] = box_m[(size_t)i * 6 + a];
    }
    if (i < n - 1) {
        const uint32_t f = ploc_first((uint32_t)i, parent_of_internal[i], left, right, parent_of_internal, size);
        first[i] = f, last[i] = f + size[i] - 1u;
    }
}
// child references to triangles: Morton slot -> depth-first position
__global__ void k_ploc_relink(int n, const uint32_t *__restrict__ dfs_of_slot, uint32_t *__restrict__ left,
                              uint32_t *__restrict__ right) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const uint32_t l = left[i], r = right[i];
    if (l & 0x80000000u) left[i] = 0x80000000u | dfs_of_slot[l & 0x7FFFFFFFu];
    if (r & 0x80000000u) right[i] = 0x80000000u | dfs_of_slot[r & 0x7FFFFFFFu];
}

template <class T>
T *carve(unsigned char *&cursor, size_t count) {
    T *p = (T *)cursor;
    cursor += (count * sizeof(T) + 255) & ~(size_t)255;
    return p;
}

}  // namespace

void lbvh_release(LbvhDevice &d) {
    if (d.arena) (void)hipFree(d.arena);
    if (d.geom) (void)hipFree(d.geom);
    if (d.attrs) (void)hipFree(d.attrs);
    d = LbvhDevice{};
}

// The whole build on the device: upload -> Morton keys -> radix sort -> Karras hierarchy -> bottom-up fit ->
// InnerRecord / TriRecord / AttrRecord emission.  Nothing comes back to the host except the root's height
// (for the traversal stack depth); the hierarchy arrays stay in `out.arena` for lbvh_export_flat, everything else
// the build needed is released before it returns.
static bool build_device(const float *pos, const float *nrm, const float *uv, uint32_t ntris, uint32_t leaf_size,
                         int device, bool ploc, LbvhDevice &out, std::string &err) {
    if (!check_bvh_input(pos, nrm, ntris, leaf_size, err)) return false;
    const int n = (int)ntris;
    const size_t ni = n > 1 ? (size_t)n - 1 : 1;
    LB_TRY(hipSetDevice(device));
    out = LbvhDevice{};
    out.ntris = ntris, out.leaf_size = leaf_size, out.n_inner = (uint32_t)ni;
    size_t tmp_bytes = 0;
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                              (uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 63));
    // two allocations (a hipMalloc per array costs more than the kernels): `arena` keeps the hierarchy for
    // lbvh_export_flat (68 B per triangle, counted in vmx_scene_desc.device_bytes); `scratch` holds the input copies,
    // keys, sort / scan temporaries and the PLOC cluster buffers (250-400 B per triangle) and is freed when the
    // records have been written
    const size_t in_floats = (size_t)n * (uv ? 24 : 18);
    size_t arena_bytes = 0, scratch_bytes = 0;
    auto keep = [&](size_t bytes) { arena_bytes += (bytes + 255) & ~(size_t)255; };
    auto add = [&](size_t bytes) { scratch_bytes += (bytes + 255) & ~(size_t)255; };
    keep((size_t)n * 4);                                     // ids
    for (int k = 0; k < 4; ++k) keep(ni * 4);                // left right first last
    keep((size_t)n * 24), keep(ni * 24);                     // leaf_box node_box
    add(in_floats * 4), add((size_t)n * 8), add((size_t)n * 8), add((size_t)n * 4);  // inputs, keys x2, ids
    add(ni * 4), add(ni * 4), add((size_t)n * 4), add(ni * 4), add(256), add(tmp_bytes);  // pint height pleaf arrivals bounds tmp
    size_t scan_bytes = 0;
    if (ploc) {
        LB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, n));
        add((size_t)n * 24), add((size_t)n * 4), add((size_t)n * 4);                        // box_m, cluster nodes x2
        add((size_t)n * 24), add((size_t)n * 24), add((size_t)n * 4), add((size_t)n * 4);   // cluster boxes x2, nn, dfs_of_slot
        add((size_t)n * 8), add((size_t)n * 8), add(ni * 4), add(scan_bytes);               // flags, scan, size, scan temp
    }
    LB_TRY(hipMalloc(&out.arena, arena_bytes));
    out.arena_bytes = arena_bytes;
    struct Scratch {  // freed on every way out of this function
        void *p = nullptr;
        ~Scratch() {
            if (p) (void)hipFree(p);
        }
    } scratch;
    LB_TRY(hipMalloc(&scratch.p, scratch_bytes));
    unsigned char *kcur = (unsigned char *)out.arena, *cur = (unsigned char *)scratch.p;
    out.ids = carve<uint32_t>(kcur, n);
    out.left = carve<uint32_t>(kcur, ni), out.right = carve<uint32_t>(kcur, ni);
    out.first = carve<uint32_t>(kcur, ni), out.last = carve<uint32_t>(kcur, ni);
    out.leaf_box = carve<float>(kcur, (size_t)n * 6), out.node_box = carve<float>(kcur, ni * 6);
    float *d_in = carve<float>(cur, in_floats);
    float *d_pos = d_in, *d_nrm = d_in + (size_t)n * 9, *d_uv = uv ? d_in + (size_t)n * 18 : nullptr;
    unsigned long long *d_keys = carve<unsigned long long>(cur, n), *d_keys2 = carve<unsigned long long>(cur, n);
    uint32_t *d_ids = carve<uint32_t>(cur, n);
    uint32_t *d_pint = carve<uint32_t>(cur, ni), *d_height = carve<uint32_t>(cur, ni);
    uint32_t *d_pleaf = carve<uint32_t>(cur, n);
    unsigned int *d_arr = carve<unsigned int>(cur, ni);
    int *d_bounds = carve<int>(cur, 64);
    unsigned char *d_tmp = carve<unsigned char>(cur, tmp_bytes);

    const size_t inner_bytes = ni * sizeof(InnerRecord), tri_bytes = (size_t)n * sizeof(TriRecord);
    if (inner_bytes + tri_bytes + 64 > 0xFFFFFFFFull) {
        err = "scene too large for 32-bit record offsets";
        return false;
    }
    LB_TRY(hipMalloc(&out.geom, inner_bytes + tri_bytes + 64));
    LB_TRY(hipMalloc(&out.attrs, (size_t)n * sizeof(AttrRecord)));
    out.geom_bytes = inner_bytes + tri_bytes + 64, out.tri_off = (uint32_t)inner_bytes;

    hipStream_t s = 0;
    LB_TRY(hipMemcpyAsync(d_pos, pos, (size_t)n * 36, hipMemcpyHostToDevice, s));
    LB_TRY(hipMemcpyAsync(d_nrm, nrm, (size_t)n * 36, hipMemcpyHostToDevice, s));
    if (uv) LB_TRY(hipMemcpyAsync(d_uv, uv, (size_t)n * 24, hipMemcpyHostToDevice, s));
    LB_TRY(hipMemsetAsync(d_arr, 0, ni * 4, s));
    LB_TRY(hipMemsetAsync(d_height, 0, ni * 4, s));
    LB_TRY(hipMemsetAsync((unsigned char *)out.geom + inner_bytes + tri_bytes, 0, 64, s));
    const int init_bounds[6] = {0x7F800000, 0x7F800000, 0x7F800000, (int)0x80800000u, (int)0x80800000u, (int)0x80800000u};  // keys of +inf / -inf
    LB_TRY(hipMemcpyAsync(d_bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, s));

    const dim3 blk(256), grd((n + 255) / 256);
    hipLaunchKernelGGL(k_lbvh_bounds, dim3(std::min(1024, (n + 255) / 256)), blk, 0, s, d_pos, ntris, d_bounds);
    hipLaunchKernelGGL(k_lbvh_keys, grd, blk, 0, s, d_pos, ntris, d_bounds, d_keys, d_ids);
    LB_TRY(hipGetLastError());
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_ids, out.ids, n, 0, 63, s));
    if (ploc) {
        float *box_m = carve<float>(cur, (size_t)n * 6);
        uint32_t *c_node[2] = {carve<uint32_t>(cur, n), carve<uint32_t>(cur, n)};
        float *c_box[2] = {carve<float>(cur, (size_t)n * 6), carve<float>(cur, (size_t)n * 6)};
        uint32_t *d_nn = carve<uint32_t>(cur, n), *d_dfs = carve<uint32_t>(cur, n);
        unsigned long long *d_flags = carve<unsigned long long>(cur, n), *d_scan = carve<unsigned long long>(cur, n);
        uint32_t *d_size = carve<uint32_t>(cur, ni);
        unsigned char *d_scan_tmp = carve<unsigned char>(cur, scan_bytes);
        // out.ids holds the Morton order until k_ploc_order rewrites it in depth-first order: keep a copy in d_ids
        LB_TRY(hipMemcpyAsync(d_ids, out.ids, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_ploc_leaf_boxes, grd, blk, 0, s, d_pos, d_ids, n, box_m, c_node[0], c_box[0]);
        LB_TRY(hipGetLastError());
        int nc = n, from = 0;
        uint32_t base = (uint32_t)(n - 1);
        while (nc > 1) {
            const dim3 g((nc + 255) / 256);
            hipLaunchKernelGGL(k_ploc_nearest, g, blk, 0, s, nc, kPlocRadius, c_box[from], d_nn);
            hipLaunchKernelGGL(k_ploc_flags, g, blk, 0, s, nc, d_nn, d_flags);
            LB_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_tmp, scan_bytes, d_flags, d_scan, nc, s));
            hipLaunchKernelGGL(k_ploc_commit, g, blk, 0, s, nc, base, d_nn, d_flags, d_scan, c_node[from], c_box[from],
                               c_node[from ^ 1], c_box[from ^ 1], out.left, out.right, d_pint, d_pleaf, out.node_box,
                               d_height, d_size);
            LB_TRY(hipGetLastError());
            unsigned long long tail[2];  // totals = last scan value + last flag
            LB_TRY(hipMemcpyAsync(&tail[0], d_scan + (nc - 1), 8, hipMemcpyDeviceToHost, s));
            LB_TRY(hipMemcpyAsync(&tail[1], d_flags + (nc - 1), 8, hipMemcpyDeviceToHost, s));
            LB_TRY(hipStreamSynchronize(s));
            const unsigned long long tot = tail[0] + tail[1];
            const uint32_t merged = (uint32_t)(tot >> 32);
            if (merged == 0) {
                err = "PLOC builder: a round merged nothing";
                return false;
            }
            nc = (int)(uint32_t)tot, base -= merged, from ^= 1;
        }
        hipLaunchKernelGGL(k_ploc_order, grd, blk, 0, s, n, d_ids, box_m, out.left, out.right, d_pint, d_pleaf, d_size,
                           d_dfs, out.ids, out.leaf_box, out.first, out.last);
        if (n > 1) hipLaunchKernelGGL(k_ploc_relink, dim3((n - 1 + 255) / 256), blk, 0, s, n, d_dfs, out.left, out.right);
        LB_TRY(hipGetLastError());
    } else {
        if (n > 1) {
            hipLaunchKernelGGL(k_lbvh_hierarchy, dim3((n - 1 + 255) / 256), blk, 0, s, d_keys2, n, out.left, out.right,
                               out.first, out.last, d_pint, d_pleaf);
            LB_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(k_lbvh_fit, grd, blk, 0, s, d_pos, out.ids, n, out.left, out.right, d_pint, d_pleaf, out.leaf_box,
                           out.node_box, d_height, d_arr);
    }
    if (n > 1)
        hipLaunchKernelGGL(k_lbvh_emit_inner, dim3((n - 1 + 255) / 256), blk, 0, s, n, leaf_size, out.left, out.right,
                           out.first, out.last, out.leaf_box, out.node_box, (InnerRecord *)out.geom);
    else
        LB_TRY(hipMemsetAsync(out.geom, 0, sizeof(InnerRecord), s));
    hipLaunchKernelGGL(k_lbvh_emit_tris, grd, blk, 0, s, n, out.ids, d_pos, d_nrm, d_uv,
                       (TriRecord *)((unsigned char *)out.geom + inner_bytes), (AttrRecord *)out.attrs);
    LB_TRY(hipGetLastError());
    uint32_t h = 0;
    if (n > 1) LB_TRY(hipMemcpyAsync(&h, d_height, 4, hipMemcpyDeviceToHost, s));
    LB_TRY(hipStreamSynchronize(s));
    out.height = h;  // edges from the root to the deepest primitive: an upper bound of the collapsed tree's depth
    if (out.height + 2 > kMaxStack) {
        err = "BVH deeper than the reference's 64-entry traversal stack (bvh.cpp:54)";
        return false;
    }
    out.root_ref = ntris <= leaf_size ? (kLeafBit | (ntris << kLeafCountShift)) : 0u;
    return true;
}

bool build_bvh_lbvh_device(const float *pos, const float *nrm, const float *uv, uint32_t ntris, uint32_t leaf_size,
                           int device, LbvhDevice &out, std::string &err) {
    return build_device(pos, nrm, uv, ntris, leaf_size, device, false, out, err);
}
bool build_bvh_ploc_device(const float *pos, const float *nrm, const float *uv, uint32_t ntris, uint32_t leaf_size,
                           int device, LbvhDevice &out, std::string &err) {
    return build_device(pos, nrm, uv, ntris, leaf_size, device, true, out, err);
}

// The reference's flat layout (bvh.h:11-14) of a device-built tree, for vmx_scene_bvh / vmx_scene_describe:
// downloads the hierarchy and walks it depth first (pre-order, left child = i + 1); subtrees of at most
// leaf_size triangles become leaves — the same collapse rule as k_lbvh_emit_inner.
bool lbvh_export_flat(const LbvhDevice &d, int device, HostBvh &out, std::string &err) {
    LB_TRY(hipSetDevice(device));
    const int n = (int)d.ntris;
    const size_t ni = n > 1 ? (size_t)n - 1 : 1;
    std::vector<uint32_t> left(ni), right(ni), first(ni), last(ni);
    std::vector<float> node_box(ni * 6), leaf_box((size_t)n * 6);
    out.prim_order.resize(n);
    LB_TRY(hipMemcpy(out.prim_order.data(), d.ids, (size_t)n * 4, hipMemcpyDeviceToHost));
    LB_TRY(hipMemcpy(leaf_box.data(), d.leaf_box, (size_t)n * 24, hipMemcpyDeviceToHost));
    if (n > 1) {
        LB_TRY(hipMemcpy(left.data(), d.left, ni * 4, hipMemcpyDeviceToHost));
        LB_TRY(hipMemcpy(right.data(), d.right, ni * 4, hipMemcpyDeviceToHost));
        LB_TRY(hipMemcpy(first.data(), d.first, ni * 4, hipMemcpyDeviceToHost));
        LB_TRY(hipMemcpy(last.data(), d.last, ni * 4, hipMemcpyDeviceToHost));
        LB_TRY(hipMemcpy(node_box.data(), d.node_box, ni * 24, hipMemcpyDeviceToHost));
    }
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
        const bool leaf = end - begin <= d.leaf_size;
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
    return true;
}

}  // namespace vmx
