// bvh_build.cpp — BVH with the reference's topology, built over index arrays.
//
// Follows BVH::build (core/accelerators/bvh.cpp:179-279) so that triangle IDs,
// tie order and traversal order on the device equal the reference's:
//   * node bounds = union of Triangle::getBBox (triangle.cpp:107-114)
//   * split axis from the centroid bounds via BBox::maxDimension, which
//     compares z with y only (bbox.cpp:41-46)
//   * centroid = (v0+v1+v2)*0.333f (triangle.cpp:116-119)
//   * split at the midpoint of the centroid bounds, in-place partition that
//     swaps element i with element `mid` (bvh.cpp:246-255)
//   * degenerate partition -> split at the middle index (bvh.cpp:258-260)
//   * leaf when nPrims <= leafSize (bvh.cpp:219); DFS pre-order, left first
#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>

namespace vmx {
namespace {

struct Span {
    uint32_t begin, end, parent, depth;
    bool is_right;
};

inline float fmin2(float a, float b) { return b < a ? b : a; }
inline float fmax2(float a, float b) { return a < b ? b : a; }

}  // namespace

namespace {

// per-triangle bounds and centroid (triangle.cpp:107-119), computed once
void triangle_bounds(const float *pos, uint32_t ntris, std::vector<float> &lo, std::vector<float> &hi,
                     std::vector<float> &cen) {
    lo.resize((size_t)ntris * 3), hi.resize((size_t)ntris * 3), cen.resize((size_t)ntris * 3);
    for (uint32_t t = 0; t < ntris; ++t) {
        const float *p = pos + (size_t)t * 9;
        for (int a = 0; a < 3; ++a) {
            lo[(size_t)t * 3 + a] = fmin2(fmin2(p[a], p[3 + a]), p[6 + a]);
            hi[(size_t)t * 3 + a] = fmax2(fmax2(p[a], p[3 + a]), p[6 + a]);
            cen[(size_t)t * 3 + a] = ((p[a] + p[3 + a]) + p[6 + a]) * 0.333f;
        }
    }
}

bool check_input(const float *pos, const float *nrm, uint32_t ntris, uint32_t &leaf_size, std::string &err) {
    if (!pos || !nrm || ntris == 0) {
        err = "scene needs positions, normals and at least one triangle";
        return false;
    }
    if (ntris > kMaxTris) {
        err = "too many triangles for the 26-bit leaf reference";
        return false;
    }
    if (leaf_size == 0) leaf_size = 4;  // bvh.h:29
    if (leaf_size > kMaxLeafSize) {
        err = "leaf_size above 31 does not fit the leaf reference";
        return false;
    }
    for (size_t i = 0; i < (size_t)ntris * 9; ++i) {
        if (!std::isfinite(pos[i])) {
            err = "non-finite vertex position";
            return false;
        }
    }
    return true;
}

// flat tree (reference layout) -> 2-wide device records, leaf-ordered triangle / attribute records
bool flatten(const float *pos, const float *nrm, const float *uv, uint32_t ntris, HostBvh &out, std::string &err) {
    std::vector<uint32_t> &order = out.prim_order;
    const uint32_t n_nodes = (uint32_t)out.start.size();
    if (out.max_depth + 2 > kMaxStack) {
        err = "BVH deeper than the reference's 64-entry traversal stack (bvh.cpp:54)";
        return false;
    }

    // ---- flatten to 2-wide records ---------------------------------------
    std::vector<uint32_t> inner_index(n_nodes, 0xffffffffu);
    uint32_t n_inner = 0;
    for (uint32_t i = 0; i < n_nodes; ++i)
        if (out.right_offset[i] != 0) inner_index[i] = n_inner++;
    auto ref_of = [&](uint32_t node) -> uint32_t {
        if (out.right_offset[node] == 0)
            return kLeafBit | (out.nprims[node] << kLeafCountShift) | out.start[node];
        return inner_index[node];
    };
    out.inner.assign(n_inner, InnerRecord{});
    for (uint32_t i = 0; i < n_nodes; ++i) {
        if (out.right_offset[i] == 0) continue;
        InnerRecord &r = out.inner[inner_index[i]];
        const uint32_t l = i + 1, rr = i + out.right_offset[i];
        std::memcpy(r.lmin, &out.bbox[(size_t)l * 6], 12);
        std::memcpy(r.lmax, &out.bbox[(size_t)l * 6 + 3], 12);
        std::memcpy(r.rmin, &out.bbox[(size_t)rr * 6], 12);
        std::memcpy(r.rmax, &out.bbox[(size_t)rr * 6 + 3], 12);
        r.left = ref_of(l);
        r.right = ref_of(rr);
    }
    out.root_ref = ref_of(0);

    out.tris.assign(ntris, TriRecord{});
    out.attrs.assign(ntris, AttrRecord{});
    for (uint32_t slot = 0; slot < ntris; ++slot) {
        const uint32_t t = order[slot];
        const float *p = pos + (size_t)t * 9, *n = nrm + (size_t)t * 9;
        TriRecord &tr = out.tris[slot];
        for (int a = 0; a < 3; ++a) {
            tr.v0[a] = p[a];
            tr.e1[a] = p[3 + a] - p[a];  // triangle.cpp:12
            tr.e2[a] = p[6 + a] - p[a];  // triangle.cpp:13
        }
        tr.id = t;
        AttrRecord &ar = out.attrs[slot];
        std::memcpy(ar.n0, n, 12);
        std::memcpy(ar.n1, n + 3, 12);
        std::memcpy(ar.n2, n + 6, 12);
        if (uv) {
            const float *q = uv + (size_t)t * 6;
            std::memcpy(ar.uv0, q, 8);
            std::memcpy(ar.uv1, q + 2, 8);
            std::memcpy(ar.uv2, q + 4, 8);
        }
    }
    return true;
}

}  // namespace

bool build_bvh(const float *pos, const float *nrm, const float *uv, uint32_t ntris,
               uint32_t leaf_size, HostBvh &out, std::string &err) {
    if (!check_input(pos, nrm, ntris, leaf_size, err)) return false;
    std::vector<float> lo, hi, cen;
    triangle_bounds(pos, ntris, lo, hi, cen);

    std::vector<uint32_t> &order = out.prim_order;
    order.resize(ntris);
    for (uint32_t t = 0; t < ntris; ++t) order[t] = t;

    out.start.clear(), out.nprims.clear(), out.right_offset.clear(), out.bbox.clear();
    out.n_leaves = 0;
    out.max_depth = 0;
    std::vector<uint32_t> node_depth;

    std::vector<Span> work;
    work.push_back({0u, ntris, 0u, 0u, false});
    while (!work.empty()) {
        const Span s = work.back();
        work.pop_back();
        const uint32_t me = (uint32_t)out.start.size();
        const uint32_t count = s.end - s.begin;

        float nlo[3], nhi[3], clo[3], chi[3];
        {
            const uint32_t t0 = order[s.begin];
            for (int a = 0; a < 3; ++a) {
                nlo[a] = lo[(size_t)t0 * 3 + a];
                nhi[a] = hi[(size_t)t0 * 3 + a];
                clo[a] = chi[a] = cen[(size_t)t0 * 3 + a];
            }
        }
        for (uint32_t i = s.begin + 1; i < s.end; ++i) {
            const uint32_t t = order[i];
            for (int a = 0; a < 3; ++a) {
                nlo[a] = fmin2(nlo[a], lo[(size_t)t * 3 + a]);
                nhi[a] = fmax2(nhi[a], hi[(size_t)t * 3 + a]);
                clo[a] = fmin2(clo[a], cen[(size_t)t * 3 + a]);
                chi[a] = fmax2(chi[a], cen[(size_t)t * 3 + a]);
            }
        }
        const bool leaf = count <= leaf_size;
        out.start.push_back(s.begin);
        out.nprims.push_back(count);
        out.right_offset.push_back(leaf ? 0u : 0xffffffffu);
        for (int a = 0; a < 3; ++a) out.bbox.push_back(nlo[a]);
        for (int a = 0; a < 3; ++a) out.bbox.push_back(nhi[a]);
        node_depth.push_back(s.depth);
        if (s.depth > out.max_depth) out.max_depth = s.depth;
        if (s.is_right) out.right_offset[s.parent] = me - s.parent;  // bvh.cpp:233-235
        if (leaf) {
            out.n_leaves++;
            continue;
        }

        // BBox::maxDimension on the centroid bounds (bbox.cpp:41-46)
        const float ex = chi[0] - clo[0], ey = chi[1] - clo[1], ez = chi[2] - clo[2];
        int axis = 0;
        if (ey > ex) axis = 1;
        if (ez > ey) axis = 2;
        const float split = .5f * (clo[axis] + chi[axis]);
        uint32_t mid = s.begin;
        for (uint32_t i = s.begin; i < s.end; ++i) {
            if (cen[(size_t)order[i] * 3 + axis] < split) {
                std::swap(order[i], order[mid]);
                ++mid;
            }
        }
        if (mid == s.begin || mid == s.end) mid = s.begin + (s.end - s.begin) / 2;
        work.push_back({mid, s.end, me, s.depth + 1, true});     // popped second
        work.push_back({s.begin, mid, me, s.depth + 1, false});  // popped first: index me+1
    }

    return flatten(pos, nrm, uv, ntris, out, err);
}


bool check_bvh_input(const float *pos, const float *nrm, uint32_t ntris, uint32_t &leaf_size, std::string &err) {
    return check_input(pos, nrm, ntris, leaf_size, err);
}
bool flatten_bvh(const float *pos, const float *nrm, const float *uv, uint32_t ntris, HostBvh &out, std::string &err) {
    return flatten(pos, nrm, uv, ntris, out, err);
}

// ---------------------------------------------------------------------------
// Quality builder: binned surface-area heuristic, 16 bins per
// axis, same flat layout.  NOT the reference's topology: the nearest hit is the
// same triangle test arithmetic, but traversal order — hence which of two
// triangles wins an exact distance tie, and what the `near > t` pruning skips —
// differs from BVH::build's tree.  For throughput runs, not for ID-parity runs.
// ---------------------------------------------------------------------------
bool build_bvh_sah(const float *pos, const float *nrm, const float *uv, uint32_t ntris, uint32_t leaf_size,
                   HostBvh &out, std::string &err) {
    if (!check_input(pos, nrm, ntris, leaf_size, err)) return false;
    std::vector<float> lo, hi, cen;
    triangle_bounds(pos, ntris, lo, hi, cen);
    std::vector<uint32_t> &order = out.prim_order;
    order.resize(ntris);
    for (uint32_t t = 0; t < ntris; ++t) order[t] = t;
    out.start.clear(), out.nprims.clear(), out.right_offset.clear(), out.bbox.clear();
    out.n_leaves = 0;
    out.max_depth = 0;
    constexpr int kBins = 16;
    auto area = [](const float *l, const float *h) {
        const float ex = h[0] - l[0], ey = h[1] - l[1], ez = h[2] - l[2];
        return 2.f * (ex * ey + ey * ez + ez * ex);
    };
    std::vector<Span> work;
    work.push_back({0u, ntris, 0u, 0u, false});
    std::vector<uint32_t> scratch;
    while (!work.empty()) {
        const Span s = work.back();
        work.pop_back();
        const uint32_t me = (uint32_t)out.start.size();
        const uint32_t count = s.end - s.begin;
        float nlo[3] = {INFINITY, INFINITY, INFINITY}, nhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = s.begin; i < s.end; ++i) {
            const uint32_t t = order[i];
            for (int a = 0; a < 3; ++a) {
                nlo[a] = fmin2(nlo[a], lo[(size_t)t * 3 + a]);
                nhi[a] = fmax2(nhi[a], hi[(size_t)t * 3 + a]);
                clo[a] = fmin2(clo[a], cen[(size_t)t * 3 + a]);
                chi[a] = fmax2(chi[a], cen[(size_t)t * 3 + a]);
            }
        }
        const bool leaf = count <= leaf_size;
        out.start.push_back(s.begin);
        out.nprims.push_back(count);
        out.right_offset.push_back(leaf ? 0u : 0xffffffffu);
        for (int a = 0; a < 3; ++a) out.bbox.push_back(nlo[a]);
        for (int a = 0; a < 3; ++a) out.bbox.push_back(nhi[a]);
        if (s.depth > out.max_depth) out.max_depth = s.depth;
        if (s.is_right) out.right_offset[s.parent] = me - s.parent;
        if (leaf) {
            out.n_leaves++;
            continue;
        }
        // best binned split over the three axes; deep subtrees fall back to the index median so
        // that the tree stays within the 64-entry traversal stack
        uint32_t mid = s.begin + count / 2;
        int best_axis = -1, best_bin = -1;
        float best_cost = INFINITY;
        if (s.depth < 40) {
            for (int axis = 0; axis < 3; ++axis) {
                const float ext = chi[axis] - clo[axis];
                if (!(ext > 0.f)) continue;
                uint32_t cnt[kBins] = {0};
                float blo[kBins][3], bhi[kBins][3];
                for (int b = 0; b < kBins; ++b)
                    for (int a = 0; a < 3; ++a) blo[b][a] = INFINITY, bhi[b][a] = -INFINITY;
                const float scale = (float)kBins / ext;
                for (uint32_t i = s.begin; i < s.end; ++i) {
                    const uint32_t t = order[i];
                    int b = (int)((cen[(size_t)t * 3 + axis] - clo[axis]) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    cnt[b]++;
                    for (int a = 0; a < 3; ++a) {
                        blo[b][a] = fmin2(blo[b][a], lo[(size_t)t * 3 + a]);
                        bhi[b][a] = fmax2(bhi[b][a], hi[(size_t)t * 3 + a]);
                    }
                }
                float rarea[kBins];
                uint32_t rcnt[kBins];
                float al[3] = {INFINITY, INFINITY, INFINITY}, ah[3] = {-INFINITY, -INFINITY, -INFINITY};
                uint32_t c = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    for (int a = 0; a < 3; ++a) al[a] = fmin2(al[a], blo[b][a]), ah[a] = fmax2(ah[a], bhi[b][a]);
                    c += cnt[b];
                    rarea[b] = c ? area(al, ah) : 0.f;
                    rcnt[b] = c;
                }
                for (int a = 0; a < 3; ++a) al[a] = INFINITY, ah[a] = -INFINITY;
                c = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    for (int a = 0; a < 3; ++a) al[a] = fmin2(al[a], blo[b][a]), ah[a] = fmax2(ah[a], bhi[b][a]);
                    c += cnt[b];
                    if (c == 0 || rcnt[b + 1] == 0) continue;
                    const float cost = area(al, ah) * (float)c + rarea[b + 1] * (float)rcnt[b + 1];
                    if (cost < best_cost) best_cost = cost, best_axis = axis, best_bin = b;
                }
            }
        }
        if (best_axis >= 0) {
            const float ext = chi[best_axis] - clo[best_axis];
            const float scale = (float)kBins / ext;
            // stable partition by bin index <= best_bin
            scratch.clear();
            uint32_t w = s.begin;
            for (uint32_t i = s.begin; i < s.end; ++i) {
                const uint32_t t = order[i];
                int b = (int)((cen[(size_t)t * 3 + best_axis] - clo[best_axis]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                if (b <= best_bin) order[w++] = t;
                else scratch.push_back(t);
            }
            mid = w;
            for (uint32_t t : scratch) order[w++] = t;
            if (mid == s.begin || mid == s.end) mid = s.begin + count / 2;
        } else {
            // index median along the widest centroid axis
            int axis = 0;
            if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
            if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
            std::nth_element(order.begin() + s.begin, order.begin() + mid, order.begin() + s.end,
                             [&](uint32_t x, uint32_t y) {
                                 const float cx = cen[(size_t)x * 3 + axis], cy = cen[(size_t)y * 3 + axis];
                                 return cx < cy || (cx == cy && x < y);
                             });
        }
        work.push_back({mid, s.end, me, s.depth + 1, true});
        work.push_back({s.begin, mid, me, s.depth + 1, false});
    }
    return flatten(pos, nrm, uv, ntris, out, err);
}

}  // namespace vmx
