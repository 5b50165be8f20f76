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

bool build_bvh(const float *pos, const float *nrm, const float *uv, uint32_t ntris,
               uint32_t leaf_size, HostBvh &out, std::string &err) {
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

    // per-triangle bounds and centroid, computed once (the reference recomputes
    // the same values on every use)
    std::vector<float> lo((size_t)ntris * 3), hi((size_t)ntris * 3), cen((size_t)ntris * 3);
    for (uint32_t t = 0; t < ntris; ++t) {
        const float *p = pos + (size_t)t * 9;
        for (int a = 0; a < 3; ++a) {
            lo[(size_t)t * 3 + a] = fmin2(fmin2(p[a], p[3 + a]), p[6 + a]);
            hi[(size_t)t * 3 + a] = fmax2(fmax2(p[a], p[3 + a]), p[6 + a]);
            cen[(size_t)t * 3 + a] = ((p[a] + p[3 + a]) + p[6 + a]) * 0.333f;
        }
    }

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

}  // namespace vmx
