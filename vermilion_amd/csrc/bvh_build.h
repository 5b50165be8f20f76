// bvh_build.h — host-side BVH construction with the reference's topology
// (core/accelerators/bvh.cpp:179-279) and its flattening into the 2-wide
// device records of vmx_device.h.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "vmx_device.h"

namespace vmx {

struct HostBvh {
    // reference flat layout (bvh.h:11-14), DFS pre-order, left child = i+1
    std::vector<uint32_t> start, nprims, right_offset;
    std::vector<float> bbox;          // [n_nodes*6] min,max
    std::vector<uint32_t> prim_order; // final build_prims permutation: leaf slot -> triangle id
    uint32_t n_leaves = 0;
    uint32_t max_depth = 0;
    // device layout
    std::vector<InnerRecord> inner;
    std::vector<TriRecord> tris;
    std::vector<AttrRecord> attrs;
    uint32_t root_ref = 0;
};

// pos/nrm: [ntris*9], uv: [ntris*6] or nullptr.  Returns false and sets `err`
// on invalid input (non-finite vertices, too many triangles, tree too deep).
bool build_bvh(const float *pos, const float *nrm, const float *uv, uint32_t ntris,
               uint32_t leaf_size, HostBvh &out, std::string &err);

// binned-SAH quality builder (not the reference's topology; see bvh_build.cpp)
bool build_bvh_sah(const float *pos, const float *nrm, const float *uv, uint32_t ntris,
                   uint32_t leaf_size, HostBvh &out, std::string &err);

// linear BVH built ENTIRELY on the GPU (lbvh_build.hip): Morton sort + Karras hierarchy + bottom-up fit +
// device-record emission.  The product buffers (`geom` = inner records then triangle records, `attrs`) are
// device memory handed to the scene; the hierarchy arrays stay in `arena` for lbvh_export_flat.
struct LbvhDevice {
    void *arena = nullptr;  // the hierarchy (kept for lbvh_export_flat); inputs and temporaries are freed by the build
    size_t arena_bytes = 0;
    void *geom = nullptr;   // InnerRecord[n_inner] then TriRecord[ntris] (+ 64 bytes of padding)
    size_t geom_bytes = 0;
    void *attrs = nullptr;  // AttrRecord[ntris]
    uint32_t ntris = 0, leaf_size = 4;
    uint32_t n_inner = 0;   // inner record slots = max(ntris - 1, 1): record i is Karras node i
    uint32_t tri_off = 0, root_ref = 0, height = 0;
    // hierarchy (device pointers into the arena): children (bit 31 = sorted primitive), key ranges, boxes,
    // sorted position -> triangle id
    uint32_t *left = nullptr, *right = nullptr, *first = nullptr, *last = nullptr, *ids = nullptr;
    float *leaf_box = nullptr, *node_box = nullptr;
};
bool build_bvh_lbvh_device(const float *pos, const float *nrm, const float *uv, uint32_t ntris, uint32_t leaf_size,
                           int device, LbvhDevice &out, std::string &err);
// the same product from parallel locally-ordered clustering (PLOC): a quality tree, also built entirely on the GPU
bool build_bvh_ploc_device(const float *pos, const float *nrm, const float *uv, uint32_t ntris, uint32_t leaf_size,
                           int device, LbvhDevice &out, std::string &err);
// reference flat layout (start / nprims / right_offset / bbox / prim_order, n_leaves, max_depth) of such a tree
bool lbvh_export_flat(const LbvhDevice &d, int device, HostBvh &out, std::string &err);
void lbvh_release(LbvhDevice &d);

// shared by the builders: input validation; flat tree (start/nprims/right_offset/bbox/prim_order) -> device records
bool check_bvh_input(const float *pos, const float *nrm, uint32_t ntris, uint32_t &leaf_size, std::string &err);
bool flatten_bvh(const float *pos, const float *nrm, const float *uv, uint32_t ntris, HostBvh &out, std::string &err);

}  // namespace vmx
