// vmx_device.h — structures shared by the host orchestration (vmx_api.cpp) and
// the gfx950 kernels (vmx_kernels.hip).  Plain C++; no HIP types beyond float4.
#pragma once
#include <stdint.h>

namespace vmx {

// ---- BVH child reference ---------------------------------------------------
// bit 31 set  : leaf,  bits 0..25 = first triangle (leaf order), bits 26..30 = count
// bit 31 clear: index of a 2-wide inner record
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kLeafStartMask = 0x03FFFFFFu;
constexpr uint32_t kLeafCountShift = 26;
constexpr uint32_t kMaxLeafSize = 31;
constexpr uint32_t kMaxTris = kLeafStartMask + 1;
constexpr uint32_t kMaxSpheres = 16;
constexpr uint32_t kMaxStack = 64;  // bvh.cpp:54

// One 2-wide inner record = 64 bytes = 4 x float4, read by one lane as four
// 16-byte loads from one base address:
//   q0 = (L.min.x, L.min.y, L.min.z, L.max.x)
//   q1 = (L.max.y, L.max.z, R.min.x, R.min.y)
//   q2 = (R.min.z, R.max.x, R.max.y, R.max.z)
//   q3 = (bits(left ref), bits(right ref), 0, 0)
// L is the reference's node ni+1, R is node ni+rightOffset (bvh.cpp:99-100).
struct InnerRecord {
    float lmin[3];
    float lmax[3];
    float rmin[3];
    float rmax[3];
    uint32_t left, right, pad0, pad1;
};
static_assert(sizeof(InnerRecord) == 64, "inner record must be 64 bytes");

// Intersection record, leaf order, 48 bytes = 3 x float4:
//   q0 = (v0.x, v0.y, v0.z, e1.x)  q1 = (e1.y, e1.z, e2.x, e2.y)  q2 = (e2.z, bits(id), 0, 0)
// e1 = v1 - v0, e2 = v2 - v0 are the values triangle.cpp:12-13 recomputes per test.
struct TriRecord {
    float v0[3];
    float e1[3];
    float e2[3];
    uint32_t id;
    uint32_t pad[2];
};
static_assert(sizeof(TriRecord) == 48, "tri record must be 48 bytes");

// Shading attributes, leaf order, 64 bytes = 4 x float4 (triangle.cpp:81-82 inputs)
struct AttrRecord {
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    float pad;
};
static_assert(sizeof(AttrRecord) == 64, "attr record must be 64 bytes");

struct SphereDev {
    float cx, cy, cz, rad;
    float rad2;  // float product rad*rad (meshEngine.cpp:188)
    float colr, colg, colb;
    float ncx, ncy, ncz, nsign;
    uint32_t flags, pad0, pad1, pad2;
};
static_assert(sizeof(SphereDev) == 64, "sphere record must be 64 bytes");

struct SceneDev {
    const void *inner;    // InnerRecord[n_inner]
    const void *tris;     // TriRecord[ntris]
    const void *attrs;    // AttrRecord[ntris]
    const SphereDev *spheres;
    uint32_t root_ref;
    uint32_t nspheres;
    uint32_t stack_entries;  // per-lane LDS stack entries (max tree depth + 2)
    uint32_t ntris;
    // boundTextures[0] (meshEngine.h:62): float[h][w][c] as bindTexture reads it (meshEngine.cpp:74-93)
    const float *tex;
    uint32_t tex_w, tex_h, tex_c;
    uint32_t tri_off;  // byte offset of `tris` from `inner` (one allocation: unified record fetch of k_trace_w<1>)
    // boundTextures[1]: BruteForceTracer's albedo (integrators.cpp:141-147)
    const float *tex1;
    uint32_t tex1_w, tex1_h, tex1_c;
    uint32_t emit_prefix;  // index of the last VMX_SPHERE_EMIT sphere + 1 (0: none): hitColour is decided by spheres [0, emit_prefix)
};

// per-stage device counters (one set for depth-0 steps, one for bounce steps)
struct StageCounters {
    unsigned long long rays, inner_visits, tri_tests, tri_hits, continued;
};
struct DevCounters {
    StageCounters stage[2];
    unsigned long long samples, discarded, pixels_done, overflow;
};

// Division of a 32-bit index by a launch constant (samples per pixel of a pass, image width, rows per stripe) as a
// multiply-high and two shifts (Granlund-Montgomery, the branch-free form): exact for every 32-bit numerator.  A plain
// n / d with a run-time d is ~25 VALU instructions, and k_shade<0> did three per camera path.
//   l = ceil(log2 d)   m = floor(2^32 (2^l - d) / d) + 1   q = (t + ((n - t) >> min(l, 1))) >> max(l - 1, 0),  t = mulhi(m, n)
struct FastDiv {
    uint32_t m, sh1, sh2, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
    if (d == 0) d = 1;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    FastDiv f;
    f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1ull);
    f.sh1 = l < 1 ? l : 1u;
    f.sh2 = l > 0 ? l - 1 : 0u;
    f.d = d;
    return f;
}

// Camera/frame constants for ray generation (pathtracer.cpp:216-221, 251-280)
struct FrameDev {
    float m[9];  // column-major 3x3 camera matrix
    float px, py, pz;
    float film_dist, sensor_x, sensor_y;
    uint32_t width, height;       // full image
    uint32_t spp, quarter;        // uSamplesPerPixel, spp/4
    uint32_t kmax;                // 4*quarter
    uint32_t nmin;                // floor(sqrt(spp)): early stop needs n > sqrt(spp)
    uint32_t early_stop;
    uint32_t lead;                // per pass: > 0 in the first early-stop pass — paths j < lead are samples j,
                                  // paths j >= lead the first samples of the following strata (sample_index)
    float r2scale;                // 10 (parity) or 1 (corrected)
    uint32_t libm_double;         // VMX_SAMPLING_LIBM_DOUBLE: cos/sin(float r1) of pathtracer.cpp:162 as C's double functions
    uint32_t elide_dead;          // VMX_SAMPLING_ELIDE_DEAD: camera paths with provably zero radiance are not traced
    uint32_t bounce_bits;         // k_shade stores step_bits of the next step with every bounce ray (k_trace_w<1, .., SORT>)
    uint32_t camera_bits;         // k_raygen writes step_bits of the first step with every camera ray (read by k_trace_w<0, .., SORT>
                                  // and k_shade_ends<0>; a frame whose steps are all shaded in full does not draw them)
    uint32_t local_rows;          // rows owned by this rank
    uint32_t stripe_rows, rank, world;
    uint64_t seed;
    double inv_width, inv_height;  // RN(1 / width), RN(1 / height) by IEEE division on the host (div_by_count)
    FastDiv div_width, div_stripe;  // index / width, row / stripe_rows (global_pixel)
};

// Path state across a bounce boundary: 6 planes of 16 bytes, plane p of slot s
// at planes[p * capacity + s] (SoA of float4: every wave access is 1 KiB contiguous).
//   P0 = (o.x, o.y, o.z, d.x)   P1 = (d.y, d.z, thr.r, thr.g)
//   P2 = (thr.b, acc.r, acc.g, acc.b)   P3 = (acc.w, bits(depth), bits(dest), 0)
//   P4 = rng s0,s1   P5 = rng s2,s3
constexpr uint32_t kPathPlanes = 6;
constexpr uint32_t kPathBytes = kPathPlanes * 16;

}  // namespace vmx
