// vmx_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the path-tracing hot path.
//
// Compiled with -ffp-contract=off: every float operation rounds once, in the
// operation order of the reference's GLM expressions, so that triangle IDs,
// distances and whole frames are bit-identical to the CPU oracle.  Divide and
// sqrt are hipcc's correctly rounded forms (its default).
//
// Kernels
//   k_trace        BVH::getIntersection for explicit rays          (bvh.cpp:47-145)
//   k_raycast      MeshEngine::RayCast for explicit rays           (meshEngine.cpp:239-509)
//   k_raygen       camera rays of a pass, each with the step_bits of its first Radiance step (pathtracer.cpp:251-280);
//                  <1> + k_raygen_live: only the live ones (VMX_SAMPLING_ELIDE_DEAD, with path_compact.hip)
//   k_trace_w      persistent BVH traversal of camera / bounce rays (bvh.cpp:47-145); <.., SORT>: settles the finished rays
//                  whose step ends by its draws, hands the others on as records (DESIGN_HISTORY.md 5.1); k_trace_q: counting form
//   k_shade        RayCast tail + one Radiance step, id compaction  (meshEngine.cpp:365-508, pathtracer.cpp:36-196);
//                  k_shade_ends: the two-phase form's first phase
//   k_paths        traversal + shading fused, paths kept to their end (small passes, the tail of a pass)
//   k_resolve      per-pixel accumulation, early stop, pixel write (pathtracer.cpp:282-324)
//   k_bruteforce*  BruteForceTracer::Render                        (integrators.cpp:9-186)
// (the first-generation kernels k_primary / k_bounce live in vmx_kernels_ab.inc: A/B library only)
//
// Execution model: persistent blocks stride over block-sized work items; the
// item -> image-region map is XCD-aware (items b and b+8 run on one XCD and
// are given neighbouring pixel tiles, so an XCD's 4 MiB L2 holds the part of
// the BVH its rays walk).  Each lane owns one ray/path; its BVH stack lives in
// LDS, lane-strided (entry e of lane l at [e*64 + l], 8 bytes: node ref +
// entry distance), which is conflict-free for ds_read/write_b64 at any mix of
// per-lane stack depths.  Surviving paths are compacted between bounces with a
// wave ballot + prefix popcount and one atomic per wave on one of 16 sub-queue
// tails.  No MFMA: there is no dense contraction in this workload.
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include <stdint.h>

#include "vmx_kernels.h"
#include <type_traits>

#ifndef VMX_CAM_BATCH
#define VMX_CAM_BATCH 16  // steps of a camera-ray wave between two refill tests (a descent counts as 5)
#endif
#ifndef VMX_TRACE_WAVES_PER_SIMD
#define VMX_TRACE_WAVES_PER_SIMD 7  // register budget of the trace kernel: 512 / 7 -> 72 VGPRs
#endif


namespace vmx {
namespace {

constexpr float kInf = __builtin_huge_valf();

struct Cnt {
    uint32_t inner, tris;
};

// wave-uniform tallies, flushed once per wave at kernel exit
struct Tally {
    uint32_t rays[2], hits[2], cont[2];
};

__device__ __forceinline__ uint32_t lane_index() { return __lane_id(); }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------
// RNG: xoshiro256** keyed by (seed, pixel, sample)  — DESIGN.md §3
// ---------------------------------------------------------------------------
struct Rng {
    uint64_t s0, s1, s2, s3;
};

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Stream of sample k of pixel p (the reference seeds a thread-local mt19937_64 from std::random_device,
// pathtracer.cpp:231, and is not reproducible: the stream definition is this build's own, shared with the oracle).
// Round 4: the pixel half of the key is two mix64 per PIXEL — a wave of k_raygen / k_shade<0> is 64 samples of one pixel,
// so it runs once per wave on the scalar unit — and the sample half costs three 64-bit multiplies (round 3: ten per
// sample, twice per camera path):
//   a = mix64(seed ^ (p << 32))   b = mix64(a + golden)                       per pixel
//   s0 = mix64(a + k)   t = (s0 ^ b) * M3   s1 = t ^ (t >> 32)   s2 = rotl(s0, 24) ^ b   s3 = rotl(s1, 37) ^ a
struct PixelKey {
    uint64_t a, b;
};
__device__ __forceinline__ PixelKey rng_pixel_key(uint64_t seed, uint32_t pixel) {
    PixelKey pk;
    pk.a = mix64(seed ^ ((uint64_t)pixel << 32));
    pk.b = mix64(pk.a + 0x9E3779B97F4A7C15ull);
    return pk;
}
__device__ __forceinline__ void rng_init_keyed(Rng &r, PixelKey pk, uint32_t k) {
    r.s0 = mix64(pk.a + (uint64_t)k);
    const uint64_t t = (r.s0 ^ pk.b) * 0xD6E8FEB86659FD93ull;
    r.s1 = t ^ (t >> 32);
    r.s2 = rotl64(r.s0, 24) ^ pk.b;
    r.s3 = rotl64(r.s1, 37) ^ pk.a;
}
__device__ __forceinline__ void rng_init(Rng &r, uint64_t seed, uint32_t pixel, uint32_t k) {
    rng_init_keyed(r, rng_pixel_key(seed, pixel), k);
}

__device__ __forceinline__ uint64_t rng_next(Rng &r) {
    const uint64_t out = rotl64(r.s1 * 5, 7) * 9;
    const uint64_t t = r.s1 << 17;
    r.s2 ^= r.s0;
    r.s3 ^= r.s1;
    r.s1 ^= r.s2;
    r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = rotl64(r.s3, 45);
    return out;
}

// stands in for uniform_real_distribution<double>(0,1) (pathtracer.cpp:23)
__device__ __forceinline__ double rng_u01(Rng &r) {
    return __longlong_as_double((long long)(0x3FF0000000000000ull | (rng_next(r) >> 12))) - 1.0;
}
// stands in for uniform_real_distribution<float>(0,0.5) (pathtracer.cpp:230)
__device__ __forceinline__ float rng_jitter(Rng &r) {
    return (__uint_as_float(0x3F800000u | (uint32_t)(rng_next(r) >> 41)) - 1.0f) * 0.5f;
}

// ---------------------------------------------------------------------------
// small vector helpers in GLM's operation order
// ---------------------------------------------------------------------------
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx + ay * by) + az * bz;  // glm::dot: tmp.x + tmp.y + tmp.z
}
// glm::normalize = v * (1 / sqrt(dot(v,v)))
__device__ __forceinline__ void normalize3(float &x, float &y, float &z) {
    const float s = 1.0f / sqrtf(dot3(x, y, z, x, y, z));
    x = x * s;
    y = y * s;
    z = z * s;
}
// glm::cross(a, b)
__device__ __forceinline__ void cross3(float ax, float ay, float az, float bx, float by, float bz,
                                       float &cx, float &cy, float &cz) {
    cx = ay * bz - by * az;
    cy = az * bx - bz * ax;
    cz = ax * by - bx * ay;
}
__device__ __forceinline__ bool finite3(float x, float y, float z) {
    return (fabsf(x) < kInf) && (fabsf(y) < kInf) && (fabsf(z) < kInf);
}

// ---------------------------------------------------------------------------
// BBox::intersect (bbox.cpp:70-83)
// ---------------------------------------------------------------------------
// Fast form: v_min/v_max.  Equal to the reference's compare-select form
// whenever no slab product is NaN, which `exact == false` guarantees (finite
// origin, finite non-zero reciprocal direction).
// The min/max are issued as plain v_min/v_max/v_max3/v_min3 (inline asm): without it hipcc adds a
// canonicalising v_max(x,x) in front of every fminf/fmaxf operand that comes out of a packed multiply.
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ bool box_fast(float lx, float ly, float lz, float hx, float hy, float hz,
                                         float ox, float oy, float oz, float ix, float iy, float iz,
                                         float &tnear) {
    const float t0x = (lx - ox) * ix, t0y = (ly - oy) * iy, t0z = (lz - oz) * iz;
    const float t1x = (hx - ox) * ix, t1y = (hy - oy) * iy, t1z = (hz - oz) * iz;
    const float n = vmax3(vmin(t0x, t1x), vmin(t0y, t1y), vmin(t0z, t1z));
    const float f = vmin3(vmax(t0x, t1x), vmax(t0y, t1y), vmax(t0z, t1z));
    tnear = n;
    return n <= f;
}
// Same with (lo - o), (hi - o) already formed (k_camera_tables)
__device__ __forceinline__ bool box_fast_rel(float lx, float ly, float lz, float hx, float hy, float hz, float ix,
                                             float iy, float iz, float &tnear) {
    const float t0x = lx * ix, t0y = ly * iy, t0z = lz * iz;
    const float t1x = hx * ix, t1y = hy * iy, t1z = hz * iz;
    const float n = vmax3(vmin(t0x, t1x), vmin(t0y, t1y), vmin(t0z, t1z));
    const float f = vmin3(vmax(t0x, t1x), vmax(t0y, t1y), vmax(t0z, t1z));
    tnear = n;
    return n <= f;
}

// Exact form: glm::min/max = (b<a)?b:a / (a<b)?b:a, std::max/min likewise, NaN and all.
__device__ __forceinline__ float sel_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float sel_max(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ bool box_exact(float lx, float ly, float lz, float hx, float hy, float hz,
                                       float ox, float oy, float oz, float ix, float iy, float iz,
                                       float &tnear) {
    const float t0x = (lx - ox) * ix, t0y = (ly - oy) * iy, t0z = (lz - oz) * iz;
    const float t1x = (hx - ox) * ix, t1y = (hy - oy) * iy, t1z = (hz - oz) * iz;
    const float sx = sel_min(t0x, t1x), sy = sel_min(t0y, t1y), sz = sel_min(t0z, t1z);
    const float bx = sel_max(t0x, t1x), by = sel_max(t0y, t1y), bz = sel_max(t0z, t1z);
    const float n = sel_max(sel_max(sx, sy), sz);
    const float f = sel_min(sel_min(bx, by), bz);
    tnear = n;
    return n <= f;
}

__device__ __forceinline__ bool box_exact_rel(float lx, float ly, float lz, float hx, float hy, float hz, float ix,
                                              float iy, float iz, float &tnear) {
    const float t0x = lx * ix, t0y = ly * iy, t0z = lz * iz;
    const float t1x = hx * ix, t1y = hy * iy, t1z = hz * iz;
    const float sx = sel_min(t0x, t1x), sy = sel_min(t0y, t1y), sz = sel_min(t0z, t1z);
    const float bx = sel_max(t0x, t1x), by = sel_max(t0y, t1y), bz = sel_max(t0z, t1z);
    const float n = sel_max(sel_max(sx, sy), sz);
    const float f = sel_min(sel_min(bx, by), bz);
    tnear = n;
    return n <= f;
}

// min/max network of one box from the six slab products (k_trace_w, k_paths): one asm block, so that no
// hazard nops land between the pieces; the exact form is the compare-select network of box_exact
__device__ __forceinline__ void box_net(float t0x, float t0y, float t0z, float t1x, float t1y, float t1z,
                                        float &tnear, float &tfar) {
    float n, f, s1, s2;
    asm("v_min_f32 %0, %4, %7\n\t"
        "v_min_f32 %2, %5, %8\n\t"
        "v_min_f32 %3, %6, %9\n\t"
        "v_max3_f32 %0, %0, %2, %3\n\t"
        "v_max_f32 %1, %4, %7\n\t"
        "v_max_f32 %2, %5, %8\n\t"
        "v_max_f32 %3, %6, %9\n\t"
        "v_min3_f32 %1, %1, %2, %3"
        : "=&v"(n), "=&v"(f), "=&v"(s1), "=&v"(s2)
        : "v"(t0x), "v"(t0y), "v"(t0z), "v"(t1x), "v"(t1y), "v"(t1z));
    tnear = n;
    tfar = f;
}
__device__ __forceinline__ void box_net_exact(float t0x, float t0y, float t0z, float t1x, float t1y, float t1z,
                                              float &tnear, float &tfar) {
    const float sx = sel_min(t0x, t1x), sy = sel_min(t0y, t1y), sz = sel_min(t0z, t1z);
    const float bx = sel_max(t0x, t1x), by = sel_max(t0y, t1y), bz = sel_max(t0z, t1z);
    tnear = sel_max(sel_max(sx, sy), sz);
    tfar = sel_min(sel_min(bx, by), bz);
}

// ---------------------------------------------------------------------------
// BVH::getIntersection, nearest hit (bvh.cpp:47-145) + Triangle::getIntersection
// (triangle.cpp:4-54).  `stk` points at this lane's column of the wave's LDS stack.
// ---------------------------------------------------------------------------
// (per-lane traversal stack helpers, defined with the persistent kernels below)
__device__ __forceinline__ void stack_push(uint2 *stk, uint2 *ovf, int lds_entries, int sp, uint2 e);
__device__ __forceinline__ uint2 stack_pop(const uint2 *stk, const uint2 *ovf, int lds_entries, int sp);

template <bool COUNT>
__device__ __forceinline__ void bvh_nearest(const SceneDev &sc, float ox, float oy, float oz, float dx,
                                            float dy, float dz, uint2 *stk, float &best_out,
                                            int &slot_out, Cnt &cnt, uint2 *ovf = nullptr,
                                            int lds_entries = 0x7FFFFFFF /* levels of `stk` in LDS; deeper ones in `ovf` */) {
    const float4 *__restrict__ inner = (const float4 *)sc.inner;
    const float4 *__restrict__ tris = (const float4 *)sc.tris;
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;  // Ray.h:10
    const bool exact = !(finite3(ix, iy, iz) && finite3(ox, oy, oz));

    float best = 999999999.f;  // bvh.cpp:48
    int slot = -1;
    int sp = 0;
    uint32_t cur = sc.root_ref;
    float cur_near = -9999999.f;  // bvh.cpp:59
    bool have = true;
    for (;;) {
        if (!have) {
            if (sp == 0) break;
            --sp;
            const uint2 e = stack_pop(stk, ovf, lds_entries, sp);
            cur = e.x;
            cur_near = __uint_as_float(e.y);
        }
        have = false;
        if (cur_near > best) continue;  // bvh.cpp:69
        if (cur & kLeafBit) {
            const uint32_t first = cur & kLeafStartMask;
            const uint32_t n = (cur >> kLeafCountShift) & 31u;
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t ti = (first + i) * 3;
                const float4 a = tris[ti], b = tris[ti + 1], c = tris[ti + 2];
                if (COUNT) cnt.tris++;
                const float e1x = a.w, e1y = b.x, e1z = b.y, e2x = b.z, e2y = b.w, e2z = c.x;
                float px, py, pz;
                cross3(dx, dy, dz, e2x, e2y, e2z, px, py, pz);
                const float det = dot3(e1x, e1y, e1z, px, py, pz);
                // (det < 1e-8 && det > -1e-8) in double  <=>  |det| <= float(1e-8)  (float(1e-8) < 1e-8)
                const bool parallel = fabsf(det) <= 9.99999993922529e-09f;
                const float inv_det = 1.0f / det;
                const float tx = ox - a.x, ty = oy - a.y, tz = oz - a.z;
                const float u = dot3(tx, ty, tz, px, py, pz) * inv_det;
                const bool u_out = (u < 0.0f) || (u > 1.0f);
                float qx, qy, qz;
                cross3(tx, ty, tz, e1x, e1y, e1z, qx, qy, qz);
                const float v = dot3(dx, dy, dz, qx, qy, qz) * inv_det;
                const bool v_out = (v < 0.0f) || (u + v > 1.0f);
                const float dist = dot3(e2x, e2y, e2z, qx, qy, qz) * inv_det;
                const bool hit = !parallel && !u_out && !v_out && (dist > 0.0f);
                if (hit && dist < best) {  // strict <: first tested wins ties (bvh.cpp:90)
                    best = dist;
                    slot = (int)(first + i);
                }
            }
        } else {
            const float4 q0 = inner[cur * 4], q1 = inner[cur * 4 + 1], q2 = inner[cur * 4 + 2],
                         q3 = inner[cur * 4 + 3];
            if (COUNT) cnt.inner++;
            float tn0, tn1;
            bool h0, h1;
            if (exact) {
                h0 = box_exact(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ox, oy, oz, ix, iy, iz, tn0);
                h1 = box_exact(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ox, oy, oz, ix, iy, iz, tn1);
            } else {
                h0 = box_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ox, oy, oz, ix, iy, iz, tn0);
                h1 = box_fast(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ox, oy, oz, ix, iy, iz, tn1);
            }
            const uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
            if (h0 && h1) {
                // left assumed closer; swap if the right child is strictly closer (bvh.cpp:106-114)
                const bool sw = tn1 < tn0;
                const uint32_t closer = sw ? rref : lref, other = sw ? lref : rref;
                const float nc = sw ? tn1 : tn0, no = sw ? tn0 : tn1;
                stack_push(stk, ovf, lds_entries, sp, make_uint2(other, __float_as_uint(no)));  // farther first (bvh.cpp:120)
                ++sp;
                cur = closer;
                cur_near = nc;
                have = true;
            } else if (h0) {
                cur = lref;
                cur_near = tn0;
                have = true;
            } else if (h1) {
                cur = rref;
                cur_near = tn1;
                have = true;
            }
        }
    }
    best_out = best;
    slot_out = slot;
}

// Triangle::getNormal (triangle.cpp:67-86) + normalize in the caller (meshEngine.cpp:369)
__device__ __forceinline__ void tri_shading_normal(const SceneDev &sc, int slot, float hx, float hy,
                                                   float hz, float &nx, float &ny, float &nz, float &uvx,
                                                   float &uvy) {
    const float4 *__restrict__ tris = (const float4 *)sc.tris;
    const float4 *__restrict__ attrs = (const float4 *)sc.attrs;
    const float4 a = tris[slot * 3], b = tris[slot * 3 + 1], c = tris[slot * 3 + 2];
    const float f0x = a.w, f0y = b.x, f0z = b.y, f1x = b.z, f1y = b.w, f1z = c.x;
    const float f2x = hx - a.x, f2y = hy - a.y, f2z = hz - a.z;
    const float d00 = dot3(f0x, f0y, f0z, f0x, f0y, f0z);
    const float d01 = dot3(f0x, f0y, f0z, f1x, f1y, f1z);
    const float d11 = dot3(f1x, f1y, f1z, f1x, f1y, f1z);
    const float d20 = dot3(f2x, f2y, f2z, f0x, f0y, f0z);
    const float d21 = dot3(f2x, f2y, f2z, f1x, f1y, f1z);
    const float denom = d00 * d11 - d01 * d01;
    const float w1 = (d11 * d20 - d01 * d21) / denom;
    const float w2 = (d00 * d21 - d01 * d20) / denom;
    const float w0 = 1.0f - w1 - w2;
    const float4 g0 = attrs[slot * 4], g1 = attrs[slot * 4 + 1], g2 = attrs[slot * 4 + 2],
                 g3 = attrs[slot * 4 + 3];
    // n0 = g0.xyz, n1 = (g0.w, g1.x, g1.y), n2 = (g1.z, g1.w, g2.x)
    const float inx = (g0.x * w0 + g0.w * w1) + g1.z * w2;
    const float iny = (g0.y * w0 + g1.x * w1) + g1.w * w2;
    const float inz = (g0.z * w0 + g1.y * w1) + g2.x * w2;
    // uv0 = (g2.y, g2.z), uv1 = (g2.w, g3.x), uv2 = (g3.y, g3.z)
    uvx = (g2.y * w0 + g2.w * w1) + g3.y * w2;
    uvy = (g2.z * w0 + g3.x * w1) + g3.z * w2;
    nx = -inx;
    ny = -iny;
    nz = -inz;
    normalize3(nx, ny, nz);
}

// sphereIntersect (meshEngine.cpp:182-194): float dots, float rad*rad, double discriminant.
// The caller only uses a result in (0, limit) (limit = RayCast's nearest distance so far), and the
// double-precision part (discriminant, square root, two roots: ~75 f64 instructions per sphere, 8
// spheres per RayCast) was more than half of the shading kernels' time.  Four single-precision
// tests on the same float dot products B = op.d, C = op.op and R2 = rad*rad the reference forms
// decide, with a margin far above every rounding error involved (2^-20 relative against 2^-24 float
// and 2^-53 double steps), that the double-precision result cannot matter:
//   miss   : B^2 < (C - R2) - tol                 =>  det < 0, the reference returns 0
//   behind : B < 0 and (C - R2) > tol             =>  det < B^2 although (B^2 - C) + R2 rounds at the magnitude
//            of C (2^-53 C, ~0.3 for the wall spheres), so both roots are <= 1e-4 (|B| < 1e11): returns 0
//   far    : B > limit and (C - R2) - limit (2B - limit) > tol  =>  det < (B - limit)^2 (1 - 2^-40),
//            so the smaller root rounds to >= limit and the caller ignores it
//   inside : (C - R2) < -tol (the origin is inside the sphere: the smaller root is negative, the result is the
//            larger one) and (C - R2) - limit (2B - limit) < -tol  =>  limit lies between the roots, the larger
//            root is > limit and the caller ignores it.  This is the common case of the reference's room: every
//            ray starts inside four of its six 5e7-radius wall spheres and mostly ends on a triangle first
// Only lanes that pass none of them need the exact evaluation; a wave runs it if any lane does (the
// 64 camera rays of a wave belong to one pixel and mostly agree).
// (op = centre - origin and C = op.op are passed in: for camera rays they are the same for every path
// of a frame and come precomputed, by the same float operations, from LDS)
__device__ __forceinline__ float sphere_hit_op(float opx, float opy, float opz, float C, float R2, float dx, float dy,
                                               float dz, float limit) {
    const float B = dot3(opx, opy, opz, dx, dy, dz);
    const float X = C - R2, BB = B * B;
    constexpr float kRel = 9.5367431640625e-07f;  // 2^-20
    const float tol_m = kRel * (C + R2 + BB);
    const float tol_f = kRel * (C + R2 + BB + limit * (2.f * fabsf(B) + limit));
    const bool miss = BB < X - tol_m;
    const bool behind = B < 0.f && B > -1e11f && X > tol_m;
    const float Y = X - limit * (2.f * B - limit);  // limit^2 - 2 B limit + X: negative iff limit lies between the roots
    const bool far = B > limit * (1.f + kRel) && Y > tol_f;
    const bool inside_far = X < -tol_m && Y < -tol_f;
    const bool need = !(miss || behind || far || inside_far);
    float th = 0.f;
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
        const double b = (double)B;
        double det = b * b - (double)C + (double)R2;
        float v = 0.f;
        if (!(det < 0)) {
            det = sqrt(det);
            double t = b - det;
            if (t > 1e-4) v = (float)t;
            else {
                t = b + det;
                if (t > 1e-4) v = (float)t;
            }
        }
        th = need ? v : 0.f;
    }
    return th;
}
// sphere_hit_op(..., limit = infinity) > 0, without the two tests that need a finite limit (they never fire there)
__device__ __forceinline__ bool sphere_in_reach(float opx, float opy, float opz, float C, float R2, float dx, float dy, float dz) {
    const float B = dot3(opx, opy, opz, dx, dy, dz);
    const float X = C - R2, BB = B * B;
    constexpr float kRel = 9.5367431640625e-07f;  // 2^-20
    const float tol_m = kRel * (C + R2 + BB);
    const bool miss = BB < X - tol_m;
    const bool behind = B < 0.f && B > -1e11f && X > tol_m;
    const bool need = !(miss || behind);
    bool reach = false;
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
        const double b = (double)B;
        double det = b * b - (double)C + (double)R2;
        float v = 0.f;
        if (!(det < 0)) {
            det = sqrt(det);
            double t = b - det;
            if (t > 1e-4) v = (float)t;
            else {
                t = b + det;
                if (t > 1e-4) v = (float)t;
            }
        }
        reach = need && v > 0.f;
    }
    return reach;
}
__device__ __forceinline__ float sphere_hit(float ox, float oy, float oz, float dx, float dy, float dz,
                                            float4 geom /* centre, rad*rad */, float limit) {
    const float opx = geom.x - ox, opy = geom.y - oy, opz = geom.z - oz;
    return sphere_hit_op(opx, opy, opz, dot3(opx, opy, opz, opx, opy, opz), geom.w, dx, dy, dz, limit);
}

struct CastResult {
    float nearest;     // INFINITY on a miss
    float nx, ny, nz;  // pHitNormal
    float cr, cg, cb;  // pHitColour
    float uvx, uvy;
    float tri_t;
    int slot;          // leaf-order slot of the BVH hit, -1 if none
    bool material;
};

// MeshEngine::RayCast after the BVH query (meshEngine.cpp:365-508): triangle normal, sphere table
// `geom`: optional copy of the spheres' (centre, rad*rad) in LDS (k_shade) — every path tests every
// sphere, the rest of a sphere's record is read only when it becomes the nearest hit
constexpr uint32_t kLdsSpheres = 16;
static_assert(kLdsSpheres >= kMaxSpheres, "every sphere of a scene has an LDS slot");
// LDS_GEOM / CAM_OP: compile-time, so that the table reads are plain LDS reads (a pointer chosen at run time
// between LDS and global memory makes them flat loads); the API admits at most kLdsSpheres spheres
template <bool LDS_GEOM = false, bool CAM_OP = false>
__device__ __forceinline__ void cast_finish(const SceneDev &sc, float ox, float oy, float oz, float dx,
                                            float dy, float dz, float best, int slot, CastResult &r,
                                            const float4 *geom = nullptr, const float4 *cam_op = nullptr) {
    r.nearest = kInf;
    r.nx = r.ny = r.nz = 0.f;
    r.cr = r.cg = r.cb = 0.f;
    r.uvx = r.uvy = 0.f;
    r.tri_t = best;
    r.slot = slot;
    r.material = false;
    if (slot >= 0) {
        r.nearest = best;
        const float hx = ox + dx * best, hy = oy + dy * best, hz = oz + dz * best;  // bvh.cpp:140
        tri_shading_normal(sc, slot, hx, hy, hz, r.nx, r.ny, r.nz, r.uvx, r.uvy);
        r.material = true;  // hitMeshIndex = 0 (meshEngine.cpp:370), never reset
    }
    const uint32_t ns = sc.nspheres;
    for (uint32_t i = 0; i < ns; ++i) {
        float4 g;
        if (LDS_GEOM) {
            g = geom[i];
        } else {
            const SphereDev &q = sc.spheres[i];
            g = make_float4(q.cx, q.cy, q.cz, q.rad2);
        }
        float th;
        if (CAM_OP) {  // (centre - camera, its squared length): camera rays only
            const float4 q = cam_op[i];
            th = sphere_hit_op(q.x, q.y, q.z, q.w, g.w, dx, dy, dz, r.nearest);
        } else {
            th = sphere_hit(ox, oy, oz, dx, dy, dz, g, r.nearest);
        }
        if (th > 0.f && th < r.nearest) {
            const SphereDev s = sc.spheres[i];
            r.nearest = th;
            if (s.flags & 1u) {
                r.cr = s.colr;
                r.cg = s.colg;
                r.cb = s.colb;
            }
            float px = ox + (dx * th) - s.ncx, py = oy + (dy * th) - s.ncy, pz = oz + (dz * th) - s.ncz;
            normalize3(px, py, pz);
            r.nx = px * s.nsign;
            r.ny = py * s.nsign;
            r.nz = pz * s.nsign;
        }
    }
}

// MeshEngine::RayCast (meshEngine.cpp:239-509)
template <bool COUNT>
__device__ __forceinline__ void ray_cast(const SceneDev &sc, float ox, float oy, float oz, float dx,
                                         float dy, float dz, uint2 *stk, CastResult &r, Cnt &cnt, uint2 *ovf = nullptr,
                                         int lds_entries = 0x7FFFFFFF) {
    float best;
    int slot;
    bvh_nearest<COUNT>(sc, ox, oy, oz, dx, dy, dz, stk, best, slot, cnt, ovf, lds_entries);
    cast_finish(sc, ox, oy, oz, dx, dy, dz, best, slot, r);
}

// ---------------------------------------------------------------------------
// one iteration of Radiance's bounce loop (pathtracer.cpp:34-197)
// ---------------------------------------------------------------------------
struct Path {
    float ox, oy, oz, dx, dy, dz;
    float tr, tg, tb, tw;  // accumRadiance; only maintained by the textured kernels (w: the product of the samples' fourth
                           // components — it only ever reaches accumColour.w as tw * 0, i.e. as a NaN when it is not finite)
    float ar, ag, ab, aw;  // accumColour
    uint32_t depth, dest;
    Rng rng;
};

struct StepFlags {
    bool was_ray, tri_hit, continues;
};

// Radiance's loop body after RayCast (pathtracer.cpp:36-196).
// returns true if the path continues with a new ray in P, false if accumColour is final
// VermiTexture::Sample (meshEngine.cpp:21-46): wrap x - floor(x), nearest round(x*(W-1)), 1-4 channels.
// The texel index is clamped (the reference indexes out of bounds on a NaN uv).
__device__ __forceinline__ float4 tex_sample_of(const float *tex, uint32_t tex_w, uint32_t tex_h, uint32_t tex_c, float u,
                                                float v) {
    const float sx = u - floorf(u), sy = v - floorf(v);
    uint32_t mx = (uint32_t)roundf(sx * (float)(int)(tex_w - 1));
    uint32_t my = (uint32_t)roundf(sy * (float)(int)(tex_h - 1));
    mx = min(mx, tex_w - 1);
    my = min(my, tex_h - 1);
    const float *p = tex + ((size_t)my * tex_w + mx) * tex_c;
    switch (tex_c) {
        case 1: return make_float4(p[0], p[0], p[0], p[0]);
        case 2: return make_float4(p[0], p[1], 0.f, 0.f);
        case 3: return make_float4(p[0], p[1], p[2], 0.f);
        default: return make_float4(p[0], p[1], p[2], p[3]);
    }
}
__device__ __forceinline__ float4 tex_sample(const SceneDev &sc, float u, float v) {
    return tex_sample_of(sc.tex, sc.tex_w, sc.tex_h, sc.tex_c, u, v);
}

// cosf / sinf of r1 in [0, 2 pi] as glibc computes them (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h —
// ARM optimized-routines): the default reading of the unqualified cos(r1) / sin(r1) with float r1 at
// pathtracer.cpp:155,162 (include/vermilion_hip.h: VMX_SAMPLING_LIBM_DOUBLE).  Double-precision operations in
// glibc's order, no contraction; the test suite's CPU restatement of the same algorithm is pinned against the
// host libm on every float of the range, and vmx_trig() exposes this function for the comparison.
// which = 0: sinf, 1: cosf.
__device__ __forceinline__ float libm_sincosf_poly(double x, double x2, bool neg, int n) {
    if ((n & 1) == 0) {
        const double x3 = x * x2, s1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13, x7 = x3 * x2,
                     s = x + x3 * -0x1.555545995a603p-3;
        return (float)(s + x7 * s1);
    }
    // the second table is the first with the cosine coefficients negated
    const double c0 = neg ? -0x1p0 : 0x1p0, c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2,
                 c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5,
                 c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10,
                 c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double x4 = x2 * x2, k2 = c3 + x2 * c4, k1 = c0 + x2 * c1, x6 = x4 * x2, c = k1 + x4 * c2;
    return (float)(c + x6 * k2);
}
__device__ __forceinline__ float libm_sincosf(float y, int which) {
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;
    double x = (double)y;
    if (top < 0x3f4u) {                      // |y| < pi/4 (abstop12 of 0x1.921FB6p-1f)
        if (top < 0x398u) return which ? 1.0f : y;  // |y| < 2^-12
        return libm_sincosf_poly(x, x * x, false, which);
    }
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int)r + 0x800000) >> 24;
    x = x - (double)n * 0x1.921FB54442D18p0;
    const double sx = ((n + 1) & 2) ? -x : x;   // sign[n & 3] = {1, -1, -1, 1}
    return libm_sincosf_poly(sx, x * x, (n & 2) != 0, n ^ which);
}

// sampling configuration of a Radiance step: r2 = r2scale * U (pathtracer.cpp:156,170; 10 or 1), and the
// reading of cos/sin(float) at :155,162 (VMX_SAMPLING_LIBM_DOUBLE)
struct SampCfg {
    float r2scale;
    uint32_t libm_double;
    uint32_t elide;  // VMX_SAMPLING_ELIDE_DEAD (step_is_dead)
};

// The loop body is split where the cosine-lobe branch needs cos/sin of r1 (double precision, a few
// hundred instructions): only ~10 % of the paths get there (the others end, or bounce off the
// mirror lobe) — a wave in which no lane gets there skips them.
struct ShadeMid {
    float sx, sy, sz;  // origin of the next ray
    float r2s, q;      // sqrt(r2), sqrt(1 - r2)
    double angle;      // r1
    float angle_f;     // r1 where the reference holds it in a float (material branch, :155); NaN marks the double branch
};
// cos and sin of a path's r1 under the configured reading
__device__ __forceinline__ void shade_trig(const ShadeMid &m, uint32_t libm_double, float &cs, float &sn) {
    if (!libm_double && m.angle_f == m.angle_f) {
        cs = libm_sincosf(m.angle_f, 1);
        sn = libm_sincosf(m.angle_f, 0);
    } else {
        // one range reduction for both (sincos returns exactly sin and cos: same kernels; the frame tests would show)
        double s, c;
        sincos(m.angle, &s, &c);
        cs = (float)c, sn = (float)s;
    }
}
enum { kPathEnded = 0, kPathNextRay = 1, kPathNeedsTrig = 2 };

__device__ __forceinline__ bool path_set_ray(Path &P, StepFlags &fl, float sx, float sy, float sz, float ndx, float ndy,
                                             float ndz) {
    P.ox = sx, P.oy = sy, P.oz = sz;
    P.dx = ndx, P.dy = ndy, P.dz = ndz;
    // An all-NaN direction (r2 > 1, pathtracer.cpp:156-162) misses the BVH and
    // every sphere in RayCast, so Radiance returns accumColour: end the path here.
    if ((ndx != ndx) && (ndy != ndy) && (ndz != ndz)) return false;
    fl.continues = finite3(ndx, ndy, ndz);
    return true;
}

template <bool TEX>
__device__ __forceinline__ int path_shade_begin(const SceneDev &sc, float r2scale, Path &P, const CastResult &c,
                                                StepFlags &fl, ShadeMid &m) {
    m.angle_f = __uint_as_float(0x7FC00000u);
    fl.tri_hit = c.slot >= 0;
    fl.continues = false;
    if (!(c.nearest < kInf)) return kPathEnded;  // pathtracer.cpp:36-41
    if (TEX) {  // :43
        P.ar = P.ar + P.tr * c.cr;
        P.ag = P.ag + P.tg * c.cg;
        P.ab = P.ab + P.tb * c.cb;
        P.aw = P.aw + P.tw * 0.f;  // accumRadiance * vec4(hitColour, 0.f): NaN once the fourth component is not finite
    } else {
        // untextured: accumRadiance stays (1,1,1,1) — the only factor ever applied to it is the white
        // albedo (:75-79,153) — so the product is exactly hitColour
        P.ar = P.ar + c.cr;
        P.ag = P.ag + c.cg;
        P.ab = P.ab + c.cb;
    }
    if (P.depth == 0) P.aw = c.nearest;                                   // :44-47
    if (sqrtf(dot3(c.cr, c.cg, c.cb, c.cr, c.cg, c.cb)) > 1.f) return kPathEnded;  // :52
    P.depth++;
    if (P.depth > 5) {  // :56 — the draw happens only past depth 5
        const double rr = rng_u01(P.rng);
        if (rr > (double)0.95f || P.depth > 1000) return kPathEnded;
    }
    // hit location and back-off along the incoming ray (:108,163,189)
    const float lx = P.ox + (P.dx * c.nearest), ly = P.oy + (P.dy * c.nearest), lz = P.oz + (P.dz * c.nearest);
    m.sx = lx - P.dx * 0.001f, m.sy = ly - P.dy * 0.001f, m.sz = lz - P.dz * 0.001f;
    bool specular = false;
    if (c.material) specular = rng_u01(P.rng) >= 0.96;  // :98
    if (specular) {
        (void)rng_next(P.rng);
        (void)rng_next(P.rng);
        (void)rng_next(P.rng);  // unused noise, :101-103
        const float k = dot3(c.nx, c.ny, c.nz, P.dx, P.dy, P.dz);
        float ndx = P.dx - c.nx * 2.f * k;
        float ndy = P.dy - c.ny * 2.f * k;
        float ndz = P.dz - c.nz * 2.f * k;
        normalize3(ndx, ndy, ndz);
        return path_set_ray(P, fl, m.sx, m.sy, m.sz, ndx, ndy, ndz) ? kPathNextRay : kPathEnded;
    }
    if (c.material) {  // :151-165
        if (TEX) {  // accumRadiance *= sampleColour (:153), sampled at the BVH hit's uv (:63-66)
            const float4 tx = tex_sample(sc, c.uvx, c.uvy);
            P.tr = P.tr * tx.x;
            P.tg = P.tg * tx.y;
            P.tb = P.tb * tx.z;
            P.tw = P.tw * tx.w;
        }
        const float r1 = (float)(6.283185307179586 * rng_u01(P.rng));
        const float r2 = (float)((double)r2scale * rng_u01(P.rng));
        m.r2s = sqrtf(r2);
        m.angle = (double)r1;
        m.angle_f = r1;
        m.q = sqrtf(1.0f - r2);
    } else {  // :166-196 — nearest hit is a sphere and the BVH hit nothing
        const double r1 = 6.283185307179586 * rng_u01(P.rng);
        const double r2 = (double)r2scale * rng_u01(P.rng);
        m.r2s = (float)sqrt(r2);
        (void)rng_next(P.rng);
        (void)rng_next(P.rng);
        (void)rng_next(P.rng);  // :173-175
        m.angle = r1;
        m.q = (float)sqrt(1.0 - r2);
    }
    // r2 > 1 (9 draws in 10 with the reference's r2 = 10 U): q is NaN, so every component of
    // (u cs r2s + v sn r2s) + w q is NaN whatever cos and sin are, and the path ends (path_set_ray)
    if (m.q != m.q) return kPathEnded;
    return kPathNeedsTrig;
}

// second half of the cosine-lobe branch (pathtracer.cpp:158-165,176-196): cs = (float)cos(r1), sn = (float)sin(r1)
__device__ __forceinline__ bool path_shade_end(Path &P, const CastResult &c, StepFlags &fl, const ShadeMid &m, float cs,
                                               float sn) {
    const bool facing = dot3(c.nx, c.ny, c.nz, P.dx, P.dy, P.dz) < 0.f;
    const float wx = facing ? c.nx : c.nx * -1.f, wy = facing ? c.ny : c.ny * -1.f,
                wz = facing ? c.nz : c.nz * -1.f;
    const bool use_y = (double)fabsf(wx) > .1;
    const float ax = use_y ? 0.f : 1.f, ay = use_y ? 1.f : 0.f, az = 0.f;
    float ux, uy, uz, vx, vy, vz;
    cross3(ax, ay, az, wx, wy, wz, ux, uy, uz);
    normalize3(ux, uy, uz);
    cross3(wx, wy, wz, ux, uy, uz, vx, vy, vz);
    float ndx = (ux * cs * m.r2s + vx * sn * m.r2s) + wx * m.q;
    float ndy = (uy * cs * m.r2s + vy * sn * m.r2s) + wy * m.q;
    float ndz = (uz * cs * m.r2s + vz * sn * m.r2s) + wz * m.q;
    normalize3(ndx, ndy, ndz);
    return path_set_ray(P, fl, m.sx, m.sy, m.sz, ndx, ndy, ndz);
}

// VMX_SAMPLING_ELIDE_DEAD: is the Radiance step that would trace this ray provably without effect on the path's colour?
// `rng`, `depth`, the ray and the throughput are the path's state before that step.  The step adds accumRadiance *
// hitColour (pathtracer.cpp:43), and hitColour is non-zero only where a light sphere is the hit (meshEngine.cpp:382-383,
// 415-416); a miss returns accumColour as it is (:36-41).  The step is the path's last one, whatever it hits, iff the
// path's own draws say so for BOTH values of the material flag:
//   Russian roulette (:56, past depth 5): its draw > 0.95f, or
//   material (:98-165): not the mirror branch (next draw < 0.96) and r2 = float(10 * third draw) > 1, so sqrt(1 - r2)
//     is NaN and with it the next direction, which misses everything and
//   no material (:166-196): r2 = 10 * second draw > 1 (double), the same.
// No light sphere can colour it iff sphereIntersect is 0 for every emitting sphere — the reference's own arithmetic with
// nothing nearer yet (limit = infinity).  Then accumColour after the step equals accumColour before it bit for bit
// (x + t * 0 == x for finite t), so the ray need not be traced: with the reference's r2 = 10 U that is 78 % of all rays.
// (Under VMX_SAMPLING_CORRECTED r2 = U never exceeds 1: only Russian roulette, past depth 5, ever ends a path by its draws.)
// The same facts as three bits — what the kernels pass along with a ray (DESIGN_HISTORY.md 5.1), step_is_dead being "bits == 3":
//   bit 0  the step is the path's last one if its hit has a material   (Russian roulette, or the draws of :98 / :156)
//   bit 1  ... if its hit has none                                      (Russian roulette, or the draw of :170)
//   bit 2  some light sphere passes sphereIntersect > 0 for the ray with nothing nearer yet: hitColour may be non-zero
// (bits 0 and 1 are left clear for a path whose throughput is not finite: inf * 0 would be NaN, the step is shaded in full)
// (test_lights == false: the caller has shown that no ray of this pixel can pass sphereIntersect > 0 for any emitting
// sphere — pixel_may_reach_a_light — so bit 2 is clear without the per-ray tests, and the direction is not needed at all)
template <bool TEX>
__device__ __forceinline__ uint32_t step_bits(const SceneDev &sc, float r2scale, Rng rng, uint32_t depth, float ox, float oy,
                                              float oz, float dx, float dy, float dz, float tr, float tg, float tb,
                                              bool test_lights = true) {
    bool rr_end = false;
    if (depth + 1u > 5u) {
        const double rr = rng_u01(rng);
        rr_end = rr > (double)0.95f || depth + 1u > 1000u;
    }
    const double a = rng_u01(rng), b = rng_u01(rng), c = rng_u01(rng);
    const float r2m = (float)((double)r2scale * c);
    bool ends_mat = rr_end || (!(a >= 0.96) && (1.0f - r2m) < 0.0f);
    bool ends_nomat = rr_end || (1.0 - (double)r2scale * b) < 0.0;
    if (TEX && !finite3(tr, tg, tb)) ends_mat = ends_nomat = false;
    bool light = false;
    for (uint32_t i = 0; test_lights && i < sc.emit_prefix; ++i) {
        const SphereDev &q = sc.spheres[i];
        if (q.flags & 1u) {
            const float opx = q.cx - ox, opy = q.cy - oy, opz = q.cz - oz;
            if (sphere_in_reach(opx, opy, opz, dot3(opx, opy, opz, opx, opy, opz), q.rad2, dx, dy, dz)) light = true;
        }
    }
    return (ends_mat ? 1u : 0u) | (ends_nomat ? 2u : 0u) | (light ? 4u : 0u);
}
template <bool TEX>
__device__ __forceinline__ bool step_is_dead(const SceneDev &sc, float r2scale, Rng rng, uint32_t depth, float ox, float oy,
                                             float oz, float dx, float dy, float dz, float tr, float tg, float tb) {
    return step_bits<TEX>(sc, r2scale, rng, depth, ox, oy, oz, dx, dy, dz, tr, tg, tb) == 3u;
}

// Radiance's loop body after RayCast in one piece (the fused kernels)
template <bool TEX, bool LDS_GEOM = false>
__device__ __forceinline__ bool path_shade(const SceneDev &sc, SampCfg cfg, Path &P, const CastResult &c,
                                           StepFlags &fl, const float4 *geom = nullptr) {
    ShadeMid m;
    const int st = path_shade_begin<TEX>(sc, cfg.r2scale, P, c, fl, m);
    bool alive = st == kPathNextRay;
    if (st == kPathNeedsTrig) {
        float sn, cs;
        shade_trig(m, cfg.libm_double, cs, sn);
        alive = path_shade_end(P, c, fl, m, cs, sn);
    }
    if (cfg.elide && alive && step_is_dead<TEX>(sc, cfg.r2scale, P.rng, P.depth, P.ox, P.oy, P.oz, P.dx, P.dy, P.dz, P.tr, P.tg, P.tb))
        alive = false, fl.continues = false;
    return alive;
}

__device__ __forceinline__ void tally_add(Tally &tl, const StepFlags &fl, bool ran, uint32_t stage_depth0) {
    // stage 0 = steps taken at depth 0, stage 1 = bounce steps
    const unsigned long long r0 = __ballot(ran && fl.was_ray && stage_depth0);
    const unsigned long long r1 = __ballot(ran && fl.was_ray && !stage_depth0);
    const unsigned long long h0 = __ballot(ran && fl.tri_hit && fl.was_ray && stage_depth0);
    const unsigned long long h1 = __ballot(ran && fl.tri_hit && fl.was_ray && !stage_depth0);
    const unsigned long long c0 = __ballot(ran && fl.continues && stage_depth0);
    const unsigned long long c1 = __ballot(ran && fl.continues && !stage_depth0);
    tl.rays[0] += (uint32_t)__popcll(r0);
    tl.rays[1] += (uint32_t)__popcll(r1);
    tl.hits[0] += (uint32_t)__popcll(h0);
    tl.hits[1] += (uint32_t)__popcll(h1);
    tl.cont[0] += (uint32_t)__popcll(c0);
    tl.cont[1] += (uint32_t)__popcll(c1);
}

template <bool COUNT>
__device__ __forceinline__ void tally_flush(DevCounters *ctr, const Tally &tl, const Cnt &c0, const Cnt &c1) {
    uint32_t i0 = 0, t0 = 0, i1 = 0, t1 = 0;
    if (COUNT) {
        i0 = wave_sum(c0.inner);
        t0 = wave_sum(c0.tris);
        i1 = wave_sum(c1.inner);
        t1 = wave_sum(c1.tris);
    }
    if (lane_index() == 0) {
        for (int s = 0; s < 2; ++s) {
            if (tl.rays[s]) atomicAdd(&ctr->stage[s].rays, (unsigned long long)tl.rays[s]);
            if (tl.hits[s]) atomicAdd(&ctr->stage[s].tri_hits, (unsigned long long)tl.hits[s]);
            if (tl.cont[s]) atomicAdd(&ctr->stage[s].continued, (unsigned long long)tl.cont[s]);
        }
        if (COUNT) {
            if (i0) atomicAdd(&ctr->stage[0].inner_visits, (unsigned long long)i0);
            if (t0) atomicAdd(&ctr->stage[0].tri_tests, (unsigned long long)t0);
            if (i1) atomicAdd(&ctr->stage[1].inner_visits, (unsigned long long)i1);
            if (t1) atomicAdd(&ctr->stage[1].tri_tests, (unsigned long long)t1);
        }
    }
}

// ---------------------------------------------------------------------------
// ray generation (pathtracer.cpp:251-280); p = global pixel index, k = linear sample index
// ---------------------------------------------------------------------------
// a / n in double, correctly rounded, for an integer 1 <= n < 2^32 whose reciprocal y = RN(1 / n) the host computed by an
// IEEE division (FrameDev::inv_width / inv_height): three operations instead of the ~35 of a double division, two of
// which sat in every camera ray's generation (pathtracer.cpp:251-252 divide by the image size in double).  Markstein's
// correction step: q0 = RN(a y) is within 2 ulp of a / n; e = a - n q0 is exact in an FMA (a multiple of ulp(q0), at
// most 2^33 of them); q0 + e y = a / n + (e / n) delta with |delta| <= 2^-53, i.e. within 2^-104 relative of a / n —
// while a / n, n an integer below 2^32, is either a double or at least 2^-86 relative away from every midpoint
// between two doubles (a - m n, m such a midpoint, is a non-zero multiple of ulp(m) / 2: m n has more than 53
// significant bits and cannot equal a).  So RN(q0 + e y) = RN(a / n), for every finite a.  Checked exhaustively on
// the CPU for every float a camera ray can produce at the bench's and the tests' image sizes
// (tests/test_kernel_shortcuts.py::test_division_by_the_image_size...).
__device__ __forceinline__ double div_by_count(double a, uint32_t n, double y) {
    const double q0 = a * y;
    const double e = __fma_rn(-q0, (double)n, a);
    return __fma_rn(e, y, q0);
}
// (pk: the pixel half of the stream key, which a caller that walks the samples of one pixel forms once)
__device__ __forceinline__ void primary_ray_keyed(const FrameDev &fr, uint32_t p, PixelKey pk, uint32_t k, Rng &rng, float &dx,
                                                  float &dy, float &dz) {
    rng_init_keyed(rng, pk, k);
    const float jx = rng_jitter(rng);  // :251
    const float jy = rng_jitter(rng);  // :252
    const uint32_t s = k / fr.quarter;
    const float sx = (float)(s >> 1), sy = (float)(s & 1u);
    const float fx = (float)(p % fr.width) + (sx * 0.5f - 0.5f) + jx;
    const float fy = (float)(p / fr.width) + (sy * 0.5f - 0.5f) + jy;
    const float hx = (float)(div_by_count((double)fx - 0.25, fr.width, fr.inv_width) - 0.5);
    const float hy = (float)(div_by_count((double)fy - 0.25, fr.height, fr.inv_height) - 0.5);
    const float gx = hx * fr.sensor_x, gy = -(hy * fr.sensor_y), gz = -fr.film_dist;
    // cameraTransform * gridPane, GLM order (m0*x + m1*y) + (m2*z + m3*w), m3.xyz = 0, w = 1
    float rx = (fr.m[0] * gx + fr.m[3] * gy) + (fr.m[6] * gz + 0.0f);
    float ry = (fr.m[1] * gx + fr.m[4] * gy) + (fr.m[7] * gz + 0.0f);
    float rz = (fr.m[2] * gx + fr.m[5] * gy) + (fr.m[8] * gz + 0.0f);
    normalize3(rx, ry, rz);
    dx = rx, dy = ry, dz = rz;
}
__device__ __forceinline__ void primary_ray(const FrameDev &fr, uint32_t p, uint32_t k, Rng &rng, float &dx,
                                            float &dy, float &dz) {
    primary_ray_keyed(fr, p, rng_pixel_key(fr.seed, p), k, rng, dx, dy, dz);
}

// Can ANY camera ray of pixel p pass sphereIntersect > 0 (meshEngine.cpp:182-194, nothing nearer yet) for some emitting
// sphere?  A conservative answer for the whole pixel, so that k_raygen need not run the per-ray tests — nor, where the
// step's draws already say "last step", form the ray's direction at all — for the pixels that cannot: most of a frame.
// The sample directions of a pixel are normalize(M film) with the film point within half a pixel of the centre
// ((px - 0.25) / W - 0.5, pathtracer.cpp:251-252 with the offsets' range [-0.75, 0.25)): a cone of half-angle phi,
// sin phi <= half the pixel's film diagonal / |film point|, around the centre direction a.  For a sphere (centre c,
// radius r), op = c - o, cos theta = op.a / |op|: every ray of the cone has op.d / |op| <= m = cos(theta - phi), and
// sphereIntersect returns 0 when b^2 < |op|^2 - r^2 (negative discriminant) or when b < 0 with the origin outside the
// sphere (both roots negative).  "Cannot reach" is claimed only with max(m, 0)^2 (1 + 1e-4) + 1e-3 < 1 - r^2 / |op|^2:
// a margin five orders of magnitude above the float rounding of b = dot(op, d), |op|^2 and r^2 that the per-ray test
// (sphere_in_reach) computes with — a sphere within ~1.8 degrees of the cone is simply tested per ray as before.
__device__ __forceinline__ bool pixel_may_reach_a_light(const SceneDev &sc, const FrameDev &fr, uint32_t p) {
    const float hx = ((float)(p % fr.width) - 0.25f) / (float)fr.width - 0.5f;
    const float hy = ((float)(p / fr.width) - 0.25f) / (float)fr.height - 0.5f;
    const float gx = hx * fr.sensor_x, gy = -(hy * fr.sensor_y), gz = -fr.film_dist;
    const float ax = fr.m[0] * gx + fr.m[3] * gy + fr.m[6] * gz;
    const float ay = fr.m[1] * gx + fr.m[4] * gy + fr.m[7] * gz;
    const float az = fr.m[2] * gx + fr.m[5] * gy + fr.m[8] * gz;
    const float alen = sqrtf(ax * ax + ay * ay + az * az);
    const float px = fr.sensor_x / (float)fr.width, py = fr.sensor_y / (float)fr.height;
    const float sphi = fminf(0.505f * sqrtf(px * px + py * py) / alen, 1.0f);  // half the diagonal, +1 %
    const float cphi = sqrtf(fmaxf(1.0f - sphi * sphi, 0.0f));
    bool may = !(alen > 0.0f);  // (a degenerate film: no claim)
    for (uint32_t i = 0; i < sc.emit_prefix; ++i) {
        const SphereDev &q = sc.spheres[i];
        if (!(q.flags & 1u)) continue;
        const float opx = q.cx - fr.px, opy = q.cy - fr.py, opz = q.cz - fr.pz;
        const float C = opx * opx + opy * opy + opz * opz;
        const float ct = (opx * ax + opy * ay + opz * az) / (sqrtf(C) * alen);
        const float st = sqrtf(fmaxf(1.0f - ct * ct, 0.0f));
        const float m = ct >= cphi ? 1.0f : fmaxf(ct * cphi + st * sphi, 0.0f);  // cos(theta - phi); 1 inside the cone
        const float clear = 1.0f - q.rad2 / C;                                    // 1 - r^2 / |op|^2
        if (!(m * m * 1.0001f + 1e-3f < clear)) may = true;                      // (NaN / inf / origin inside: may)
    }
    return may;
}

// n / d for a launch constant d (vmx_device.h: FastDiv)
__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv &f) {
    const uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
// local pixel index (rank-local packed rows) -> global pixel index
__device__ __forceinline__ uint32_t global_pixel(const FrameDev &fr, uint32_t lp) {
    if (fr.world <= 1) return lp;
    const uint32_t lrow = fast_div(lp, fr.div_width), x = lp - lrow * fr.width;
    const uint32_t ls = fast_div(lrow, fr.div_stripe), r = lrow - ls * fr.stripe_rows;
    const uint32_t grow = (ls * fr.world + fr.rank) * fr.stripe_rows + r;
    return grow * fr.width + x;
}

// ---- path state in per-pass arrays (split wavefront) ---------------------------------------
__device__ __forceinline__ void rng_store(const PathArrays &pa, uint32_t pid, const Rng &r) {
    ((float4 *)pa.state)[(size_t)pid * 4 + 2] = make_float4(__uint_as_float((uint32_t)r.s0), __uint_as_float((uint32_t)(r.s0 >> 32)),
                                           __uint_as_float((uint32_t)r.s1), __uint_as_float((uint32_t)(r.s1 >> 32)));
    ((float4 *)pa.state)[(size_t)pid * 4 + 3] = make_float4(__uint_as_float((uint32_t)r.s2), __uint_as_float((uint32_t)(r.s2 >> 32)),
                                           __uint_as_float((uint32_t)r.s3), __uint_as_float((uint32_t)(r.s3 >> 32)));
}
__device__ __forceinline__ void rng_load(const PathArrays &pa, uint32_t pid, Rng &r) {
    const float4 e = ((const float4 *)pa.state)[(size_t)pid * 4 + 2], f = ((const float4 *)pa.state)[(size_t)pid * 4 + 3];
    r.s0 = (uint64_t)__float_as_uint(e.x) | ((uint64_t)__float_as_uint(e.y) << 32);
    r.s1 = (uint64_t)__float_as_uint(e.z) | ((uint64_t)__float_as_uint(e.w) << 32);
    r.s2 = (uint64_t)__float_as_uint(f.x) | ((uint64_t)__float_as_uint(f.y) << 32);
    r.s3 = (uint64_t)__float_as_uint(f.z) | ((uint64_t)__float_as_uint(f.w) << 32);
}
// bits: step_bits of the step that will trace this ray (k_trace_w<1, .., SORT> reads them back); 0 = "shade it in full"
__device__ __forceinline__ void ray_store(const PathArrays &pa, uint32_t pid, const Path &P, uint32_t bits = 0) {
    ((float4 *)pa.state)[(size_t)pid * 4] = make_float4(P.ox, P.oy, P.oz, P.dx);
    ((float4 *)pa.state)[(size_t)pid * 4 + 1] = make_float4(P.dy, P.dz, __uint_as_float(P.depth), __uint_as_float(bits));
}
__device__ __forceinline__ void ray_load(const PathArrays &pa, uint32_t pid, Path &P) {
    const float4 a = ((const float4 *)pa.state)[(size_t)pid * 4], b = ((const float4 *)pa.state)[(size_t)pid * 4 + 1];
    P.ox = a.x, P.oy = a.y, P.oz = a.z, P.dx = a.w;
    P.dy = b.x, P.dz = b.y, P.depth = __float_as_uint(b.z);
}
// accumColour of a path in the split passes (PathArrays::rad / rad_mask).  The mask bit of a path says "rad[pid] holds a
// non-zero colour": a path that is black so far has neither a stored radiance nor a bit — k_resolve adds an exact +0 for
// it — and the first step that colours it stores the sum and sets the bit (rare: one atomic).  Without a mask (fused
// passes, vmx_radiance) every path's radiance is in rad.
__device__ __forceinline__ bool rad_has(const PathArrays &pa, uint32_t pid) {
    return !pa.rad_mask || ((pa.rad_mask[pid >> 6] >> (pid & 63u)) & 1ull) != 0;
}
__device__ __forceinline__ float4 rad_fetch(const PathArrays &pa, uint32_t pid, bool has) {
    return has ? ((const float4 *)pa.rad)[pid] : make_float4(0.f, 0.f, 0.f, -100.f);
}
__device__ __forceinline__ void rad_commit(const PathArrays &pa, uint32_t pid, float4 v, bool had) {
    if (had || v.x != 0.f || v.y != 0.f || v.z != 0.f) {  // (NaN != 0: a NaN colour is stored)
        ((float4 *)pa.rad)[pid] = v;
        if (!had) atomicOr(&pa.rad_mask[pid >> 6], 1ull << (pid & 63u));
    }
}
// RAD = false: accumColour is not fetched (k_shade<1> fetches and stores it only for the steps that change it)
template <bool TEX, bool RAD = true>
__device__ __forceinline__ void path_load_arrays(const PathArrays &pa, uint32_t pid, Path &P) {
    ray_load(pa, pid, P);
    rng_load(pa, pid, P.rng);
    if (RAD) {
        const float4 acc = rad_fetch(pa, pid, rad_has(pa, pid));
        P.ar = acc.x, P.ag = acc.y, P.ab = acc.z, P.aw = acc.w;
    } else {
        P.ar = P.ag = P.ab = P.aw = 0.f;
    }
    if (TEX) {
        const float4 th = ((const float4 *)pa.thr)[pid];
        P.tr = th.x, P.tg = th.y, P.tb = th.z, P.tw = th.w;
    }
    P.dest = pid;
}
// id compaction: wave ballot + prefix popcount, one atomic per wave
__device__ __forceinline__ void id_append(const IdQueue &q, uint32_t sub, bool alive, uint32_t pid) {
    const unsigned long long m = __ballot(alive);
    if (m == 0) return;
    const uint32_t lane = lane_index();
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&q.counts[sub * 32], (uint32_t)__popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    // (the sub-lists are sized so that this cannot fail — ensure_paths — but an append never writes past its list: the
    // tail counter then stays above sub_capacity, which the host reads as an error, run_ids / finish_stats)
    if (alive && base + rank < q.sub_capacity) q.ids[(size_t)sub * q.sub_capacity + base + rank] = pid;
}

// Path id of sample plane j of active-pixel slot s.  pixel_major: the samples of one pixel are
// contiguous (pid = s * samples + j), so the 64 camera rays of a wave differ only by sub-pixel
// jitter and walk the BVH together; else plane-major (pid = j * n_pad + s, first-generation kernels).
__device__ __forceinline__ uint32_t path_id(const WorkDev &wk, uint32_t j, uint32_t s_idx) {
    return wk.pixel_major ? s_idx * wk.samples + j : j * wk.n_pad + s_idx;
}

// band-local work item of the primary source -> path id, global pixel, sample index
// Sample index of the j-th path a pixel gets in this pass.  Bit 31 of the cursor marks a pixel whose
// last stratum ended on the early-stop rule (pathtracer.cpp:290-311): it will most likely end the
// following strata on their first sample too, so its paths of a pass are the first samples of the
// next strata (k = cursor + j * quarter) instead of consecutive samples; k_resolve takes them in
// order and drops what the rule would not have reached.
constexpr uint32_t kCursorStrided = 0x80000000u;
__device__ __forceinline__ uint32_t sample_index(const FrameDev &fr, const PixelStateDev &px, uint32_t lp,
                                                 uint32_t j) {
    // first pass of an early-stop frame: the samples before the rule can fire (j < lead), and already
    // the first samples of the following strata, for the many pixels that stop on the first test
    if (fr.lead != 0 && j >= fr.lead) return (j - fr.lead + 1u) * fr.quarter;
    const uint32_t c = px.cursor[lp];
    return (c & ~kCursorStrided) + j * ((c & kCursorStrided) ? fr.quarter : 1u);
}

__device__ __forceinline__ bool primary_item(const FrameDev &fr, const WorkDev &wk, const PixelStateDev &px,
                                             uint32_t j, uint32_t s_idx, uint32_t &pid, uint32_t &pixel,
                                             uint32_t &k) {
    if (s_idx >= wk.n_active) return false;
    const uint32_t lp = wk.active[s_idx];
    k = sample_index(fr, px, lp, j);
    if (k >= fr.kmax) return false;
    pixel = global_pixel(fr, lp);
    pid = path_id(wk, j, s_idx);
    return true;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
template <bool COUNT>
__global__ void k_trace(SceneDev sc, const float *__restrict__ o, const float *__restrict__ d, uint32_t n,
                        int32_t *__restrict__ tri_id, float *__restrict__ t, DevCounters *ctr) {
    extern __shared__ uint2 lds_stack[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint2 *stk = lds_stack + (size_t)wave * sc.stack_entries * 64 + lane;
    const float4 *__restrict__ tris = (const float4 *)sc.tris;
    Cnt cnt = {0, 0}, none = {0, 0};
    Tally tl = {{0, 0}, {0, 0}, {0, 0}};
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        if (i < n) {
            float best;
            int slot;
            bvh_nearest<COUNT>(sc, o[i * 3], o[i * 3 + 1], o[i * 3 + 2], d[i * 3], d[i * 3 + 1], d[i * 3 + 2], stk,
                               best, slot, cnt);
            tri_id[i] = slot >= 0 ? (int32_t)__float_as_uint(tris[slot * 3 + 2].y) : -1;
            t[i] = best;
        }
    }
    if (COUNT && ctr) tally_flush<true>(ctr, tl, cnt, none);
}

__global__ void k_raycast(SceneDev sc, const float *__restrict__ o, const float *__restrict__ d, uint32_t n,
                          float4 *__restrict__ out) {
    extern __shared__ uint2 lds_stack[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint2 *stk = lds_stack + (size_t)wave * sc.stack_entries * 64 + lane;
    const float4 *__restrict__ tris = (const float4 *)sc.tris;
    Cnt cnt = {0, 0};
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        if (i >= n) continue;
        const float ox = o[i * 3], oy = o[i * 3 + 1], oz = o[i * 3 + 2];
        const float dx = d[i * 3], dy = d[i * 3 + 1], dz = d[i * 3 + 2];
        CastResult c;
        ray_cast<false>(sc, ox, oy, oz, dx, dy, dz, stk, c, cnt);
        const int32_t id = c.slot >= 0 ? (int32_t)__float_as_uint(tris[c.slot * 3 + 2].y) : -1;
        const uint32_t flags = ((c.nearest < kInf) ? 1u : 0u) | (c.material ? 2u : 0u);
        // vmx_rayhit: location[3], distance | normal[3], tri_id | uv[2], tri_t, flags | colour[3], pad
        out[(size_t)i * 4] = make_float4(ox + (dx * c.nearest), oy + (dy * c.nearest), oz + (dz * c.nearest), c.nearest);
        out[(size_t)i * 4 + 1] = make_float4(c.nx, c.ny, c.nz, __int_as_float(id));
        out[(size_t)i * 4 + 2] = make_float4(c.uvx, c.uvy, c.tri_t, __uint_as_float(flags));
        out[(size_t)i * 4 + 3] = make_float4(c.cr, c.cg, c.cb, 0.f);
    }
}

__global__ void k_primary_ids(SceneDev sc, FrameDev fr, uint32_t k, int32_t *__restrict__ tri_id,
                              float *__restrict__ t) {
    extern __shared__ uint2 lds_stack[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint2 *stk = lds_stack + (size_t)wave * sc.stack_entries * 64 + lane;
    const float4 *__restrict__ tris = (const float4 *)sc.tris;
    const uint32_t npix = fr.width * fr.height;
    Cnt cnt = {0, 0};
    for (uint32_t base = blockIdx.x * blockDim.x; base < npix; base += gridDim.x * blockDim.x) {
        const uint32_t p = base + threadIdx.x;
        if (p >= npix) continue;
        Rng rng;
        float dx, dy, dz;
        primary_ray(fr, p, k, rng, dx, dy, dz);
        float best;
        int slot;
        bvh_nearest<false>(sc, fr.px, fr.py, fr.pz, dx, dy, dz, stk, best, slot, cnt);
        tri_id[p] = slot >= 0 ? (int32_t)__float_as_uint(tris[slot * 3 + 2].y) : -1;
        t[p] = best;
    }
}

// parity hook: cosf / sinf as the shading kernels evaluate them under the default reading
__global__ void k_trig(const float *__restrict__ x, uint32_t n, float *__restrict__ cs, float *__restrict__ sn) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        cs[i] = libm_sincosf(x[i], 1);
        sn[i] = libm_sincosf(x[i], 0);
    }
}

__global__ void k_init_pixels(PixelStateDev px, uint32_t npix) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    ((float4 *)px.accum)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    px.count[i] = 0;
    px.cursor[i] = 0;
}

__global__ void k_zero_u32(unsigned int *p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

// ---------------------------------------------------------------------------
// quad-cooperative record fetch (k_trace_w<1>, k_paths).  A 64-byte record read by ONE lane as four 16-byte
// loads costs four tag lookups in the vector L1; read by the four lanes of a quad — lane j takes
// bytes [16 j, 16 j + 16) — it costs one.  Load i of a step brings the record of quad-lane i, so
// afterwards lane j holds piece j of the four records of its quad; a 4x4 transpose inside every
// quad (two butterfly stages, one v_cndmask_b32_dpp per dword and stage: D = vcc ? src1 : dpp(src0))
// hands every lane the four pieces of its own record.  tools/ta_probe.hip measures both forms.
// ---------------------------------------------------------------------------
#define VMX_XCHG_PAIR(NAME, PERM)                                                                                     \
    __device__ __forceinline__ void NAME(const float4 &p, const float4 &q, float4 &po, float4 &qo,                    \
                                         unsigned long long keep_p, unsigned long long keep_q) {                     \
        asm volatile("s_nop 1\n\t"                                                                                   \
                     "s_mov_b64 vcc, %16\n\t"                                                                        \
                     "v_cndmask_b32_dpp %0, %12, %8, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"          \
                     "v_cndmask_b32_dpp %1, %13, %9, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"          \
                     "v_cndmask_b32_dpp %2, %14, %10, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"         \
                     "v_cndmask_b32_dpp %3, %15, %11, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"         \
                     "s_mov_b64 vcc, %17\n\t"                                                                        \
                     "v_cndmask_b32_dpp %4, %8, %12, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"          \
                     "v_cndmask_b32_dpp %5, %9, %13, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"          \
                     "v_cndmask_b32_dpp %6, %10, %14, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"         \
                     "v_cndmask_b32_dpp %7, %11, %15, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf"             \
                     : "=&v"(po.x), "=&v"(po.y), "=&v"(po.z), "=&v"(po.w), "=&v"(qo.x), "=&v"(qo.y), "=&v"(qo.z),    \
                       "=&v"(qo.w)                                                                                   \
                     : "v"(p.x), "v"(p.y), "v"(p.z), "v"(p.w), "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w), "s"(keep_p),  \
                       "s"(keep_q)                                                                                   \
                     : "vcc");                                                                                       \
    }
VMX_XCHG_PAIR(xchg_pair_1, "[1,0,3,2]")
VMX_XCHG_PAIR(xchg_pair_2, "[2,3,0,1]")
#undef VMX_XCHG_PAIR

// Fetches the 64 bytes at base + off (off: 32-bit byte offset of the lane's node or triangle record, 16-byte
// aligned) into q0..q3 of every lane.  A lane that is not traversing passes off = 0 (the first record: always
// there, and hot): loading for it unconditionally is cheaper than masking the four loads.  Must be called with
// all 64 lanes active.
__device__ __forceinline__ void quad_fetch_record(const char *base, uint32_t off, uint32_t lane, float4 &q0, float4 &q1,
                                                  float4 &q2, float4 &q3) {
    const uint32_t piece = (lane & 3u) << 4;
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)off, 0x00, 0xF, 0xF, true);  // quad_perm [0,0,0,0]
    const uint32_t r1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)off, 0x55, 0xF, 0xF, true);  // [1,1,1,1]
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)off, 0xAA, 0xF, 0xF, true);  // [2,2,2,2]
    const uint32_t r3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)off, 0xFF, 0xF, 0xF, true);  // [3,3,3,3]
    // (one 32-bit offset per load: scalar base + vector offset addressing, no 64-bit address arithmetic)
    const float4 x0 = *(const float4 *)(base + (uint32_t)(r0 + piece));
    const float4 x1 = *(const float4 *)(base + (uint32_t)(r1 + piece));
    const float4 x2 = *(const float4 *)(base + (uint32_t)(r2 + piece));
    const float4 x3 = *(const float4 *)(base + (uint32_t)(r3 + piece));
    float4 a0, a1, a2, a3;
    xchg_pair_1(x0, x1, a0, a1, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull);  // lanes ^1: even lanes keep (x0, x2)
    xchg_pair_1(x2, x3, a2, a3, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull);
    xchg_pair_2(a0, a2, q0, q2, 0x3333333333333333ull, 0xCCCCCCCCCCCCCCCCull);  // lanes ^2: lanes 0,1 of a quad keep (a0, a1)
    xchg_pair_2(a1, a3, q1, q3, 0x3333333333333333ull, 0xCCCCCCCCCCCCCCCCull);
}


// ---------------------------------------------------------------------------
// k_paths — persistent waves with per-lane refill.
//
// Every lane runs the state machine  fetch -> traverse (one BVH node per loop
// iteration) -> shade -> {write radiance | continue (LOOP) | append to the
// next queue}.  A lane that finishes does not idle until its wave is done: as
// soon as `refill_min` lanes are idle the wave takes that many new items from
// its work source (a wave-private reservation of 256 items, one atomic per
// reservation), and as soon as `shade_min` lanes have finished traversal they
// are shaded together.  Results do not depend on this scheduling: each path
// carries its own RNG stream and radiance slot.
//   SRC 0: work = (sample j, slot) pairs of the active-pixel list, in 8 bands
//          of neighbouring tiles (band = block % 8 = XCD label; exhausted
//          bands are stolen from), paths start with ray generation
//   SRC 2: work = the 16 sub-queues of live path ids (the tail of a pass)
// ---------------------------------------------------------------------------
// LDS holds levels [0, lds_entries) plus one scratch entry at index lds_entries; deeper levels
// (rare) go to the per-wave global slab.  The LDS access is unconditional on a clamped index so
// that it stays a ds_*_b64 (a select between the two address spaces turns into flat accesses).
__device__ __forceinline__ void stack_push(uint2 *stk, uint2 *ovf, int lds_entries, int sp, uint2 e) {
    stk[min(sp, lds_entries) * 64] = e;
    if (sp >= lds_entries) ovf[(sp - lds_entries) * 64] = e;
}
__device__ __forceinline__ uint2 stack_pop(const uint2 *stk, const uint2 *ovf, int lds_entries, int sp) {
    uint2 e = stk[min(sp, lds_entries) * 64];
    // the empty asm pins the LDS read as its own ds_read_b64: without it the two loads are merged
    // into one flat load through a pointer select, which puts every pop on the vector-memory pipe
    asm volatile("" : "+v"(e.x), "+v"(e.y));
    if (sp >= lds_entries) e = ovf[(sp - lds_entries) * 64];
    return e;
}

#ifndef VMX_PATHS_WPS
#define VMX_PATHS_WPS 6  // waves per SIMD the fused kernel is compiled for (80 VGPRs): tail 8.8 -> 7.9 ms, early-stop frame 14.3 -> 13.8 ms (5: 8.1, 7: 8.4, 8: 8.9)
#endif
template <bool COUNT, int SRC, bool TEX>
__global__ void __launch_bounds__(256, VMX_PATHS_WPS)
k_paths(SceneDev sc, FrameDev fr, WorkDev wk, PixelStateDev px, float4 *__restrict__ rad, PathArrays pa,
        DevCounters *ctr) {
    extern __shared__ uint2 lds_stack[];
    __shared__ float4 s_geom[kLdsSpheres];  // spheres' (centre, rad*rad): read by every RayCast
    if (threadIdx.x < min(sc.nspheres, kLdsSpheres)) {
        const SphereDev &q = sc.spheres[threadIdx.x];
        s_geom[threadIdx.x] = make_float4(q.cx, q.cy, q.cz, q.rad2);
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // the first lds_entries stack levels live in LDS, deeper ones (rare) in a per-wave global slab
    const int lds_entries = (int)wk.lds_entries;
    uint2 *stk = lds_stack + (size_t)wave * (lds_entries + 1) * 64 + lane;
    uint2 *ovf = (uint2 *)wk.overflow_stack +
                 ((size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * wk.overflow_entries) * 64 + lane;
    const float4 *__restrict__ inner = (const float4 *)sc.inner;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t kReserve = wk.reserve;

    // wave-uniform scheduling state
    uint32_t src = blockIdx.x % wk.nsrc, res_lo = 0, res_hi = 0, tried = 0;
    bool exhausted = false;
    // lane state
    // traversal state as in k_trace_w: `cur` = node reference, kIdle while the lane is not traversing
    // (no path, or traversal finished and the path waits for shading); stack level 0 = bottom entry
    constexpr uint32_t kIdle = 0x7FFFFFFFu, kBottom = 0x7FFFFFFEu, kPop = 0x7FFFFFFDu;
    bool has = false, exact = false, is_ray = false;
    Path P;
    float ix = 0.f, iy = 0.f, iz = 0.f, best = 0.f;
    int slot = -1, sp = 0;
    uint32_t cur = kIdle;
    stk[0] = make_uint2(kBottom, 0xFF800000u);
    Cnt c0 = {0, 0}, c1 = {0, 0};
    Tally tl = {{0, 0}, {0, 0}, {0, 0}};

    auto start_traversal = [&]() {
        ix = 1.0f / P.dx, iy = 1.0f / P.dy, iz = 1.0f / P.dz;  // Ray.h:10
        exact = !(finite3(ix, iy, iz) && finite3(P.ox, P.oy, P.oz));
        is_ray = (P.depth == 0) || finite3(P.dx, P.dy, P.dz);
        best = 999999999.f;  // bvh.cpp:48
        slot = -1;
        sp = 1;
        cur = sc.root_ref;  // its near value, -9999999 (bvh.cpp:59), passes `near > t`
    };

    for (;;) {
        // ---- 1. shade the lanes whose traversal is finished ---------------------
        {
            const unsigned long long fin = __builtin_amdgcn_ballot_w64(has && cur == kIdle);
            const unsigned long long trav = __builtin_amdgcn_ballot_w64(cur != kIdle);
            if (fin != 0 && ((uint32_t)__popcll(fin) >= wk.shade_min || trav == 0)) {
                const bool shaded = has && cur == kIdle;
                bool alive = false;
                StepFlags fl = {false, false, false};
                uint32_t depth0 = 0;
                if (shaded) {
                    depth0 = P.depth == 0 ? 1u : 0u;
                    CastResult c;
                    cast_finish<true, false>(sc, P.ox, P.oy, P.oz, P.dx, P.dy, P.dz, best, slot, c, s_geom);
                    fl.was_ray = is_ray;
                    alive = path_shade<TEX, true>(sc, SampCfg{fr.r2scale, fr.libm_double, fr.elide_dead}, P, c, fl, s_geom);
                    if (!alive) {
                        // (tail of a split pass: the path's bit says whether it had a stored colour; fused passes: no mask)
                        if (SRC == 2) rad_commit(pa, P.dest, make_float4(P.ar, P.ag, P.ab, P.aw), rad_has(pa, P.dest));
                        else rad[P.dest] = make_float4(P.ar, P.ag, P.ab, P.aw);
                        has = false;
                    }
                }
                tally_add(tl, fl, shaded, depth0);
                if (shaded && alive) start_traversal();  // the lane keeps its path to the end
            }
        }
        // ---- 2. refill idle lanes from the work source ---------------------------
        {
            const unsigned long long idle = __builtin_amdgcn_ballot_w64(!has);
            if (idle != 0 && !exhausted && ((uint32_t)__popcll(idle) >= wk.refill_min || idle == ~0ull)) {
                for (;;) {
                    if (res_lo == res_hi) {
                        const uint32_t lim = SRC == 0 ? wk.band_items : min(wk.qids.counts[src * 32], wk.qids.sub_capacity);
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&wk.heads[src * 32], kReserve);
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base >= lim) {
                            src = (src + 1 == wk.nsrc) ? 0 : src + 1;
                            if (++tried == wk.nsrc) {
                                exhausted = true;
                                break;
                            }
                            continue;
                        }
                        tried = 0;
                        res_lo = base;
                        res_hi = min(base + kReserve, lim);
                    }
                    const unsigned long long want = __builtin_amdgcn_ballot_w64(!has);
                    if (want == 0) break;
                    const uint32_t avail = res_hi - res_lo;
                    const uint32_t rank = (uint32_t)__popcll(want & lt_mask);
                    const bool take = !has && rank < avail;
                    const uint32_t item = res_lo + rank;
                    res_lo += min((uint32_t)__popcll(want), avail);
                    if (take) {
                        bool valid = true;
                        if (SRC == 0) {
                            uint32_t j, s_idx;
                            if (wk.pixel_major) {
                                const uint32_t sl = fast_div(item, wk.div_samples);
                                j = item - sl * wk.samples, s_idx = src * wk.band_slots + sl;
                            } else {
                                j = item / wk.band_slots, s_idx = src * wk.band_slots + (item - j * wk.band_slots);
                            }
                            valid = s_idx < wk.n_active;
                            if (valid) {
                                const uint32_t lp = wk.active[s_idx];
                                const uint32_t k = sample_index(fr, px, lp, j);
                                valid = k < fr.kmax;
                                if (valid) {
                                    primary_ray(fr, global_pixel(fr, lp), k, P.rng, P.dx, P.dy, P.dz);
                                    P.ox = fr.px, P.oy = fr.py, P.oz = fr.pz;
                                                        P.ar = P.ag = P.ab = 0.f;
                                    P.aw = -100.f;  // pathtracer.cpp:29
                                    P.tr = P.tg = P.tb = P.tw = 1.f;  // :30
                                    P.depth = 0;
                                    P.dest = path_id(wk, j, s_idx);
                                    if (fr.elide_dead && step_is_dead<false>(sc, fr.r2scale, P.rng, 0, P.ox, P.oy, P.oz, P.dx, P.dy, P.dz, 1.f, 1.f, 1.f)) {
                                        rad[P.dest] = make_float4(0.f, 0.f, 0.f, -100.f);  // untraced: its radiance is zero
                                        valid = false;
                                    }
                                }
                            }
                        } else {
                            path_load_arrays<TEX>(pa, wk.qids.ids[(size_t)src * wk.qids.sub_capacity + item], P);
                        }
                        if (valid) {
                            has = true;
                            start_traversal();
                        }
                    }
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(has) == 0) break;
        // ---- 3. 8 traversal steps (bvh.cpp:61-134), the step of k_trace_w (see there) with counters:
        //         every traversing lane does its triangle step or its inner-node step
        auto step = [&](auto exact_tag) {
            constexpr bool EXACT = decltype(exact_tag)::value;
            // the node or triangle record of every traversing lane, fetched quad-cooperatively in one phase
            // (quad_fetch_record; inner and triangle records share one allocation, SceneDev::tri_off)
            float4 q0, q1, q2, q3;
            quad_fetch_record((const char *)inner,
                              cur >= kPop && (int)cur >= 0 ? 0u
                              : ((int)cur < 0 ? sc.tri_off + (cur & kLeafStartMask) * 48u : (cur << 6)),
                              lane, q0, q1, q2, q3);
            if (cur < kPop) {
                if (COUNT) {
                    if (P.depth == 0) c0.inner++;
                    else c1.inner++;
                }
                const float a0 = (q0.x - P.ox) * ix, a1 = (q0.y - P.oy) * iy, a2 = (q0.z - P.oz) * iz;
                const float a3 = (q0.w - P.ox) * ix, a4 = (q1.x - P.oy) * iy, a5 = (q1.y - P.oz) * iz;
                const float b0 = (q1.z - P.ox) * ix, b1 = (q1.w - P.oy) * iy, b2 = (q2.x - P.oz) * iz;
                const float b3 = (q2.y - P.ox) * ix, b4 = (q2.z - P.oy) * iy, b5 = (q2.w - P.oz) * iz;
                float tn0, tf0, tn1, tf1;
                if (EXACT) {
                    box_net_exact(a0, a1, a2, a3, a4, a5, tn0, tf0);
                    box_net_exact(b0, b1, b2, b3, b4, b5, tn1, tf1);
                } else {
                    box_net(a0, a1, a2, a3, a4, a5, tn0, tf0);
                    box_net(b0, b1, b2, b3, b4, b5, tn1, tf1);
                }
                const uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
                const unsigned long long m0 = __builtin_amdgcn_fcmpf(tn0, tf0, 5), m1 = __builtin_amdgcn_fcmpf(tn1, tf1, 5);
                const unsigned long long m_right = m1 & (~m0 | __builtin_amdgcn_fcmpf(tn1, tn0, 4));
                const bool both = __builtin_amdgcn_inverse_ballot_w64(m0 & m1);
                const bool go_right = __builtin_amdgcn_inverse_ballot_w64(m_right);
                if (both) {
                    stack_push(stk, ovf, lds_entries, sp,
                               make_uint2(go_right ? lref : rref, __float_as_uint(go_right ? tn0 : tn1)));
                    ++sp;
                }
                const float near = go_right ? tn1 : tn0;
                const bool popn = __builtin_amdgcn_inverse_ballot_w64(~(m0 | m1) | __builtin_amdgcn_fcmpf(near, best, 2));
                cur = popn ? kPop : (go_right ? rref : lref);
            } else if ((int)cur < 0) {
                // one triangle (triangle.cpp:4-54); the leaf ref itself carries the progress
                const float4 a = q0, b = q1;
                const float e2z = q2.x;
                if (COUNT) {
                    if (P.depth == 0) c0.tris++;
                    else c1.tris++;
                }
                const float e1x = a.w, e1y = b.x, e1z = b.y, e2x = b.z, e2y = b.w;
                float pvx, pvy, pvz;
                cross3(P.dx, P.dy, P.dz, e2x, e2y, e2z, pvx, pvy, pvz);
                const float det = dot3(e1x, e1y, e1z, pvx, pvy, pvz);
                const bool parallel = fabsf(det) <= 9.99999993922529e-09f;
                const float inv_det = 1.0f / det;
                const float tx = P.ox - a.x, ty = P.oy - a.y, tz = P.oz - a.z;
                const float u = dot3(tx, ty, tz, pvx, pvy, pvz) * inv_det;
                const bool u_out = (u < 0.0f) || (u > 1.0f);
                float qx, qy, qz;
                cross3(tx, ty, tz, e1x, e1y, e1z, qx, qy, qz);
                const float v = dot3(P.dx, P.dy, P.dz, qx, qy, qz) * inv_det;
                const bool v_out = (v < 0.0f) || (u + v > 1.0f);
                const float dist = dot3(e2x, e2y, e2z, qx, qy, qz) * inv_det;
                const bool hit = !parallel && !u_out && !v_out && (dist > 0.0f);
                if (hit && dist < best) {  // strict <: first tested wins ties (bvh.cpp:90)
                    best = dist;
                    slot = (int)(cur & kLeafStartMask);
                }
                cur = (((cur >> kLeafCountShift) & 31u) == 1u) ? kPop : cur + (1u - (1u << kLeafCountShift));
            }
            if (cur == kPop) {
                // pop until an entry passes `near > t` (bvh.cpp:69); the bottom entry always does and
                // ends the traversal: the lane then waits for shading
                uint2 e;
                do {
                    --sp;
                    e = stack_pop(stk, ovf, lds_entries, sp);
                } while (__uint_as_float(e.y) > best);
                cur = e.x == kBottom ? kIdle : e.x;
            }
        };
        if (__builtin_amdgcn_ballot_w64(exact && cur != kIdle) != 0) {
#pragma unroll 1
            for (int act = 0; act < 8; ++act) step(std::true_type{});
        } else {
#pragma unroll 1
            for (int act = 0; act < 8; ++act) step(std::false_type{});
        }
    }
    tally_flush<COUNT>(ctr, tl, c0, c1);
}




// ---------------------------------------------------------------------------
// k_camera_tables — every camera ray of a frame starts at the same origin, so
// the origin-dependent half of both intersection tests is the same for all of
// them.  Per frame this kernel writes copies of the node and triangle records
// with that half applied:
//   nodes : (lo - o), (hi - o) of both children     -> the slab test is 6 multiplies per box
//   tris  : e1, e2, tvec = o - v0, qvec = cross(tvec, e1), C = dot(e2, qvec)
// Each value is produced by the same float operation on the same inputs as the
// per-ray code (bbox.cpp:72-73, triangle.cpp:30,36,42), so results stay bit-identical.
// ---------------------------------------------------------------------------
__global__ void k_camera_tables(SceneDev sc, uint32_t n_inner, float ox, float oy, float oz,
                                float4 *__restrict__ cam_inner, float4 *__restrict__ cam_tris) {
    const float4 *__restrict__ inner = (const float4 *)sc.inner;
    const float4 *__restrict__ tris = (const float4 *)sc.tris;
    const uint32_t total = n_inner + sc.ntris;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < n_inner) {
            const float4 q0 = inner[i * 4], q1 = inner[i * 4 + 1], q2 = inner[i * 4 + 2], q3 = inner[i * 4 + 3];
            // (lo - o), (hi - o) of both children, then one copy per direction octant (bit a: 1/d_a < 0)
            // with each axis' near plane first: for a finite non-NaN reciprocal, min(lo*inv, hi*inv) is
            // the product with lo when inv >= 0 and with hi otherwise (rounding is monotonic), so a wave
            // whose rays share the octant needs 6 multiplies and one max3/min3 per box, no min/max pairs.
            // Octant 0 is the plain (lo, hi) layout that the per-lane path reads.
            const float l0x = q0.x - ox, l0y = q0.y - oy, l0z = q0.z - oz, h0x = q0.w - ox, h0y = q1.x - oy, h0z = q1.y - oz;
            const float l1x = q1.z - ox, l1y = q1.w - oy, l1z = q2.x - oz, h1x = q2.y - ox, h1y = q2.z - oy, h1z = q2.w - oz;
            for (uint32_t o = 0; o < 8; ++o) {
                const bool sx = o & 1u, sy = o & 2u, sz = o & 4u;
                float4 *rec = cam_inner + ((size_t)o * n_inner + i) * 4;
                rec[0] = make_float4(sx ? h0x : l0x, sy ? h0y : l0y, sz ? h0z : l0z, sx ? l0x : h0x);
                rec[1] = make_float4(sy ? l0y : h0y, sz ? l0z : h0z, sx ? h1x : l1x, sy ? h1y : l1y);
                rec[2] = make_float4(sz ? h1z : l1z, sx ? l1x : h1x, sy ? l1y : h1y, sz ? l1z : h1z);
                rec[3] = q3;
            }
        } else {
            const uint32_t t = i - n_inner;
            const float4 a = tris[t * 3], b = tris[t * 3 + 1], c = tris[t * 3 + 2];
            const float e1x = a.w, e1y = b.x, e1z = b.y, e2x = b.z, e2y = b.w, e2z = c.x;
            const float tx = ox - a.x, ty = oy - a.y, tz = oz - a.z;
            float qx, qy, qz;
            cross3(tx, ty, tz, e1x, e1y, e1z, qx, qy, qz);
            const float cdot = dot3(e2x, e2y, e2z, qx, qy, qz);
            cam_tris[t * 4] = make_float4(e1x, e1y, e1z, e2x);
            cam_tris[t * 4 + 1] = make_float4(e2y, e2z, tx, ty);
            cam_tris[t * 4 + 2] = make_float4(tz, qx, qy, qz);
            cam_tris[t * 4 + 3] = make_float4(cdot, c.y, 0.f, 0.f);
        }
    }
}

// ---------------------------------------------------------------------------
// k_raygen — camera rays of one pass (pathtracer.cpp:251-280), written to rayA[pid] as
// (direction, flag word): all camera rays share the frame's origin, so 16 bytes per ray suffice.
// Slots without a sample (past the pixel's last one, padding) get the word ~0 and are skipped downstream;
// the others the two step_bits of their first Radiance step.
// ---------------------------------------------------------------------------
// LIVE 0: every camera ray of the pass, rayA[pid].   LIVE 1 (VMX_SAMPLING_ELIDE_DEAD): nothing is written but one word
// of live bits and its popcount per 64 consecutive path ids; launch_live_compact turns those into the ordered list of
// live path ids and k_raygen_live writes their rays, densely, to rayA[position in that list].
template <int LIVE>
__global__ void __launch_bounds__(256)
k_raygen(SceneDev sc, FrameDev fr, WorkDev wk, PixelStateDev px, PathArrays pa) {
    const uint32_t total = wk.samples * wk.n_pad;
    if (wk.pixel_major && (wk.samples & 63u) == 0) {
        // a wave's 64 path ids are 64 samples of ONE pixel: slot, pixel and cursor are wave-uniform — one division and one
        // scalar fetch each per wave instead of per lane (round 3: 3.2 -> 2.9 ms for the 530.8 M rays of the bench frame)
        // Round 4: a wave takes a whole PIXEL — all its samples, 64 at a time — so that the two dependent scalar fetches
        // (active[slot], then the pixel's cursor) and the pixel half of the stream key are paid once per pixel, not once
        // per 64 samples: the kernel was waiting on those fetches for 58 % of its wave cycles (3.2 -> 2.0 ms for the bench frame)
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t waves = gridDim.x * (blockDim.x >> 6);
        const uint32_t chunks = wk.samples >> 6;
        for (uint32_t s_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
             s_idx < wk.n_pad; s_idx += waves) {
            const uint32_t pid_base = s_idx * wk.samples;
            if (s_idx < wk.n_active) {
                const uint32_t lp = wk.active[s_idx];
                const uint32_t pixel = global_pixel(fr, lp);
                const PixelKey pk = rng_pixel_key(fr.seed, pixel);
                // (the step bits are drawn at all only where a kernel reads them: fr.camera_bits / LIVE)
                const bool lights = (LIVE || fr.camera_bits) && pixel_may_reach_a_light(sc, fr, pixel);
                const uint32_t cur = px.cursor[lp];  // (sample_index, with the cursor fetched once)
                const uint32_t k0 = cur & ~kCursorStrided, kstep = (cur & kCursorStrided) ? fr.quarter : 1u;
                for (uint32_t ch = 0; ch < chunks; ++ch) {
                    const uint32_t j = ch * 64u + lane;
                    const uint32_t k = (fr.lead != 0 && j >= fr.lead) ? (j - fr.lead + 1u) * fr.quarter : k0 + j * kstep;
                    bool live = false;
                    float dx = 0.f, dy = 0.f, dz = 0.f;
                    uint32_t depth = 0xFFFFFFFFu;
                    if (k < fr.kmax) {
                        Rng rng;
                        if (LIVE && !lights) {
                            // no ray of this pixel can meet a light: whether the path is live is decided by its draws
                            // alone — the stream keyed, the two jitter draws skipped, the step's three numbers drawn —
                            // and the ray itself is formed later, for the live paths only (k_raygen_live)
                            rng_init_keyed(rng, pk, k);
                            (void)rng_next(rng);
                            (void)rng_next(rng);
                            live = step_bits<false>(sc, fr.r2scale, rng, 0, fr.px, fr.py, fr.pz, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, false) != 3u;
                        } else {
                            primary_ray_keyed(fr, pixel, pk, k, rng, dx, dy, dz);
                            if (LIVE) live = !step_is_dead<false>(sc, fr.r2scale, rng, 0, fr.px, fr.py, fr.pz, dx, dy, dz, 1.f, 1.f, 1.f);
                            else depth = fr.camera_bits ? step_bits<false>(sc, fr.r2scale, rng, 0, fr.px, fr.py, fr.pz, dx, dy, dz, 1.f, 1.f, 1.f, lights) : 0u;
                        }
                    }
                    if (!LIVE) ((float4 *)pa.rayA)[pid_base + j] = make_float4(dx, dy, dz, __uint_as_float(depth));
                    if (LIVE) {
                        const unsigned long long bits = __builtin_amdgcn_ballot_w64(live);
                        if (lane == 0) wk.live_mask[(pid_base >> 6) + ch] = bits, wk.live_cnt[(pid_base >> 6) + ch] = (uint32_t)__popcll(bits);
                    }
                }
            } else {
                // padding slots of the pass: marked like sample slots past a pixel's last sample
                for (uint32_t ch = 0; ch < chunks; ++ch) {
                    if (!LIVE) ((float4 *)pa.rayA)[pid_base + ch * 64u + lane] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
                    if (LIVE && lane == 0) wk.live_mask[(pid_base >> 6) + ch] = 0ull, wk.live_cnt[(pid_base >> 6) + ch] = 0u;
                }
            }
        }
        return;
    }
    // (n_pad, and with it total, is a multiple of 64: a wave's 64 lanes hold 64 consecutive path ids and run the body together)
    for (uint32_t pid = blockIdx.x * blockDim.x + threadIdx.x; pid < total; pid += gridDim.x * blockDim.x) {
        uint32_t j, s_idx;
        if (wk.pixel_major) s_idx = fast_div(pid, wk.div_samples), j = pid - s_idx * wk.samples;
        else j = pid / wk.n_pad, s_idx = pid - j * wk.n_pad;
        uint32_t pid2, pixel, k;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        uint32_t depth = 0xFFFFFFFFu;
        bool live = false;
        if (s_idx < wk.n_active && primary_item(fr, wk, px, j, s_idx, pid2, pixel, k)) {
            Rng rng;
            primary_ray(fr, pixel, k, rng, dx, dy, dz);
            if (LIVE) live = !step_is_dead<false>(sc, fr.r2scale, rng, 0, fr.px, fr.py, fr.pz, dx, dy, dz, 1.f, 1.f, 1.f);
            else depth = fr.camera_bits ? step_bits<false>(sc, fr.r2scale, rng, 0, fr.px, fr.py, fr.pz, dx, dy, dz, 1.f, 1.f, 1.f) : 0u;
        }
        // camera rays share the origin (FrameDev): one 16-byte record in rayA — direction, and a word that is ~0 for a
        // slot without a sample (past the pixel's last one, or padding), else the step's step_bits
        if (!LIVE) ((float4 *)pa.rayA)[pid] = make_float4(dx, dy, dz, __uint_as_float(depth));
        if (LIVE) {
            const unsigned long long bits = __builtin_amdgcn_ballot_w64(live);
            if ((threadIdx.x & 63u) == 0) wk.live_mask[pid >> 6] = bits, wk.live_cnt[pid >> 6] = (uint32_t)__popcll(bits);
        }
    }
}

// the rays of the live camera paths, in list order: rayA[i] belongs to path live_ids[i]
__global__ void __launch_bounds__(256)
k_raygen_live(SceneDev sc, FrameDev fr, WorkDev wk, PixelStateDev px, PathArrays pa) {
    const uint32_t n = *wk.live_count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t pid = wk.live_ids[i];
        uint32_t j, s_idx;
        if (wk.pixel_major) s_idx = fast_div(pid, wk.div_samples), j = pid - s_idx * wk.samples;
        else j = pid / wk.n_pad, s_idx = pid - j * wk.n_pad;
        uint32_t pid2, pixel, k;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        uint32_t depth = 0xFFFFFFFFu;
        if (primary_item(fr, wk, px, j, s_idx, pid2, pixel, k)) {  // (always: the path was found live from the same item)
            Rng rng;
            primary_ray(fr, pixel, k, rng, dx, dy, dz);
            depth = fr.camera_bits ? step_bits<false>(sc, fr.r2scale, rng, 0, fr.px, fr.py, fr.pz, dx, dy, dz, 1.f, 1.f, 1.f) : 0u;
        }
        ((float4 *)pa.rayA)[i] = make_float4(dx, dy, dz, __uint_as_float(depth));
    }
}

// VMX_SAMPLING_ELIDE_DEAD: the n live camera paths of a pass are split into 8 bands of whole waves, one per work source
__device__ __forceinline__ uint32_t live_band(uint32_t n) { return ((n + 511u) / 512u) * 64u; }

// ---------------------------------------------------------------------------
// k_trace_q — the first form of the split wavefront's persistent traversal kernel
// (BVH::getIntersection only, bvh.cpp:47-145), now the instrumented build (visit
// counters, vmx_opts.collect_counters) and an A/B reference for k_trace_w below.
// Lane state is one ray; lanes refill themselves from the work source.
// Writes hit[pid] = (t, leaf slot).
//   SRC 0: camera rays written by k_raygen (direction in rayA, the frame's origin)
//   SRC 1: bounce rays of the queued path ids, read from rayA/rayB
// ---------------------------------------------------------------------------
template <bool COUNT, int SRC>
__global__ void __launch_bounds__(256, VMX_TRACE_WAVES_PER_SIMD)
k_trace_q(SceneDev sc, FrameDev fr, WorkDev wk, PixelStateDev px, PathArrays pa, DevCounters *ctr) {
    extern __shared__ uint2 lds_stack[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int lds_entries = (int)wk.lds_entries;
    uint2 *stk = lds_stack + (size_t)wave * (lds_entries + 1) * 64 + lane;
    uint2 *ovf = (uint2 *)wk.overflow_stack +
                 ((size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * wk.overflow_entries) * 64 + lane;
    // camera rays (SRC 0) read the per-frame origin-relative copies written by k_camera_tables
    const float4 *__restrict__ inner = SRC == 0 ? (const float4 *)wk.cam_inner : (const float4 *)sc.inner;
    const float4 *__restrict__ tris = SRC == 0 ? (const float4 *)wk.cam_tris : (const float4 *)sc.tris;
    float2 *__restrict__ hit_out = (float2 *)pa.hit;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t kReserve = wk.reserve;
    constexpr int kActionsPerCheck = 8;  // traversal actions between two scheduling checks
    const uint32_t nsrc = wk.nsrc, refill_min = wk.refill_min;
    const uint32_t band_slots = wk.band_slots, band_items = wk.band_items;
    const uint32_t root_ref = sc.root_ref;

    // wave-uniform scheduling state (kept in SGPRs through readfirstlane)
    uint32_t src = blockIdx.x % nsrc, res_lo = 0, res_hi = 0, tried = 0;
    bool exhausted = false;
    // lane state: one ray
    bool has = false, exact = false, counted = false;
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
    float ix = 0.f, iy = 0.f, iz = 0.f, best = 0.f, cur_near = 0.f;
    int slot = -1, sp = 0;
    uint32_t cur = 0, pid = 0;
    Cnt cn = {0, 0};

    for (;;) {
        // ---- refill idle lanes -------------------------------------------------------
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!has);
        if (idle != 0 && !exhausted && ((uint32_t)__popcll(idle) >= refill_min || idle == ~0ull)) {
            for (;;) {
                if (res_lo == res_hi) {
                    const uint32_t lim = SRC == 0 ? band_items : min(wk.qids.counts[src * 32], wk.qids.sub_capacity);
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&wk.heads[src * 32], kReserve);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= lim) {
                        src = (src + 1 == nsrc) ? 0 : src + 1;
                        if (++tried == nsrc) {
                            exhausted = true;
                            break;
                        }
                        continue;
                    }
                    tried = 0;
                    res_lo = base;
                    res_hi = min(base + kReserve, lim);
                }
                const unsigned long long want = __builtin_amdgcn_ballot_w64(!has);
                if (want == 0) break;
                const uint32_t avail = res_hi - res_lo;
                const uint32_t rank = (uint32_t)__popcll(want & lt_mask);
                const bool take = !has && rank < avail;
                const uint32_t item = res_lo + rank;
                res_lo = __builtin_amdgcn_readfirstlane(res_lo + min((uint32_t)__popcll(want), avail));
                if (take) {
                    bool valid = true;
                    if (SRC == 0) {
                        // band-local item -> path id; the ray was written by k_raygen
                        uint32_t j, s_idx;
                        if (wk.pixel_major) {
                            const uint32_t sl = fast_div(item, wk.div_samples);
                            j = item - sl * wk.samples, s_idx = src * band_slots + sl;
                        } else {
                            j = item / band_slots, s_idx = src * band_slots + (item - j * band_slots);
                        }
                        valid = s_idx < wk.n_active;
                        pid = path_id(wk, j, s_idx);
                    } else {
                        pid = wk.qids.ids[(size_t)src * wk.qids.sub_capacity + item];
                    }
                    if (valid) {
                        uint32_t depth;
                        if (SRC == 0) {  // camera ray: (direction, flag word), origin from the frame
                            const float4 a = ((const float4 *)pa.rayA)[pid];
                            ox = fr.px, oy = fr.py, oz = fr.pz, dx = a.x, dy = a.y, dz = a.z;
                            depth = __float_as_uint(a.w) == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u;  // (else: step_bits)
                        } else {
                            const float4 a = ((const float4 *)pa.state)[(size_t)pid * 4], b = ((const float4 *)pa.state)[(size_t)pid * 4 + 1];
                            ox = a.x, oy = a.y, oz = a.z, dx = a.w, dy = b.x, dz = b.y;
                            depth = __float_as_uint(b.z);
                        }
                        valid = depth != 0xFFFFFFFFu;  // k_raygen marks sample slots past the pixel's last sample
                        counted = (depth == 0u) || finite3(dx, dy, dz);
                    }
                    if (valid) {
                        has = true;
                        ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;  // Ray.h:10
                        exact = !(finite3(ix, iy, iz) && finite3(ox, oy, oz));
                        best = 999999999.f;  // bvh.cpp:48
                        slot = -1;
                        sp = 0;
                        cur = root_ref;
                        cur_near = -9999999.f;  // bvh.cpp:59
                    }
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(has) == 0) break;
        // ---- traversal actions --------------------------------------------------------------
        // every iteration runs the one-triangle step for the lanes at a leaf and the inner-node step
        // for the others (a vote between the two was measured and is no faster)
        for (int act = 0; act < kActionsPerCheck; ++act) {
            const bool at_leaf = has && (cur & kLeafBit) != 0;
            if (__builtin_amdgcn_ballot_w64(has) == 0) break;
            bool need_next = false, carry = false;
            if (at_leaf) {
                float det, inv_det, u, v, dist;
                if (SRC == 0) {
                    const uint32_t ti = (cur & kLeafStartMask) * 4;
                    const float4 a = tris[ti], b = tris[ti + 1], c = tris[ti + 2], e = tris[ti + 3];
                    // a = (e1, e2.x)  b = (e2.yz, tvec.xy)  c = (tvec.z, qvec)  e.x = dot(e2, qvec)
                    float pvx, pvy, pvz;
                    cross3(dx, dy, dz, a.w, b.x, b.y, pvx, pvy, pvz);
                    det = dot3(a.x, a.y, a.z, pvx, pvy, pvz);
                    inv_det = 1.0f / det;
                    u = dot3(b.z, b.w, c.x, pvx, pvy, pvz) * inv_det;
                    v = dot3(dx, dy, dz, c.y, c.z, c.w) * inv_det;
                    dist = e.x * inv_det;
                } else {
                    const uint32_t ti = (cur & kLeafStartMask) * 3;
                    const float4 a = tris[ti], b = tris[ti + 1], c = tris[ti + 2];
                    const float e1x = a.w, e1y = b.x, e1z = b.y, e2x = b.z, e2y = b.w, e2z = c.x;
                    float pvx, pvy, pvz;
                    cross3(dx, dy, dz, e2x, e2y, e2z, pvx, pvy, pvz);
                    det = dot3(e1x, e1y, e1z, pvx, pvy, pvz);
                    inv_det = 1.0f / det;
                    const float tx = ox - a.x, ty = oy - a.y, tz = oz - a.z;
                    u = dot3(tx, ty, tz, pvx, pvy, pvz) * inv_det;
                    float qx, qy, qz;
                    cross3(tx, ty, tz, e1x, e1y, e1z, qx, qy, qz);
                    v = dot3(dx, dy, dz, qx, qy, qz) * inv_det;
                    dist = dot3(e2x, e2y, e2z, qx, qy, qz) * inv_det;
                }
                if (COUNT && counted) cn.tris++;
                const bool parallel = fabsf(det) <= 9.99999993922529e-09f;
                const bool u_out = (u < 0.0f) || (u > 1.0f);
                const bool v_out = (v < 0.0f) || (u + v > 1.0f);
                const bool hit = !parallel && !u_out && !v_out && (dist > 0.0f);
                if (hit && dist < best) {  // strict <: first tested wins ties (bvh.cpp:90)
                    best = dist;
                    slot = (int)(cur & kLeafStartMask);
                }
                const uint32_t left = ((cur >> kLeafCountShift) & 31u) - 1u;
                need_next = left == 0;
                if (!need_next) cur = kLeafBit | (left << kLeafCountShift) | ((cur & kLeafStartMask) + 1u);
            }
            if (has && !at_leaf) {
                // The kernel is bound by the vector-memory pipe (4 x 1 KiB per wave and node visit).
                // Camera rays of one 8x8 tile mostly walk the same nodes: when every lane of this
                // step is at the same node, the record is fetched once through the scalar cache.
                float4 q0, q1, q2, q3;
                const uint32_t cur0 = __builtin_amdgcn_readfirstlane(cur);
                if (SRC == 0 && __builtin_amdgcn_ballot_w64(cur != cur0) == 0) {
                    // constant address space + wave-uniform address -> s_load_dwordx16
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    typedef const __attribute__((address_space(4))) f32x4 *scalar_ptr;
                    const scalar_ptr rec = (scalar_ptr)(uintptr_t)(inner + (size_t)cur0 * 4);
                    const f32x4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
                    q0 = make_float4(r0.x, r0.y, r0.z, r0.w);
                    q1 = make_float4(r1.x, r1.y, r1.z, r1.w);
                    q2 = make_float4(r2.x, r2.y, r2.z, r2.w);
                    q3 = make_float4(r3.x, r3.y, r3.z, r3.w);
                } else {
                    q0 = inner[cur * 4], q1 = inner[cur * 4 + 1], q2 = inner[cur * 4 + 2], q3 = inner[cur * 4 + 3];
                }
                if (COUNT && counted) cn.inner++;
                float tn0, tn1;
                bool h0, h1;
                if (SRC == 0) {
                    if (exact) {
                        h0 = box_exact_rel(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ix, iy, iz, tn0);
                        h1 = box_exact_rel(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ix, iy, iz, tn1);
                    } else {
                        h0 = box_fast_rel(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ix, iy, iz, tn0);
                        h1 = box_fast_rel(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ix, iy, iz, tn1);
                    }
                } else if (exact) {
                    h0 = box_exact(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ox, oy, oz, ix, iy, iz, tn0);
                    h1 = box_exact(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ox, oy, oz, ix, iy, iz, tn1);
                } else {
                    h0 = box_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ox, oy, oz, ix, iy, iz, tn0);
                    h1 = box_fast(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ox, oy, oz, ix, iy, iz, tn1);
                }
                const uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
                // bvh.cpp:103-132 without branches: both hit -> the strictly closer right child (else the
                // left) is next and the other one is pushed; one hit -> that child is next
                const bool both = h0 && h1;
                const bool go_right = both ? (tn1 < tn0) : h1;
                if (both) {
                    stack_push(stk, ovf, lds_entries, sp,
                               make_uint2(go_right ? lref : rref, __float_as_uint(go_right ? tn0 : tn1)));
                    ++sp;
                }
                cur = go_right ? rref : lref;
                cur_near = go_right ? tn1 : tn0;
                carry = h0 || h1;
                need_next = true;
            }
            if (need_next) {
                if (carry && cur_near > best) carry = false;  // bvh.cpp:69
                while (!carry) {
                    if (sp == 0) {
                        hit_out[pid] = make_float2(best, __int_as_float(slot));
                        has = false;
                        break;
                    }
                    --sp;
                    const uint2 e = stack_pop(stk, ovf, lds_entries, sp);
                    if (!(__uint_as_float(e.y) > best)) {
                        cur = e.x;
                        cur_near = __uint_as_float(e.y);
                        carry = true;
                    }
                }
            }
        }
    }
    if (COUNT) {
        const uint32_t iv = wave_sum(cn.inner), tt = wave_sum(cn.tris);
        if (lane == 0) {
            const int st = SRC == 0 ? 0 : 1;
            if (iv) atomicAdd(&ctr->stage[st].inner_visits, (unsigned long long)iv);
            if (tt) atomicAdd(&ctr->stage[st].tri_tests, (unsigned long long)tt);
        }
    }
}


// ---------------------------------------------------------------------------
// k_trace_w — k_trace_q's production form (no counters), written for the
// limit the SQ counters show this loop runs at: instruction delivery.  One
// traversal step of the first form executed ~62 VALU + ~47 SALU + ~12 branch
// instructions, and the instruction cache two CUs share was busy ~100 % of the
// kernel's cycles (SQC_ICACHE_BUSY_CYCLES; profiles/).  This form issues fewer of them:
//   * lane state is the node reference alone (idle = kIdle, "must pop" = kPop, bit 31 = leaf):
//     no has/carry/need_next lane masks to merge with SALU triples at every join;
//   * stack level 0 holds a bottom entry (ref kBottom, near = -inf): the pop loop is a
//     single do-while on `near > best` and a finished ray falls out of it;
//   * the NaN-exact box form is chosen per wave (a wave with any such ray runs the
//     exact form for all its lanes — both forms agree on finite data), not per lane;
//   * the min/max network of a box is one asm block (no hazard nops between the
//     pieces) and a scalar-fetched node record is multiplied straight from SGPRs;
//   * the all-lanes-idle test is made once per 8 steps, not every step.
// Each ray still performs exactly the reference's sequence of tests (bvh.cpp:47-145).
// ---------------------------------------------------------------------------
#ifdef VMX_STEP_PROFILE
#define VMX_DESCENT_COUNT "s_add_u32 %[done], %[done], 1\n\ts_nop 0\n\t"  // nodes done, for tools/step_profile.py
#else
#define VMX_DESCENT_COUNT "s_nop 1\n\t"  // the product does not count them: a scalar instruction is an issue slot
#endif
// ---------------------------------------------------------------------------
// uniform_descent — camera rays: the stretch of a traversal in which all traversing lanes of a wave are at
// the same inner node and take the same child, as one assembly loop.
//
// k_trace_w<0> is bound by instruction issue: on gfx950 a SIMD issues about one instruction per 2.2
// cycles whatever its kind — scalar ALU instructions do not overlap with vector ones — and a taken
// branch costs about three instructions (tools/sload_probe.hip, profiles/r02_sload_probe.txt); its
// counters add up to exactly that ((SQ_INSTS_VALU + SQ_INSTS_SALU) x 2.15 cycles = the kernel's SIMD
// cycles).  64 samples of one pixel walk the tree together, and the compiled step spends ~63
// instructions and two taken branches on such a node (lane dispatch, uniformity test, exec-masked push,
// lane masks merged with scalar ops) where the arithmetic is 16.  One iteration here is 39 instructions
// and the back edge:
//   * the 12 products and two max3/min3 pairs straight from the scalar-fetched record of the wave's
//     octant copy (near planes first, as the OCT route of the general step);
//   * a missed child gets the entry distance +inf, so "the right child is strictly closer, or the only
//     one hit" is one compare, near/far are a min and a max, and `near > t` (bvh.cpp:69) prunes a missed
//     child like a far one;
//   * the other child is stored above the stack top by every lane (no exec mask) and counts as an entry
//     (sp + 1) only where its distance is finite, i.e. where both children were hit (a child hit at +inf
//     is treated as missed: the reference would pop it again unvisited, `inf > t`);
//   * the next node is chosen on the scalar unit (s_cselect) and its record fetch issued at once.
// Round 3: when EVERY lane prunes at a node (both children missed, or beyond the hit so far — a wave's 64 samples of
// one pixel mostly agree on that too) the pop is done here as well, uniformly (label 5): 12 % of the kernel's
// wave-steps were whole waves leaving the loop only to pop and come back (tools/step_profile.py).
// While the lanes agree, none of them prunes and the next node is an inner node, the loop needs no
// per-lane node reference, no exec masking and no refill test (no lane can finish here).  When lanes
// disagree or one of them prunes, the node is finished per lane (same values, lane selects) and the
// general step goes on from there; a leaf, or a lane whose next push would leave the LDS levels, ends the
// loop as well.  Every ray performs exactly the tests of the general step, in the same order.
//   in: trav = the traversing lanes, scur = the inner node they all are at, sp < lds_entries in each of them
//   out: cur (per lane: next node, leaf or kPop), sp; returns the number of nodes done (diagnostic build; else 0)
// The record lives in s[36:51] (an inline-asm operand cannot be a 16-register tuple).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uniform_descent(int &sp, uint32_t &cur, float ix, float iy, float iz, float &best,
                                                    int &slot, float dx, float dy, float dz, uint32_t stk_lds,
                                                    const char *base, uint32_t tri_delta, uint32_t scur, int lds_entries,
                                                    unsigned long long trav) {
    float t0, t1, t2, n0, tf, n1, near, far;
    uint32_t va, oth, soff, soth, sidx, cnt, done = 0;
    unsigned long long tmp, pop;
    asm volatile(
        "s_mov_b64 exec, %[trav]\n\t"
        "s_lshl_b32 %[soff], %[scur], 6\n\t"
        "s_load_dwordx16 s[36:51], %[base], %[soff]\n"
        "1:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mul_f32_e32 %[t0], s36, %[ix]\n\t"
        "v_mul_f32_e32 %[t1], s37, %[iy]\n\t"
        "v_mul_f32_e32 %[t2], s38, %[iz]\n\t"
        "v_max3_f32 %[n0], %[t0], %[t1], %[t2]\n\t"
        "v_mul_f32_e32 %[t0], s39, %[ix]\n\t"
        "v_mul_f32_e32 %[t1], s40, %[iy]\n\t"
        "v_mul_f32_e32 %[t2], s41, %[iz]\n\t"
        "v_min3_f32 %[tf], %[t0], %[t1], %[t2]\n\t"
        "v_cmp_le_f32_e32 vcc, %[n0], %[tf]\n\t"              // left child hit (bbox.cpp:82)
        "v_mul_f32_e32 %[t0], s42, %[ix]\n\t"
        "v_mul_f32_e32 %[t1], s43, %[iy]\n\t"
        "v_mul_f32_e32 %[t2], s44, %[iz]\n\t"
        "v_cndmask_b32_e32 %[n0], %[inf], %[n0], vcc\n\t"  // its entry distance, +inf if missed
        "v_max3_f32 %[n1], %[t0], %[t1], %[t2]\n\t"
        "v_mul_f32_e32 %[t0], s45, %[ix]\n\t"
        "v_mul_f32_e32 %[t1], s46, %[iy]\n\t"
        "v_mul_f32_e32 %[t2], s47, %[iz]\n\t"
        "v_min3_f32 %[tf], %[t0], %[t1], %[t2]\n\t"
        "v_cmp_le_f32_e32 vcc, %[n1], %[tf]\n\t"              // right child hit
        VMX_DESCENT_COUNT  // (two wait states between the compare and the select on its mask either way)
        "v_cndmask_b32_e32 %[n1], %[inf], %[n1], vcc\n\t"
        "v_cmp_lt_f32_e32 vcc, %[n1], %[n0]\n\t"              // go right: strictly closer, or the only one hit (bvh.cpp:106)
        "v_min_f32_e32 %[near], %[n0], %[n1]\n\t"
        "v_max_f32_e32 %[far], %[n0], %[n1]\n\t"
        "v_cmp_gt_f32_e64 %[pop], %[near], %[best]\n\t"       // the child taken is pruned, or none was hit (bvh.cpp:69)
        "s_cmp_eq_u64 vcc, 0\n\t"                             // every lane goes left?
        "s_cselect_b32 %[scur], s48, s49\n\t"
        "s_cselect_b32 %[soth], s49, s48\n\t"
        "s_cselect_b64 %[tmp], exec, vcc\n\t"
        "s_andn2_b64 %[tmp], %[tmp], %[pop]\n\t"
        "s_cmp_eq_u64 %[tmp], exec\n\t"                       // all lanes the same way and none prunes
        "s_cbranch_scc0 2f\n\t"
        "v_mov_b32_e32 %[oth], %[soth]\n\t"
        "v_lshl_add_u32 %[va], %[sp], 9, %[stk]\n\t"
        "ds_write2_b32 %[va], %[oth], %[far] offset1:1\n\t"   // the other child, above the top
        "v_cmp_gt_f32_e32 vcc, %[inf], %[far]\n\t"        // ... an entry where both children were hit
        "s_lshl_b32 %[soff], %[scur], 6\n\t"
        "s_bitcmp1_b32 %[scur], 31\n\t"                       // the child taken is a leaf?
        "v_addc_co_u32_e32 %[sp], vcc, 0, %[sp], vcc\n\t"
        "s_cbranch_scc1 9f\n\t"
        "s_load_dwordx16 s[36:51], %[base], %[soff]\n\t"
        "v_cmp_le_i32_e32 vcc, %[lds], %[sp]\n\t"             // a lane whose next push would leave the LDS levels?
        "s_cbranch_vccz 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"                            // nothing may land in s[36:51] after the block
        "s_branch 3f\n"
        // ---- the lanes part ways here, or some of them prune
        "2:\n\t"
        "s_cmp_eq_u64 %[pop], exec\n\t"                     // EVERY lane prunes (the common case: both children missed, or
        "s_cbranch_scc1 5f\n\t"                            // beyond the hit so far): pop here, uniformly, as long as that works
        // finish this node per lane (bvh.cpp:103-132)
        "v_mov_b32_e32 %[t0], s48\n\t"
        "v_mov_b32_e32 %[t1], s49\n\t"
        "v_cndmask_b32_e32 %[oth], %[t1], %[t0], vcc\n\t"     // stored: go right ? left : right
        "v_cndmask_b32_e32 %[t2], %[t0], %[t1], vcc\n\t"      // taken
        "v_lshl_add_u32 %[va], %[sp], 9, %[stk]\n\t"
        "ds_write2_b32 %[va], %[oth], %[far] offset1:1\n\t"
        "v_mov_b32_e32 %[t0], 0x7ffffffd\n\t"
        "v_cmp_gt_f32_e32 vcc, %[inf], %[far]\n\t"
        "v_cndmask_b32_e64 %[cur], %[t2], %[t0], %[pop]\n\t"  // kPop where the lane prunes
        "s_nop 0\n\t"
        "v_addc_co_u32_e32 %[sp], vcc, 0, %[sp], vcc\n\t"
        "s_branch 4f\n"
        // ---- uniform pop (bvh.cpp:61-70).  The child taken is pruned in every lane, so the other child — never nearer —
        // would be popped and pruned at once: it is not pushed at all.  Each lane reads ITS entry below the top; the wave
        // goes on together while all lanes skip it (near > t), or all take it and it is the same inner node for all of
        // them (back into the loop) or the same leaf (out, at that leaf).  Anything else — lanes disagree, the bottom
        // entry (the ray is done) — leaves sp untouched and hands every lane to the general step as "must pop".
        "5:\n\t"
        "v_add_u32_e32 %[t0], -1, %[sp]\n\t"
        "v_lshl_add_u32 %[va], %[t0], 9, %[stk]\n\t"
        "ds_read_b32 %[t1], %[va]\n\t"                      // node reference of the entry
        "ds_read_b32 %[t2], %[va] offset:4\n\t"             // its entry distance
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_gt_f32_e32 vcc, %[t2], %[best]\n\t"          // lanes that skip it
        "s_cbranch_vccz 6f\n\t"
        "s_cmp_eq_u64 vcc, exec\n\t"
        "s_cbranch_scc0 7f\n\t"                            // some skip, some take: per lane from here
        "v_mov_b32_e32 %[sp], %[t0]\n\t"                    // all skip: next entry
        "s_branch 5b\n"
        "6:\n\t"
        "v_readfirstlane_b32 %[soth], %[t1]\n\t"
        "s_nop 1\n\t"
        "v_cmp_ne_u32_e32 vcc, %[soth], %[t1]\n\t"
        "s_cbranch_vccnz 7f\n\t"                           // not the same entry in every lane
        "s_cmp_ge_u32 %[soth], 0x7ffffffd\n\t"              // leaf (bit 31) or bottom entry: not an inner node
        "s_cbranch_scc1 8f\n\t"
        "v_mov_b32_e32 %[sp], %[t0]\n\t"
        "s_mov_b32 %[scur], %[soth]\n\t"
        "s_lshl_b32 %[soff], %[scur], 6\n\t"
        "s_load_dwordx16 s[36:51], %[base], %[soff]\n\t"
        "s_branch 1b\n"
        "8:\n\t"
        "s_bitcmp1_b32 %[soth], 31\n\t"
        "s_cbranch_scc0 7f\n\t"                            // the bottom entry: the general step finishes the rays
        "v_mov_b32_e32 %[sp], %[t0]\n\t"
        "s_mov_b32 %[scur], %[soth]\n"
        // ---- uniform leaf (triangle.cpp:4-54 on the camera-relative records of k_camera_tables): every traversing lane
        // tests every triangle of the leaf, the record comes through the scalar cache into s[36:51]
        //   s36-38 e1   s39-41 e2   s42-44 tvec = o - v0   s45-47 qvec = cross(tvec, e1)   s48 dot(e2, qvec)
        // same operations in the same order as the general step; then the uniform pop above
        "9:\n\t"
        "s_and_b32 %[sidx], %[scur], 0x3ffffff\n\t"         // leaf slot of the first triangle
        "s_bfe_u32 %[cnt], %[scur], 0x5001a\n\t"            // triangles in the leaf (5 bits at 26)
        "s_lshl_b32 %[soff], %[sidx], 6\n\t"
        "s_add_u32 %[soff], %[soff], %[tdelta]\n\t"
        "s_load_dwordx16 s[36:51], %[base], %[soff]\n"
        "10:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mul_f32_e32 %[t0], s41, %[dy]\n\t"               // pvec = cross(d, e2)
        "v_mul_f32_e32 %[t1], s40, %[dz]\n\t"
        "v_sub_f32_e32 %[n0], %[t0], %[t1]\n\t"
        "v_mul_f32_e32 %[t0], s39, %[dz]\n\t"
        "v_mul_f32_e32 %[t1], s41, %[dx]\n\t"
        "v_sub_f32_e32 %[tf], %[t0], %[t1]\n\t"
        "v_mul_f32_e32 %[t0], s40, %[dx]\n\t"
        "v_mul_f32_e32 %[t1], s39, %[dy]\n\t"
        "v_sub_f32_e32 %[n1], %[t0], %[t1]\n\t"
        "v_mul_f32_e32 %[t0], s36, %[n0]\n\t"               // det = dot(e1, pvec)
        "v_mul_f32_e32 %[t1], s37, %[tf]\n\t"
        "v_add_f32_e32 %[t0], %[t0], %[t1]\n\t"
        "v_mul_f32_e32 %[t1], s38, %[n1]\n\t"
        "v_add_f32_e32 %[near], %[t0], %[t1]\n\t"
        "v_div_scale_f32 %[t0], %[pop], %[near], %[near], 1.0\n\t"   // inv_det = 1 / det, correctly rounded
        "v_rcp_f32_e32 %[t1], %[t0]\n\t"
        "v_div_scale_f32 %[t2], vcc, 1.0, %[near], 1.0\n\t"
        "v_fma_f32 %[far], -%[t0], %[t1], 1.0\n\t"
        "v_fmac_f32_e32 %[t1], %[far], %[t1]\n\t"
        "v_mul_f32_e32 %[far], %[t2], %[t1]\n\t"
        "v_fma_f32 %[oth], -%[t0], %[far], %[t2]\n\t"
        "v_fmac_f32_e32 %[far], %[oth], %[t1]\n\t"
        "v_fma_f32 %[t0], -%[t0], %[far], %[t2]\n\t"
        "v_div_fmas_f32 %[t0], %[t0], %[t1], %[far]\n\t"
        "v_div_fixup_f32 %[t0], %[t0], %[near], 1.0\n\t"
        "v_mul_f32_e32 %[t1], s42, %[n0]\n\t"               // u = dot(tvec, pvec) * inv_det
        "v_mul_f32_e32 %[t2], s43, %[tf]\n\t"
        "v_add_f32_e32 %[t1], %[t1], %[t2]\n\t"
        "v_mul_f32_e32 %[t2], s44, %[n1]\n\t"
        "v_add_f32_e32 %[t1], %[t1], %[t2]\n\t"
        "v_mul_f32_e32 %[n0], %[t1], %[t0]\n\t"
        "v_mul_f32_e32 %[t1], s45, %[dx]\n\t"               // v = dot(d, qvec) * inv_det
        "v_mul_f32_e32 %[t2], s46, %[dy]\n\t"
        "v_add_f32_e32 %[t1], %[t1], %[t2]\n\t"
        "v_mul_f32_e32 %[t2], s47, %[dz]\n\t"
        "v_add_f32_e32 %[t1], %[t1], %[t2]\n\t"
        "v_mul_f32_e32 %[tf], %[t1], %[t0]\n\t"
        "v_mul_f32_e32 %[n1], s48, %[t0]\n\t"               // t = dot(e2, qvec) * inv_det
        "v_add_f32_e32 %[t1], %[n0], %[tf]\n\t"             // u + v
        "v_and_b32_e32 %[t2], 0x7fffffff, %[near]\n\t"      // |det|
        // the next triangle's record can be on its way while this one is judged
        "s_cmp_lg_u32 %[cnt], 1\n\t"
        "s_cbranch_scc0 11f\n\t"
        "s_add_u32 %[soff], %[soff], 64\n\t"
        "s_load_dwordx16 s[36:51], %[base], %[soff]\n"
        "11:\n\t"
        "s_mov_b64 %[tmp], exec\n\t"
        "v_cmpx_nge_f32_e32 vcc, 0x322bcc77, %[t2]\n\t"     // not (|det| <= float(1e-8))           triangle.cpp:25
        "v_cmpx_ngt_f32_e32 vcc, 0, %[n0]\n\t"              // not (u < 0)                          :34
        "v_cmpx_nlt_f32_e32 vcc, 1.0, %[n0]\n\t"            // not (u > 1)
        "v_cmpx_ngt_f32_e32 vcc, 0, %[tf]\n\t"              // not (v < 0)                          :40
        "v_cmpx_nlt_f32_e32 vcc, 1.0, %[t1]\n\t"            // not (u + v > 1)
        "v_cmpx_lt_f32_e32 vcc, 0, %[n1]\n\t"               // t > 0                                :44
        "v_cmpx_lt_f32_e32 vcc, %[n1], %[best]\n\t"         // strictly nearer: the first tested wins ties (bvh.cpp:90)
        "v_mov_b32_e32 %[best], %[n1]\n\t"
        "v_mov_b32_e32 %[slot], %[sidx]\n\t"
        "s_mov_b64 exec, %[tmp]\n\t"
        "s_add_u32 %[sidx], %[sidx], 1\n\t"
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"
        "s_cmp_lg_u32 %[cnt], 0\n\t"
        "s_cbranch_scc1 10b\n\t"
        "s_branch 5b\n"
        "7:\n\t"
        "v_mov_b32_e32 %[cur], 0x7ffffffd\n\t"              // kPop
        "s_branch 4f\n"
        "3:\n\t"
        "v_mov_b32_e32 %[cur], %[scur]\n"
        "4:\n\t"
        "s_mov_b64 exec, -1"
        : [sp] "+v"(sp), [cur] "+v"(cur), [best] "+v"(best), [slot] "+v"(slot), [scur] "+s"(scur), [done] "+s"(done),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [n0] "=&v"(n0), [tf] "=&v"(tf), [n1] "=&v"(n1),
          [near] "=&v"(near), [far] "=&v"(far), [va] "=&v"(va), [oth] "=&v"(oth), [soff] "=&s"(soff), [soth] "=&s"(soth),
          [sidx] "=&s"(sidx), [cnt] "=&s"(cnt), [tmp] "=&s"(tmp), [pop] "=&s"(pop)
        : [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz), [stk] "v"(stk_lds),
          [base] "s"(base), [tdelta] "s"(tri_delta), [lds] "s"(lds_entries), [trav] "s"(trav), [inf] "v"(__builtin_inff())
        : "vcc", "scc", "memory", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47",
          "s48", "s49", "s50", "s51");
    return done;
}

#ifdef VMX_STEP_PROFILE
#include "vmx_step_profile.inc"
#else
#define VMX_PROF_STEP(EX, OC) step(EX, OC)
#endif

// LIVE (SRC 0, VMX_SAMPLING_ELIDE_DEAD): the work is the dense list of live camera paths — ray and hit record of list
// entry i sit at rayA[i] / hit[i] — cut into 8 bands of whole waves (live_band)
// SORT (classified output, DESIGN_HISTORY.md 5.1): instead of a hit record per ray, the finished rays of a wave are sorted on the
// spot when the wave refills.  A ray whose Radiance step is the path's last one by the path's own draws (the bits that
// came with the ray) and that cannot meet a light sphere needs nothing more: it is counted here and forgotten.  The
// others get a 32-byte record (direction, flag word | t, leaf slot, position) appended to a dense list — the wave
// reserves 256 entries at a time with one atomic, fills them to the last one, and pads the tail of its last chunk with
// position = ~0 when it ends — which k_shade reads front to back.
constexpr uint32_t kOutChunk = 256;
template <int SRC, bool LIVE = false, bool SORT = false>
__global__ void __launch_bounds__(256, VMX_TRACE_WAVES_PER_SIMD) __attribute__((amdgpu_num_sgpr(VMX_TRACE_SGPRS)))
k_trace_w(SceneDev sc, FrameDev fr, WorkDev wk, PathArrays pa) {
    extern __shared__ uint2 lds_stack[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int lds_entries = (int)wk.lds_entries;
#ifdef VMX_STEP_PROFILE
    __shared__ unsigned long long s_prof[4][10][2];
    if (lane < 20) s_prof[wave][lane >> 1][lane & 1] = 0;
    unsigned long long prof_t = __builtin_readcyclecounter();
    if (lane == 0) atomicMin(&g_wave_span[SRC][0], (unsigned long long)wall_clock64());
#endif
    uint2 *stk = lds_stack + (size_t)wave * (lds_entries + 1) * 64 + lane;
    uint2 *ovf = (uint2 *)wk.overflow_stack +
                 ((size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * wk.overflow_entries) * 64 + lane;
    const float4 *__restrict__ inner = SRC == 0 ? (const float4 *)wk.cam_inner : (const float4 *)sc.inner;
    const float4 *__restrict__ tris = SRC == 0 ? (const float4 *)wk.cam_tris : (const float4 *)sc.tris;
    float2 *__restrict__ hit_out = (float2 *)pa.hit;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t kReserve = wk.reserve;
    constexpr uint32_t kIdle = 0x7FFFFFFFu, kBottom = 0x7FFFFFFEu, kPop = 0x7FFFFFFDu;  // never valid inner indices
    const uint32_t nsrc = wk.nsrc, refill_min = wk.refill_min;
    const uint32_t band_slots = wk.band_slots, band_items = wk.band_items;
    const uint32_t root_ref = sc.root_ref;

    uint32_t src = blockIdx.x % nsrc, res_lo = 0, res_hi = 0, tried = 0;
    bool exhausted = false;
    bool exact = false;
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
    float ix = 0.f, iy = 0.f, iz = 0.f, best = 0.f;
    int slot = -1, sp = 0;
    uint32_t cur = kIdle, pid = 0;
    // SORT: the ray's flag word; bit 31 set while the lane holds a finished ray that has not been sorted yet
    uint32_t rflags = 0;
    uint32_t out_lo = 0, out_hi = 0, n_rays = 0, n_hits = 0;  // (wave-uniform) reserved list entries; rays / triangle hits settled here
    // camera rays: direction octant of the wave's rays if they all share it (else 8): selects the
    // per-octant copy of the node table (k_camera_tables) for wave-uniform steps
    uint32_t wave_octant = 8;
    stk[0] = make_uint2(kBottom, 0xFF800000u);  // bottom entry; pushes start at level 1, so it stays

    // SORT: settle the finished rays the wave's lanes hold (wave-uniform code: called before a refill and at the end)
    auto sort_finished = [&]() {
        const bool fin = (rflags >> 31) != 0;
        const bool material = slot >= 0;
        // ends by its draws (bit 0 / 1 by the material flag) and no light sphere in reach (bit 2): nothing to shade
        // (wk.keep_all: nothing is settled here — every ray gets its record and its full RayCast in k_shade)
        const bool done = !wk.keep_all && fin && (((material ? rflags : rflags >> 1) & 1u) != 0) && (rflags & 4u) == 0;
        // (a bounce "ray" with a non-finite direction is traced like the others but is not counted as a ray: tally_add)
        const bool counted = SRC == 0 || finite3(dx, dy, dz);
        n_rays += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(done && counted));
        n_hits += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(done && counted && material));
        const bool todo = fin && !done;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(todo);
        const uint32_t n = (uint32_t)__popcll(m);
        float4 *__restrict__ rec = (float4 *)wk.out_rec;
        if (n != 0) {
            constexpr uint32_t kRec = SRC == 0 ? 2 : 1;  // float4s per entry; the last one holds (t, leaf slot, position, -)
            // the entries fill the wave's chunk to its end and go on in a new one: nothing is left unused but the tail of
            // the wave's last chunk (padded at the end), so the list never holds more than rays + 255 per wave
            const uint32_t rem = out_hi - out_lo, rank = (uint32_t)__popcll(m & lt_mask);
            uint32_t at = out_lo + rank;
            if (n > rem) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(wk.out_count, kOutChunk);
                base = __builtin_amdgcn_readfirstlane(base);
                if (rank >= rem) at = base + (rank - rem);
                out_lo = base + (n - rem);
                out_hi = base + kOutChunk;
            } else {
                out_lo += n;
            }
            // (the list is sized for every path of the pass + one chunk per wave, so `at` cannot reach its end; an entry
            // that would is dropped and reported instead of written — never a store past the allocation)
            if (todo && at < wk.out_capacity) {
                if (SRC == 0) rec[(size_t)at * kRec] = make_float4(dx, dy, dz, __uint_as_float(rflags & 7u));  // (bounce paths: ray and stream are in state[pid])
                rec[(size_t)at * kRec + (kRec - 1)] = make_float4(best, __int_as_float(slot), __uint_as_float(pid), 0.f);
            } else if (todo) {
                atomicAdd(&wk.out_ctr->overflow, 1ull);
            }
        }
        rflags &= 0x7FFFFFFFu;
    };

    // One traversal step of every active lane.  The NaN-exact box form runs only while one of the
    // wave's rays needs it.
    // base of the node table the wave-uniform steps of this batch read: the wave's octant copy if its
    // rays share an octant (OCT), else the plain copy; records are addressed by a 32-bit byte offset
    const char *uni_base = (const char *)inner;
    auto step = [&](auto exact_tag, auto oct_tag) {
        constexpr bool EXACT = decltype(exact_tag)::value;  // NaN-exact box form
        constexpr bool OCT = decltype(oct_tag)::value;      // uniform steps read near-plane-first records
        if constexpr (SRC == 1) {
            // ---- bounce rays: ONE fetch phase per step.  Incoherent rays put inner-node lanes and leaf lanes in
            // the same wave at every step; fetched separately (node loads, wait, box math; triangle loads, wait,
            // triangle math) a wave-step pays two dependent memory round trips, and this kernel is bound by that
            // latency (SQ_WAIT_ANY 67 % of its wave cycles, 5 waves per SIMD; profiles/).  Inner records and
            // triangle records live in one allocation (SceneDev::tri_off), so every traversing lane forms a
            // 32-bit byte offset to ITS record, all of them load 56 bytes in one set of four loads, and the wave
            // waits once.  Same tests per ray in the same order (bvh.cpp:47-145).
            const bool leaf = (int)cur < 0;
            float4 q0, q1, q2, q3;
            // byte offset of this lane's record (0 while the lane is not traversing)
            // (round 3: the same transposition through LDS — 4 ds_write_b128 + 4 ds_read_b128 instead of the 32 DPP
            // selects — was measured in the kernel as well: its 4 KB of staging per wave cost waves or LDS stack levels,
            // 50.5 ms for the bounce stage against 42.9; profiles/r03_lds_transpose.txt, tools/ta_probe.hip)
            quad_fetch_record((const char *)inner,
                              cur == kIdle ? 0u : (leaf ? sc.tri_off + (cur & kLeafStartMask) * 48u : (cur << 6)),
                              lane, q0, q1, q2, q3);
            if (cur != kIdle) {
                if (!leaf) {
                    const float a0 = (q0.x - ox) * ix, a1 = (q0.y - oy) * iy, a2 = (q0.z - oz) * iz;
                    const float a3 = (q0.w - ox) * ix, a4 = (q1.x - oy) * iy, a5 = (q1.y - oz) * iz;
                    const float b0 = (q1.z - ox) * ix, b1 = (q1.w - oy) * iy, b2 = (q2.x - oz) * iz;
                    const float b3 = (q2.y - ox) * ix, b4 = (q2.z - oy) * iy, b5 = (q2.w - oz) * iz;
                    float tn0, tf0, tn1, tf1;
                    if (EXACT) {
                        box_net_exact(a0, a1, a2, a3, a4, a5, tn0, tf0);
                        box_net_exact(b0, b1, b2, b3, b4, b5, tn1, tf1);
                    } else {
                        box_net(a0, a1, a2, a3, a4, a5, tn0, tf0);
                        box_net(b0, b1, b2, b3, b4, b5, tn1, tf1);
                    }
                    const uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
                    const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
                    const bool both = h0 && h1;
                    const bool go_right = h1 && (!h0 || tn1 < tn0);  // both: the strictly closer right child; one: that child
                    if (both) {
                        const uint2 e = make_uint2(go_right ? lref : rref, __float_as_uint(go_right ? tn0 : tn1));
                        stack_push(stk, ovf, lds_entries, sp, e);
                        ++sp;
                    }
                    const float near = go_right ? tn1 : tn0;
                    // no child hit, or the child taken directly fails `near > t` (bvh.cpp:69): pop
                    cur = (!(h0 || h1) || near > best) ? kPop : (go_right ? rref : lref);
                } else {
                    // one triangle of the leaf (triangle.cpp:4-54): q0 = (v0, e1.x) q1 = (e1.yz, e2.xy) q2.x = e2.z
                    const float e1x = q0.w, e1y = q1.x, e1z = q1.y, e2x = q1.z, e2y = q1.w, e2z = q2.x;
                    float pvx, pvy, pvz;
                    cross3(dx, dy, dz, e2x, e2y, e2z, pvx, pvy, pvz);
                    const float det = dot3(e1x, e1y, e1z, pvx, pvy, pvz);
                    const float inv_det = 1.0f / det;
                    const float tx = ox - q0.x, ty = oy - q0.y, tz = oz - q0.z;
                    const float u = dot3(tx, ty, tz, pvx, pvy, pvz) * inv_det;
                    float qx, qy, qz;
                    cross3(tx, ty, tz, e1x, e1y, e1z, qx, qy, qz);
                    const float v = dot3(dx, dy, dz, qx, qy, qz) * inv_det;
                    const float dist = dot3(e2x, e2y, e2z, qx, qy, qz) * inv_det;
                    const bool parallel = fabsf(det) <= 9.99999993922529e-09f;
                    const bool u_out = (u < 0.0f) || (u > 1.0f);
                    const bool v_out = (v < 0.0f) || (u + v > 1.0f);
                    const bool hit = !parallel && !u_out && !v_out && (dist > 0.0f);
                    if (hit && dist < best) {  // strict <: first tested wins ties (bvh.cpp:90)
                        best = dist;
                        slot = (int)(cur & kLeafStartMask);
                    }
                    cur = (((cur >> kLeafCountShift) & 31u) == 1u) ? kPop : cur + (1u - (1u << kLeafCountShift));
                }
            }
        } else
        // inner references are the values below kPop; leaf references have bit 31; kPop/kBottom/kIdle lie between
        if (cur < kPop) {
            // ---- inner node: both children boxes (bbox.cpp:70-83), nearer child first (bvh.cpp:103-132)
            float tn0, tf0, tn1, tf1;
            uint32_t lref, rref;
            const uint32_t cur0 = __builtin_amdgcn_readfirstlane(cur);
            if (SRC == 0 && !EXACT && __builtin_amdgcn_ballot_w64(cur != cur0) == 0) {
                // every lane of this step is at the same node: one fetch through the scalar cache,
                // and the products take the record straight from SGPRs
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                typedef const __attribute__((address_space(4))) f32x4 *scalar_ptr;
                const scalar_ptr rec = (scalar_ptr)(uintptr_t)(uni_base + (uint32_t)(cur0 << 6));
                const f32x4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
                if (OCT) {  // near planes first in this octant's copy: no min/max pairs
                    tn0 = vmax3(r0.x * ix, r0.y * iy, r0.z * iz);
                    tf0 = vmin3(r0.w * ix, r1.x * iy, r1.y * iz);
                    tn1 = vmax3(r1.z * ix, r1.w * iy, r2.x * iz);
                    tf1 = vmin3(r2.y * ix, r2.z * iy, r2.w * iz);
                } else {
                    box_net(r0.x * ix, r0.y * iy, r0.z * iz, r0.w * ix, r1.x * iy, r1.y * iz, tn0, tf0);
                    box_net(r1.z * ix, r1.w * iy, r2.x * iz, r2.y * ix, r2.z * iy, r2.w * iz, tn1, tf1);
                }
                lref = __float_as_uint(r3.x), rref = __float_as_uint(r3.y);
            } else {
                // (32-bit byte offset from a scalar base: the record tables are < 4 GB, and an index scaled in 64 bits
                // costs two 64-bit VALU operations per fetch)
                const float4 *rec = (const float4 *)((const char *)inner + (uint32_t)(cur << 6));
                const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
                const float2 q3 = *(const float2 *)(rec + 3);
                float a0, a1, a2, a3, a4, a5, b0, b1, b2, b3, b4, b5;
                if (SRC == 0) {
                    a0 = q0.x * ix, a1 = q0.y * iy, a2 = q0.z * iz, a3 = q0.w * ix, a4 = q1.x * iy, a5 = q1.y * iz;
                    b0 = q1.z * ix, b1 = q1.w * iy, b2 = q2.x * iz, b3 = q2.y * ix, b4 = q2.z * iy, b5 = q2.w * iz;
                } else {
                    a0 = (q0.x - ox) * ix, a1 = (q0.y - oy) * iy, a2 = (q0.z - oz) * iz;
                    a3 = (q0.w - ox) * ix, a4 = (q1.x - oy) * iy, a5 = (q1.y - oz) * iz;
                    b0 = (q1.z - ox) * ix, b1 = (q1.w - oy) * iy, b2 = (q2.x - oz) * iz;
                    b3 = (q2.y - ox) * ix, b4 = (q2.z - oy) * iy, b5 = (q2.w - oz) * iz;
                }
                if (EXACT) {
                    box_net_exact(a0, a1, a2, a3, a4, a5, tn0, tf0);
                    box_net_exact(b0, b1, b2, b3, b4, b5, tn1, tf1);
                } else {
                    box_net(a0, a1, a2, a3, a4, a5, tn0, tf0);
                    box_net(b0, b1, b2, b3, b4, b5, tn1, tf1);
                }
                lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
            }
            // the compares as lane masks (v_cmp into SGPR pairs), combined with scalar ALU ops and handed
            // back to the lanes as conditions: written with &&/|| hipcc evaluates part of this under
            // temporary exec masks (fcmp predicates: 5 = ordered <=, 4 = ordered <, 2 = ordered >)
            const unsigned long long m0 = __builtin_amdgcn_fcmpf(tn0, tf0, 5), m1 = __builtin_amdgcn_fcmpf(tn1, tf1, 5);
            const unsigned long long m_right = m1 & (~m0 | __builtin_amdgcn_fcmpf(tn1, tn0, 4));
            const bool both = __builtin_amdgcn_inverse_ballot_w64(m0 & m1);
            const bool go_right = __builtin_amdgcn_inverse_ballot_w64(m_right);  // both: the strictly closer right child; one: that child
            if (both) {
                const uint2 e = make_uint2(go_right ? lref : rref, __float_as_uint(go_right ? tn0 : tn1));
                stack_push(stk, ovf, lds_entries, sp, e);
                ++sp;
            }
            const float near = go_right ? tn1 : tn0;
            // no child hit, or the child taken directly fails `near > t` (bvh.cpp:69): pop
            const bool popn = __builtin_amdgcn_inverse_ballot_w64(~(m0 | m1) | __builtin_amdgcn_fcmpf(near, best, 2));
            cur = popn ? kPop : (go_right ? rref : lref);
                } else if ((int)cur < 0) {
            // ---- one triangle of the leaf (triangle.cpp:4-54); the ref itself carries the progress
            float det, inv_det, u, v, dist;
            if (SRC == 0) {
                const uint32_t ti = (cur & kLeafStartMask) * 4;
                const float4 a = tris[ti], b = tris[ti + 1], c = tris[ti + 2];
                const float cd = ((const float *)tris)[ti * 4 + 12];
                // a = (e1, e2.x)  b = (e2.yz, tvec.xy)  c = (tvec.z, qvec)  cd = dot(e2, qvec)
                float pvx, pvy, pvz;
                cross3(dx, dy, dz, a.w, b.x, b.y, pvx, pvy, pvz);
                det = dot3(a.x, a.y, a.z, pvx, pvy, pvz);
                inv_det = 1.0f / det;
                u = dot3(b.z, b.w, c.x, pvx, pvy, pvz) * inv_det;
                v = dot3(dx, dy, dz, c.y, c.z, c.w) * inv_det;
                dist = cd * inv_det;
            } else {
                const uint32_t ti = (cur & kLeafStartMask) * 3;
                const float4 a = tris[ti], b = tris[ti + 1];
                const float e2z = ((const float *)tris)[ti * 4 + 8];
                const float e1x = a.w, e1y = b.x, e1z = b.y, e2x = b.z, e2y = b.w;
                float pvx, pvy, pvz;
                cross3(dx, dy, dz, e2x, e2y, e2z, pvx, pvy, pvz);
                det = dot3(e1x, e1y, e1z, pvx, pvy, pvz);
                inv_det = 1.0f / det;
                const float tx = ox - a.x, ty = oy - a.y, tz = oz - a.z;
                u = dot3(tx, ty, tz, pvx, pvy, pvz) * inv_det;
                float qx, qy, qz;
                cross3(tx, ty, tz, e1x, e1y, e1z, qx, qy, qz);
                v = dot3(dx, dy, dz, qx, qy, qz) * inv_det;
                dist = dot3(e2x, e2y, e2z, qx, qy, qz) * inv_det;
            }
            const bool parallel = fabsf(det) <= 9.99999993922529e-09f;
            const bool u_out = (u < 0.0f) || (u > 1.0f);
            const bool v_out = (v < 0.0f) || (u + v > 1.0f);
            const bool hit = !parallel && !u_out && !v_out && (dist > 0.0f);
            if (hit && dist < best) {  // strict <: first tested wins ties (bvh.cpp:90)
                best = dist;
                slot = (int)(cur & kLeafStartMask);
            }
            // next triangle: start + 1, count - 1; after the last one the lane pops
            cur = (((cur >> kLeafCountShift) & 31u) == 1u) ? kPop : cur + (1u - (1u << kLeafCountShift));
        }
        if (cur == kPop) {
            // pop until an entry passes `near > t` (bvh.cpp:69); level 0 holds the bottom entry
            // (near = -inf), which always passes and ends the ray
            uint2 e;
            do {
                --sp;
                e = stack_pop(stk, ovf, lds_entries, sp);
            } while (__uint_as_float(e.y) > best);
            cur = e.x;
            if (cur == kBottom) {
                if (SORT) rflags |= 0x80000000u;  // settled at the next refill (the lane keeps t, slot, direction until then)
                else hit_out[pid] = make_float2(best, __int_as_float(slot));
                cur = kIdle;
            }
        }
    };

    for (;;) {
        // ---- refill idle lanes -------------------------------------------------------
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(cur == kIdle);
        if (idle != 0 && !exhausted && ((uint32_t)__popcll(idle) >= refill_min || idle == ~0ull)) {
            if (SORT) sort_finished();
            for (;;) {
                if (res_lo == res_hi) {
                    uint32_t lim = SRC == 0 ? band_items : min(wk.qids.counts[src * 32], wk.qids.sub_capacity);
                    if (LIVE) {  // (scalar loads, once per reservation)
                        const uint32_t n = *wk.live_count, seg = live_band(n);
                        lim = min(seg, n - min(n, src * seg));
                    }
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&wk.heads[src * 32], kReserve);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= lim) {
                        src = (src + 1 == nsrc) ? 0 : src + 1;
                        if (++tried == nsrc) {
                            exhausted = true;
                            break;
                        }
                        continue;
                    }
                    tried = 0;
                    res_lo = base;
                    res_hi = min(base + kReserve, lim);
                }
                const unsigned long long want = __builtin_amdgcn_ballot_w64(cur == kIdle);
                if (want == 0) break;
                const uint32_t avail = res_hi - res_lo;
                const uint32_t rank = (uint32_t)__popcll(want & lt_mask);
                const bool take = cur == kIdle && rank < avail;
                const uint32_t item = res_lo + rank;
                res_lo = __builtin_amdgcn_readfirstlane(res_lo + min((uint32_t)__popcll(want), avail));
                if (take) {
                    bool valid = true;
                    if (LIVE) {
                        pid = src * live_band(*wk.live_count) + item;  // list position
                    } else if (SRC == 0) {
                        uint32_t j, s_idx;
                        if (wk.pixel_major) {
                            const uint32_t sl = fast_div(item, wk.div_samples);
                            j = item - sl * wk.samples, s_idx = src * band_slots + sl;
                        } else {
                            j = item / band_slots, s_idx = src * band_slots + (item - j * band_slots);
                        }
                        valid = s_idx < wk.n_active;
                        pid = path_id(wk, j, s_idx);
                    } else {
                        pid = wk.qids.ids[(size_t)src * wk.qids.sub_capacity + item];
                    }
                    if (valid) {
                        if (SRC == 0) {  // camera ray: (direction, flag word), origin from the frame
                            const float4 a = ((const float4 *)pa.rayA)[pid];
                            ox = fr.px, oy = fr.py, oz = fr.pz, dx = a.x, dy = a.y, dz = a.z;
                            valid = __float_as_uint(a.w) != 0xFFFFFFFFu;  // sample slot past the pixel's last sample
                            if (SORT && valid) rflags = __float_as_uint(a.w) & 7u;
                        } else {
                            const float4 a = ((const float4 *)pa.state)[(size_t)pid * 4], b = ((const float4 *)pa.state)[(size_t)pid * 4 + 1];
                            ox = a.x, oy = a.y, oz = a.z, dx = a.w, dy = b.x, dz = b.y;
                            valid = __float_as_uint(b.z) != 0xFFFFFFFFu;
                            if (SORT && valid) rflags = __float_as_uint(b.w) & 7u;  // k_shade: step_bits of this step
                        }
                    }
                    if (valid) {
                        ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;  // Ray.h:10
                        exact = !(finite3(ix, iy, iz) && finite3(ox, oy, oz));
                        best = 999999999.f;  // bvh.cpp:48
                        slot = -1;
                        sp = 1;
                        cur = root_ref;  // its near value, -9999999 (bvh.cpp:59), passes `near > t`
                    }
                }
            }
        }
        const unsigned long long active = __builtin_amdgcn_ballot_w64(cur != kIdle);
#ifdef VMX_STEP_PROFILE
        {
            const unsigned long long now = __builtin_readcyclecounter();
            if (lane == 0) {
                s_prof[wave][5][0] += now - prof_t;
                s_prof[wave][5][1] += 1;
            }
        }
#endif
        if (active == 0) break;
        if (SRC == 0) {
            const uint32_t oct = (__float_as_uint(ix) >> 31) | ((__float_as_uint(iy) >> 31) << 1) |
                                 ((__float_as_uint(iz) >> 31) << 2);
            const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)oct, (int)__ffsll((long long)active) - 1);
            wave_octant = __builtin_amdgcn_ballot_w64(cur != kIdle && oct != o0) == 0 ? o0 : 8u;
        }
        if (__builtin_amdgcn_ballot_w64(exact && cur != kIdle) != 0) {
#pragma unroll 1
            for (int act = 0; act < 8; ++act) VMX_PROF_STEP(std::true_type{}, std::false_type{});
        } else if (SRC == 0 && wave_octant < 8) {
            uni_base = (const char *)inner + (size_t)wave_octant * wk.cam_n_inner * 64;
            // (16 steps per batch here: these waves refill only when all 64 lanes are done; measured 8 / 16 / 32)
#pragma unroll 1
            for (int act = 0; act < VMX_CAM_BATCH; ++act) {
                // all 64 lanes at the same inner node: the assembly loop takes the wave down the tree while that
                // holds (uniform_descent), the general step goes on from where it stopped
                const unsigned long long trav = __builtin_amdgcn_ballot_w64(cur != kIdle);
                if (trav == 0) break;  // every ray of the wave is done: on to the refill (these waves refill all 64 lanes at once)
                {
                    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)__ffsll((long long)trav) - 1);
                    if (c0 < kPop && __builtin_amdgcn_ballot_w64(cur != kIdle && (cur != c0 || sp >= lds_entries)) == 0)
                    {
#ifdef VMX_STEP_PROFILE
                        const unsigned long long t0_ = __builtin_readcyclecounter();
#endif
                        const uint32_t done = uniform_descent(sp, cur, ix, iy, iz, best, slot, dx, dy, dz, (uint32_t)(uintptr_t)stk,
                                                              uni_base, (uint32_t)((const char *)tris - uni_base), c0, lds_entries,
                                                              trav);
#ifdef VMX_STEP_PROFILE
                        act += (int)done;
#else
                        (void)done;
                        act += 4;  // about what a descent takes (4.7 nodes on the bench frame); the batch only paces the refill test
#endif
#ifdef VMX_STEP_PROFILE
                        if (lane == 0) {
                            s_prof[wave][7][0] += __builtin_readcyclecounter() - t0_;
                            s_prof[wave][7][1] += done;
                            s_prof[wave][8][1] += 1;  // entries
                            s_prof[wave][8][0] += (unsigned long long)__popcll(trav);  // traversing lanes at entry
                        }
#endif
                    }
                }
                VMX_PROF_STEP(std::false_type{}, std::true_type{});
            }
        } else {
            uni_base = (const char *)inner;
#pragma unroll 1
            for (int act = 0; act < 8; ++act) VMX_PROF_STEP(std::false_type{}, std::false_type{});
        }
#ifdef VMX_STEP_PROFILE
        prof_t = __builtin_readcyclecounter();
#endif
    }
    if (SORT) {
        sort_finished();
        float4 *__restrict__ rec = (float4 *)wk.out_rec;
        constexpr uint32_t kRec = SRC == 0 ? 2 : 1;
        for (uint32_t i = out_lo + lane; i < min(out_hi, wk.out_capacity); i += 64u) rec[(size_t)i * kRec + (kRec - 1)] = make_float4(0.f, 0.f, __uint_as_float(0xFFFFFFFFu), 0.f);
        if (lane == 0) {
            if (n_rays) atomicAdd(&wk.out_ctr->stage[SRC].rays, (unsigned long long)n_rays);
            if (n_hits) atomicAdd(&wk.out_ctr->stage[SRC].tri_hits, (unsigned long long)n_hits);
        }
    }
#ifdef VMX_STEP_PROFILE
    if (lane < 20) atomicAdd(&g_step_prof[SRC][lane >> 1][lane & 1], s_prof[wave][lane >> 1][lane & 1]);
    if (lane == 0) {
        const unsigned long long now = (unsigned long long)wall_clock64();
        atomicMax(&g_wave_span[SRC][1], now);
        atomicAdd(&g_wave_span[SRC][2], now);
        atomicAdd(&g_wave_span[SRC][3], 1ull);
    }
#endif
}

// ---------------------------------------------------------------------------
// k_shade — the wide shading kernel of the split wavefront: the part of
// MeshEngine::RayCast after the BVH query (triangle normal, sphere table,
// meshEngine.cpp:365-508) and one iteration of Radiance's loop
// (pathtracer.cpp:36-196), one lane per path, then wave-ballot compaction of
// the surviving path ids.
//   SRC 0: depth-0 steps; the camera ray and the RNG stream are regenerated
//          from (pixel, sample) instead of being stored by the trace kernel
//   SRC 1: queued path ids; ray, RNG and accumulated colour come from the arrays
// ---------------------------------------------------------------------------
#ifndef VMX_SHADE_WPS
#define VMX_SHADE_WPS 7  // waves per SIMD k_shade is compiled for (<= 72 VGPRs): left to itself hipcc takes 100 (5 waves) since the
                        // cosf/sinf path came in; k_shade<0> 14.6 ms at 5-6 waves (the cap at 6 spills into the hot path), 13.1 at 7, 13.4 at 8
#endif
// FROMQ: 0 every path of the generation; 1 the ordered list of positions k_shade_ends + launch_live_compact left
// (wk.flat_ids); 2 the records k_trace_w<.., SORT> appended (wk.out_rec: ray, hit and position in one place)
template <int SRC, bool TEX, bool ELIDE = false, int FROMQ = 0>
__global__ void __launch_bounds__(256, VMX_SHADE_WPS)
k_shade(SceneDev sc, FrameDev fr, WorkDev wk, PixelStateDev px, PathArrays pa, IdQueue qout, uint32_t max_chunks,
        DevCounters *ctr) {
    const float2 *__restrict__ hits = (const float2 *)pa.hit;
    float4 *__restrict__ rad = (float4 *)pa.rad;
    __shared__ float4 s_geom[kLdsSpheres];
    __shared__ float4 s_cam_op[kLdsSpheres];  // SRC 0: (centre - camera position, squared length), same for every path
    if (threadIdx.x < min(sc.nspheres, kLdsSpheres)) {
        const SphereDev &q = sc.spheres[threadIdx.x];
        s_geom[threadIdx.x] = make_float4(q.cx, q.cy, q.cz, q.rad2);
        const float opx = q.cx - fr.px, opy = q.cy - fr.py, opz = q.cz - fr.pz;
        s_cam_op[threadIdx.x] = make_float4(opx, opy, opz, dot3(opx, opy, opz, opx, opy, opz));
    }
    __syncthreads();
    Tally tl = {{0, 0}, {0, 0}, {0, 0}};
    // FROMQ (two-phase shading): the work is the ordered list k_shade_ends + launch_live_compact left — the positions
    // of the paths whose step does not end by its draws alone (wk.flat_ids, *wk.flat_count).  A position is an index
    // into rayA / hit for camera paths (`src`), a place in the id queue for bounce paths
    uint32_t items = (SRC == 0 ? (wk.samples * wk.n_pad + blockDim.x - 1) / blockDim.x : max_chunks * kSubQueues);
    uint32_t flat_n = 0;
    if (FROMQ == 1) flat_n = *wk.flat_count, items = (flat_n + blockDim.x - 1) / blockDim.x;
    if (FROMQ == 2) flat_n = min(*wk.out_count, wk.out_capacity), items = (flat_n + blockDim.x - 1) / blockDim.x;
    // VMX_SAMPLING_ELIDE_DEAD: the pass's live camera paths only; ray and hit record sit at the list position `src`
    constexpr bool listed = SRC == 0 && ELIDE;
    uint32_t live_n = 0;
    if (listed) {
        live_n = *wk.live_count;
        if (FROMQ == 0) items = (live_n + blockDim.x - 1) / blockDim.x;
    }
    for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
        bool run;
        uint32_t pid = 0, src = 0;
        float2 hrec = make_float2(0.f, 0.f);  // FROMQ 2: the hit record came with the list entry
        Path P;
        if (SRC == 0) {
            src = pid = item * blockDim.x + threadIdx.x;
            float4 rec0 = make_float4(0.f, 0.f, 0.f, 0.f), rec1 = rec0;
            if (FROMQ == 1) {
                src = pid = src < flat_n ? wk.flat_ids[src] : 0xFFFFFFFFu;
                if (listed) live_n = 0xFFFFFFFFu;  // (a listed position is a valid one)
            }
            if (FROMQ == 2) {  // (direction, flag word)(t, leaf slot, position, -); position ~0: padding of a chunk
                const bool in = src < flat_n;
                if (in) rec0 = ((const float4 *)wk.out_rec)[(size_t)src * 2], rec1 = ((const float4 *)wk.out_rec)[(size_t)src * 2 + 1];
                src = pid = in ? __float_as_uint(rec1.z) : 0xFFFFFFFFu;
                hrec = make_float2(rec1.x, rec1.y);
                if (listed) live_n = 0xFFFFFFFFu;
            }
            if (listed) pid = src < live_n ? wk.live_ids[src] : 0xFFFFFFFFu;  // (no such path: j >= samples below)
            uint32_t j, s_idx;
            if (wk.pixel_major) s_idx = fast_div(pid, wk.div_samples), j = pid - s_idx * wk.samples;
            else j = pid / wk.n_pad, s_idx = pid - j * wk.n_pad;
            uint32_t pixel = 0, k = 0, pid2;
            run = j < wk.samples && s_idx < wk.n_pad && primary_item(fr, wk, px, j, s_idx, pid2, pixel, k);
            if (run) {
                // the ray comes from k_raygen; the stream is re-keyed and its two jitter draws skipped
                const float4 a = FROMQ == 2 ? rec0 : ((const float4 *)pa.rayA)[src];
                P.ox = fr.px, P.oy = fr.py, P.oz = fr.pz, P.dx = a.x, P.dy = a.y, P.dz = a.z, P.depth = 0;
                if (FROMQ == 0 && !listed && wk.pixel_major && (wk.samples & 63u) == 0) {
                    // the wave's 64 paths are 64 samples of ONE pixel (path ids of an item are consecutive): the pixel
                    // half of the key once per wave, on the scalar unit
                    const uint32_t pu = (uint32_t)__builtin_amdgcn_readfirstlane((int)pixel);
                    rng_init_keyed(P.rng, rng_pixel_key(fr.seed, pu), k);
                } else {
                    rng_init(P.rng, fr.seed, pixel, k);
                }
                (void)rng_next(P.rng);
                (void)rng_next(P.rng);
                P.ar = P.ag = P.ab = 0.f;
                P.aw = -100.f;  // pathtracer.cpp:29
                P.tr = P.tg = P.tb = P.tw = 1.f;  // :30
            }
        } else {
            uint32_t sub = item % kSubQueues, chunk = item / kSubQueues;
            uint32_t pos = chunk * blockDim.x + threadIdx.x;
            if (FROMQ == 1) {  // a listed position = (block item of k_shade_ends) * 256 + lane
                const uint32_t at = item * blockDim.x + threadIdx.x;
                run = at < flat_n;
                const uint32_t p = run ? wk.flat_ids[at] : 0u;
                sub = (p >> 8) % kSubQueues, chunk = (p >> 8) / kSubQueues, pos = chunk * 256u + (p & 255u);
            } else if (FROMQ == 2) {  // (t, leaf slot, path id, -); path id ~0: padding of a chunk
                const uint32_t at = item * blockDim.x + threadIdx.x;
                float4 r = make_float4(0.f, 0.f, __uint_as_float(0xFFFFFFFFu), 0.f);
                if (at < flat_n) r = ((const float4 *)wk.out_rec)[at];
                pid = __float_as_uint(r.z);
                hrec = make_float2(r.x, r.y);
                run = pid != 0xFFFFFFFFu;
            } else {
                run = pos < min(wk.qids.counts[sub * 32], wk.qids.sub_capacity);
            }
            if (run) {
                if (FROMQ != 2) pid = wk.qids.ids[(size_t)sub * wk.qids.sub_capacity + pos];
                path_load_arrays<TEX, false>(pa, pid, P);  // (accumColour: below, only where the step changes it)
            }
        }
        StepFlags fl = {false, false, false};
        uint32_t depth0 = 0;
        int st = kPathEnded;
        ShadeMid mid;
        CastResult c;
        if (run) {
            P.dest = pid;
            depth0 = P.depth == 0 ? 1u : 0u;
            fl.was_ray = depth0 ? true : finite3(P.dx, P.dy, P.dz);
            const float2 h = FROMQ == 2 ? hrec : hits[SRC == 0 ? src : pid];
            cast_finish<true, SRC == 0>(sc, P.ox, P.oy, P.oz, P.dx, P.dy, P.dz, h.x, __float_as_int(h.y), c, s_geom,
                                        s_cam_op);
            // A bounce step changes accumColour only by += accumRadiance * hitColour (pathtracer.cpp:43) — nothing when no
            // light coloured the hit (x + t * 0 == x for finite t; the sums are never -0) — and, at depth 0 (explicit rays
            // of vmx_radiance), by its fourth component (:44-47).  Six steps in seven change nothing: their path's
            // 16 bytes of a 64-byte line are neither fetched nor written back (a third of this kernel's HBM traffic).
            bool touch = SRC == 0, had = true;
            if (SRC != 0) {
                touch = P.depth == 0 || c.cr != 0.f || c.cg != 0.f || c.cb != 0.f;
                if (TEX) touch = touch || !(finite3(P.tr, P.tg, P.tb) && fabsf(P.tw) < kInf);
                if (touch) {
                    had = rad_has(pa, pid);
                    const float4 acc = rad_fetch(pa, pid, had);
                    P.ar = acc.x, P.ag = acc.y, P.ab = acc.z, P.aw = acc.w;
                }
            }
            st = path_shade_begin<TEX>(sc, fr.r2scale, P, c, fl, mid);
            if (SRC != 0) {
                if (touch) rad_commit(pa, pid, make_float4(P.ar, P.ag, P.ab, P.aw), had);
            } else if (!pa.rad_mask) {
                rad[pid] = make_float4(P.ar, P.ag, P.ab, P.aw);
            }
        }
        bool alive = st == kPathNextRay;
        // (gathering the block's ~10 % of angles in LDS to evaluate cos/sin in full waves was measured:
        // the three barriers it needs cost what it saves)
        if (st == kPathNeedsTrig) {
            float sn, cs;
            shade_trig(mid, fr.libm_double, cs, sn);
            alive = path_shade_end(P, c, fl, mid, cs, sn);
        }
        if (ELIDE && alive &&
            step_is_dead<TEX>(sc, fr.r2scale, P.rng, P.depth, P.ox, P.oy, P.oz, P.dx, P.dy, P.dz, P.tr, P.tg, P.tb))
            alive = false, fl.continues = false;  // VMX_SAMPLING_ELIDE_DEAD: the next ray cannot change the path's colour
        if (alive) {
            uint32_t bits = 0;
            if (fr.bounce_bits) bits = step_bits<TEX>(sc, fr.r2scale, P.rng, P.depth, P.ox, P.oy, P.oz, P.dx, P.dy, P.dz, P.tr, P.tg, P.tb);
            ray_store(pa, pid, P, bits);
            rng_store(pa, pid, P.rng);
            if (TEX) ((float4 *)pa.thr)[pid] = make_float4(P.tr, P.tg, P.tb, P.tw);
        }
        if (SRC == 0 && pa.rad_mask) {
            // camera paths: most end at their first hit with accumColour.rgb == 0 (no light sphere hit): nothing to store
            // for them, k_resolve adds +0 (x + 0 == x bit for bit; the sums are never -0).  A wave's 64 paths are one
            // aligned word of the mask (path ids of an item are consecutive, 256 per block)
            // (a path that goes on black stores nothing either: the step that colours it later sets its bit, rad_commit)
            const bool need = run && (P.ar != 0.f || P.ag != 0.f || P.ab != 0.f);
            if (need) rad[pid] = make_float4(P.ar, P.ag, P.ab, P.aw);
            if (listed || FROMQ != 0) {  // the mask was cleared for the pass / written by k_shade_ends; neighbours share words
                if (need) atomicOr(&pa.rad_mask[pid >> 6], 1ull << (pid & 63u));
            } else {
                const unsigned long long word = __builtin_amdgcn_ballot_w64(need);
                if ((threadIdx.x & 63u) == 0) pa.rad_mask[(item * blockDim.x + threadIdx.x) >> 6] = word;
            }
        }
        tally_add(tl, fl, run, depth0);
        id_append(qout, item % kSubQueues, alive, pid);
    }
    const Cnt none = {0, 0};
    tally_flush<false>(ctr, tl, none, none);
}

// ---------------------------------------------------------------------------
// k_shade_ends — first phase of the two-phase shading of a frame's split passes.  Under the reference's sampling four
// Radiance steps in five are the path's last one because of the path's own draws (Russian roulette :56; r2 = 10 U > 1,
// :156 / :170, where the hit's material flag — "the BVH hit a triangle", meshEngine.cpp:370 — picks the draw).  All
// such a step does is accumColour += accumRadiance * hitColour (:43): neither the hit's normal, nor its uv, nor the
// nearest distance is ever read.  And hitColour is decided by the sphere table's entries up to the last light
// (SceneDev::emit_prefix; `testHit < nearestHit` in table order, meshEngine.cpp:377-420).  This kernel takes every
// path of the generation, finishes those steps with exactly that — no attribute gather, no interpolation, 2 sphere
// tests instead of 8 with the reference's table — and marks the others (one bit per position; launch_live_compact
// makes the ordered list of them) for k_shade, which then runs in dense waves.  Same colours, same counters.
// (k_shade alone does the same work with four lanes in five idle behind the branch: 13.1 ms against 3.9 + 4 here
// for the 530.8 M camera paths of the bench frame.)
//   SRC 0: camera paths; the two "ends" bits come with the ray record (k_raygen: step_bits)
//   SRC 1: queued bounce paths; the draws are read off a copy of the path's stream
//   LISTED (SRC 0, VMX_SAMPLING_ELIDE_DEAD): the pass's live-path list instead of all its path ids
// ---------------------------------------------------------------------------
template <int SRC, bool TEX, bool LISTED>
__global__ void __launch_bounds__(256)
k_shade_ends(SceneDev sc, FrameDev fr, WorkDev wk, PathArrays pa, unsigned long long *__restrict__ full_mask,
             unsigned int *__restrict__ full_cnt, uint32_t max_chunks, DevCounters *ctr) {
    const float2 *__restrict__ hits = (const float2 *)pa.hit;
    float4 *__restrict__ rad = (float4 *)pa.rad;
    __shared__ float4 s_geom[kLdsSpheres];
    __shared__ float4 s_cam_op[kLdsSpheres];
    __shared__ float4 s_col[kLdsSpheres];  // (colour, emits)
    const uint32_t nlight = min(sc.emit_prefix, kLdsSpheres);
    if (threadIdx.x < nlight) {
        const SphereDev &q = sc.spheres[threadIdx.x];
        s_geom[threadIdx.x] = make_float4(q.cx, q.cy, q.cz, q.rad2);
        const float opx = q.cx - fr.px, opy = q.cy - fr.py, opz = q.cz - fr.pz;
        s_cam_op[threadIdx.x] = make_float4(opx, opy, opz, dot3(opx, opy, opz, opx, opy, opz));
        s_col[threadIdx.x] = make_float4(q.colr, q.colg, q.colb, (q.flags & 1u) ? 1.f : 0.f);
    }
    __syncthreads();
    Tally tl = {{0, 0}, {0, 0}, {0, 0}};
    uint32_t items = SRC == 0 ? (wk.samples * wk.n_pad + blockDim.x - 1) / blockDim.x : max_chunks * kSubQueues;
    uint32_t live_n = 0;
    if (LISTED) live_n = *wk.live_count, items = (live_n + blockDim.x - 1) / blockDim.x;
    for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
        bool run, ends = false, material = false;
        uint32_t pid = 0, src = 0, depth = 0;
        float ox = fr.px, oy = fr.py, oz = fr.pz, dx = 0.f, dy = 0.f, dz = 0.f, best = 0.f;
        float tr = 1.f, tg = 1.f, tb = 1.f;
        if (SRC == 0) {
            src = pid = item * blockDim.x + threadIdx.x;
            run = LISTED ? src < live_n : src < wk.samples * wk.n_pad;
            if (run) {
                if (LISTED) pid = wk.live_ids[src];
                const float4 a = ((const float4 *)pa.rayA)[src];
                const uint32_t bits = __float_as_uint(a.w);
                run = bits != 0xFFFFFFFFu;  // a slot without a sample
                const float2 h = hits[src];
                dx = a.x, dy = a.y, dz = a.z, best = h.x;
                material = __float_as_int(h.y) >= 0;
                ends = ((material ? bits : bits >> 1) & 1u) != 0;
            }
        } else {
            const uint32_t sub = item % kSubQueues, chunk = item / kSubQueues;
            const uint32_t pos = chunk * blockDim.x + threadIdx.x;
            run = pos < min(wk.qids.counts[sub * 32], wk.qids.sub_capacity);
            if (run) {
                src = pid = wk.qids.ids[(size_t)sub * wk.qids.sub_capacity + pos];
                Path P;
                ray_load(pa, pid, P);
                Rng r;
                rng_load(pa, pid, r);
                ox = P.ox, oy = P.oy, oz = P.oz, dx = P.dx, dy = P.dy, dz = P.dz, depth = P.depth;
                const float2 h = hits[pid];
                best = h.x;
                material = __float_as_int(h.y) >= 0;
                // the draws of this step, in path_shade_begin's order
                bool rr_end = false;
                if (depth + 1u > 5u) {
                    const double rr = rng_u01(r);
                    rr_end = rr > (double)0.95f || depth + 1u > 1000u;
                }
                if (material) {
                    const double a = rng_u01(r);
                    (void)rng_next(r);
                    const double c = rng_u01(r);
                    ends = !(a >= 0.96) && (1.0f - (float)((double)fr.r2scale * c)) < 0.0f;
                } else {
                    (void)rng_next(r);
                    const double b = rng_u01(r);
                    ends = (1.0 - (double)fr.r2scale * b) < 0.0;
                }
                ends = ends || rr_end;
                if (TEX) {
                    const float4 th = ((const float4 *)pa.thr)[pid];
                    tr = th.x, tg = th.y, tb = th.z;
                    if (!finite3(tr, tg, tb)) ends = false;  // inf * 0: leave the step to k_shade
                }
            }
        }
        const bool fin = run && ends;
        // hitColour of the ending steps: the sphere table up to its last light, `testHit > 0 && testHit < nearestHit` in order
        float cr = 0.f, cg = 0.f, cb = 0.f;
        if (__builtin_amdgcn_ballot_w64(fin) != 0) {
            float nearest = material ? best : kInf;
            for (uint32_t i = 0; i < nlight; ++i) {
                float th;
                if (SRC == 0) {
                    const float4 q = s_cam_op[i];
                    th = sphere_hit_op(q.x, q.y, q.z, q.w, s_geom[i].w, dx, dy, dz, nearest);
                } else {
                    th = sphere_hit(ox, oy, oz, dx, dy, dz, s_geom[i], nearest);
                }
                if (fin && th > 0.f && th < nearest) {
                    nearest = th;
                    const float4 col = s_col[i];
                    if (col.w != 0.f) cr = col.x, cg = col.y, cb = col.z;
                }
            }
        }
        const bool lit = fin && (cr != 0.f || cg != 0.f || cb != 0.f);
        if (SRC == 0) {
            // accumColour = 0 + 1 * hitColour; stored only where it is not zero (PathArrays::rad_mask)
            if (lit) rad[pid] = make_float4(0.f + tr * cr, 0.f + tg * cg, 0.f + tb * cb, -100.f);
            if (LISTED) {  // the mask was cleared for the pass
                if (lit) atomicOr(&pa.rad_mask[pid >> 6], 1ull << (pid & 63u));
            } else {       // this wave's 64 path ids are one word; k_shade ORs the bits of its paths in afterwards
                const unsigned long long word = __builtin_amdgcn_ballot_w64(lit);
                if ((threadIdx.x & 63u) == 0 && src < wk.samples * wk.n_pad) pa.rad_mask[src >> 6] = word;
            }
        } else if (lit) {
            const bool had = rad_has(pa, pid);
            float4 acc = rad_fetch(pa, pid, had);
            if (TEX) acc.x = acc.x + tr * cr, acc.y = acc.y + tg * cg, acc.z = acc.z + tb * cb;
            else acc.x = acc.x + cr, acc.y = acc.y + cg, acc.z = acc.z + cb;
            rad_commit(pa, pid, acc, had);
        }
        StepFlags fl = {false, false, false};
        fl.was_ray = depth == 0u ? true : finite3(dx, dy, dz);
        fl.tri_hit = material;
        tally_add(tl, fl, fin, depth == 0u ? 1u : 0u);
        // the others: one word of bits per wave (position = block item * 256 + lane), compacted in order afterwards
        const unsigned long long full = __builtin_amdgcn_ballot_w64(run && !ends);
        if ((threadIdx.x & 63u) == 0) {
            const uint32_t w = item * (blockDim.x >> 6) + (threadIdx.x >> 6);
            full_mask[w] = full, full_cnt[w] = (uint32_t)__popcll(full);
        }
    }
    const Cnt none = {0, 0};
    tally_flush<false>(ctr, tl, none, none);
}

// explicit camera rays for vmx_radiance (split wavefront): path i keyed (seed, i, 0), jitter draws skipped
__global__ void k_radiance_init_ids(const float *__restrict__ o, const float *__restrict__ d, uint32_t n,
                                    uint64_t seed, PathArrays pa, IdQueue qout) {
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const bool run = i < n;
        if (run) {
            Path P;
            rng_init(P.rng, seed, i, 0);
            (void)rng_next(P.rng);
            (void)rng_next(P.rng);
            P.ox = o[i * 3], P.oy = o[i * 3 + 1], P.oz = o[i * 3 + 2];
            P.dx = d[i * 3], P.dy = d[i * 3 + 1], P.dz = d[i * 3 + 2];
            P.depth = 0;
            ray_store(pa, i, P);
            rng_store(pa, i, P.rng);
            ((float4 *)pa.rad)[i] = make_float4(0.f, 0.f, 0.f, -100.f);
            if (pa.thr) ((float4 *)pa.thr)[i] = make_float4(1.f, 1.f, 1.f, 1.f);
        }
        id_append(qout, (base / blockDim.x) % kSubQueues, run, i);
    }
}

// per-pixel accumulation in sample order + early stop + pixel write (pathtracer.cpp:282-324)
__global__ void k_resolve(FrameDev fr, const unsigned int *__restrict__ active, uint32_t n_active,
                          uint32_t n_pad, uint32_t samples, uint32_t pixel_major, const float4 *__restrict__ rad,
                          const unsigned long long *__restrict__ rad_mask, PixelStateDev px,
                          unsigned int *__restrict__ next_active, unsigned int *next_count,
                          float *__restrict__ out, DevCounters *ctr) {
    __shared__ unsigned int s_keep, s_base, s_taken, s_disc, s_done, s_brk;
    if (threadIdx.x == 0) s_keep = 0, s_taken = 0, s_disc = 0, s_done = 0, s_brk = 0;
    __syncthreads();
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    uint32_t lp = 0, taken = 0, disc = 0, done = 0, brk = 0;
    if (slot < n_active) {
        lp = active[slot];
        float4 acc = ((float4 *)px.accum)[lp];
        uint32_t n = px.count[lp];
        const uint32_t craw = px.cursor[lp];
        const uint32_t cursor = craw & ~kCursorStrided;
        // strided: this pass's paths are the first samples of the following strata (sample_index)
        const bool strided = (craw & kCursorStrided) != 0;
        const uint32_t stride = strided ? fr.quarter : 1u;
        uint32_t next = cursor + samples * stride;
        bool flag = strided;  // still in a run of early stops unless a sample says otherwise
        // samples are taken strictly in order (float sums, early stop), but their loads need not
        // wait for each other: fetch 8 at a time (pixel-major: one 128-byte line per lane)
        constexpr uint32_t kChunk = 8;
        const size_t total = (size_t)n_pad * samples;
        bool stop = false;
        for (uint32_t j0 = 0; j0 < samples && !stop; j0 += kChunk) {
            float4 buf[kChunk];
            // rad_mask (split pipeline, pixel-major ids): one bit per path says whether its radiance was stored at all;
            // the others ended black at their first hit and count as an exact zero
            unsigned long long w0 = ~0ull, w1 = ~0ull;
            const size_t pid0 = (size_t)slot * samples + j0;
            if (rad_mask) {
                w0 = rad_mask[pid0 >> 6];
                w1 = rad_mask[min(pid0 + kChunk - 1, total - 1) >> 6];
            }
#pragma unroll
            for (uint32_t i = 0; i < kChunk; ++i) {
                const uint32_t j = j0 + i;
                size_t idx = pixel_major ? (size_t)slot * samples + j : (size_t)j * n_pad + slot;
                bool stored = true;
                if (j >= samples || idx >= total) idx = (size_t)slot, stored = false;  // any valid address; value unused
                if (rad_mask) {
                    const size_t pid = pid0 + i;
                    stored = stored && ((((pid >> 6) == (pid0 >> 6) ? w0 : w1) >> (pid & 63)) & 1ull) != 0;
                }
                buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (stored) buf[i] = rad[idx];
            }
#pragma unroll
            for (uint32_t i = 0; i < kChunk; ++i) {
                const uint32_t j = j0 + i;
                const bool lead_tail = fr.lead != 0 && j >= fr.lead;  // a following stratum's first sample
                const uint32_t k = lead_tail ? (j - fr.lead + 1u) * fr.quarter : cursor + j * stride;
                if (stop || j >= samples || k >= fr.kmax) {
                    stop = true;
                    continue;
                }
                const float4 s = buf[i];
                ++n;  // :249
                acc.x = acc.x + s.x;  // :283
                acc.y = acc.y + s.y;
                acc.z = acc.z + s.z;
                ++taken;
                bool early = false;
                if (fr.early_stop && n > fr.nmin) {  // n > sqrt(spp), :292
                    const float fn = (float)n, fn1 = (float)(n + 1);
                    const float ex = acc.x / fn - (acc.x + s.x) / fn1;
                    const float ey = acc.y / fn - (acc.y + s.y) / fn1;
                    const float ez = acc.z / fn - (acc.z + s.z) / fn1;
                    early = fabsf(sqrtf(dot3(ex, ey, ez, ex, ey, ez))) < 0.00001f;
                }
                // number of this pass's paths that come after sample j
                uint32_t later = samples - 1u - j;
                if (fr.lead != 0) {
                    const uint32_t strata_after = fr.kmax / fr.quarter - 1u;
                    const uint32_t valid = fr.lead + (samples - fr.lead < strata_after ? samples - fr.lead : strata_after);
                    later = valid - 1u - j;
                } else if (cursor + (samples - 1u) * stride >= fr.kmax) {
                    later = (fr.kmax - 1u - k) / stride;
                }
                const bool strided_now = strided || lead_tail;
                const bool lead_last = fr.lead != 0 && j + 1u == fr.lead;  // the first sample the rule can stop
                if (early) {
                    brk = 1;
                    next = (k / fr.quarter + 1u) * fr.quarter;  // break the innermost loop only
                    flag = true;
                    // consecutive samples that follow in this pass are not taken — except that a lead
                    // pass goes on with the next strata's first samples, which are exactly what is due
                    if (!strided_now && !lead_last) {
                        disc = later;
                        stop = true;
                    }
                } else if (strided_now || lead_last) {
                    // this stratum goes on with sample k + 1; the later strata were not due yet
                    next = k + 1u;
                    flag = false;
                    disc = later;
                    stop = true;
                }
            }
        }
        if (next >= fr.kmax) {
            const float fn = (float)n;  // accum / nTotalSamples (uint -> float)
            float *o5 = out + (size_t)lp * 5;
            // std::max(std::min(x, 1.f), 0.f) (pathtracer.cpp:318-320): std::min(a, b) is (b < a) ? b : a, so a NaN mean —
            // an infinite throughput times a black hit, possible with a texture that holds an infinity — stays NaN
            // (fminf / fmaxf would return the other operand: found by tools/fuzz_parity.py's random textures, round 3)
            const float mx = acc.x / fn, my = acc.y / fn, mz = acc.z / fn;
            const float cx = 1.f < mx ? 1.f : mx, cy = 1.f < my ? 1.f : my, cz = 1.f < mz ? 1.f : mz;
            o5[0] = cx < 0.f ? 0.f : cx;
            o5[1] = cy < 0.f ? 0.f : cy;
            o5[2] = cz < 0.f ? 0.f : cz;
            o5[3] = 1.f;
            o5[4] = fn;
            done = 1;
        } else {
            keep = true;
        }
        ((float4 *)px.accum)[lp] = acc;
        px.count[lp] = n;
        px.cursor[lp] = next | ((flag && next < fr.kmax) ? kCursorStrided : 0u);
    }
    // block-aggregated append to the next active list
    uint32_t my = 0;
    if (keep) my = atomicAdd(&s_keep, 1u);
    if (taken) atomicAdd(&s_taken, taken);
    if (disc) atomicAdd(&s_disc, disc);
    if (done) atomicAdd(&s_done, 1u);
    if (brk) atomicAdd(&s_brk, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        s_base = s_keep ? atomicAdd(next_count, s_keep) : 0u;
        if (s_brk) atomicAdd(next_count + 1, s_brk);
        if (s_taken) atomicAdd(&ctr->samples, (unsigned long long)s_taken);
        if (s_disc) atomicAdd(&ctr->discarded, (unsigned long long)s_disc);
        if (s_done) atomicAdd(&ctr->pixels_done, (unsigned long long)s_done);
    }
    __syncthreads();
    if (keep) next_active[s_base + my] = lp;
}

// ---------------------------------------------------------------------------
// k_bruteforce / k_bruteforce_long — BruteForceTracer::Render (core/integrators/integrators.cpp:9-186), the
// engine's default integrator: the reference's sample loop — jittered camera ray (:65-81), RayCast, N.L against
// the point light (:83-88), normal perturbation by boundTextures[0] (:98-106), the mirror probe (:119-137),
// albedo (:141-156), the convergence break (:166-172) — and the pixel write (:176-183).  Jitters come from the
// stream keyed (seed, pixel, sample) (the reference shares one unsynchronised std::mt19937, :30), so a sample is
// a function of (pixel, sample index) alone (bf_sample) and only the running sums and the break are sequential
// (bf_fold).  Most pixels break after 3 samples, a few run to the last one: with one lane per pixel those few
// set the frame time (186 ms at 256 spp, of which 25 ms is work).  So k_bruteforce (one lane per pixel, 8x8-pixel
// tiles: coherent waves) takes the first kBfShort samples of every pixel and hands the pixels that have not
// broken by then — with their sums — to k_bruteforce_long, where a WAVE takes a pixel: its 64 lanes evaluate 64
// consecutive samples at once, lane 0 folds them in order, and the samples past the break are dropped.
// ---------------------------------------------------------------------------
struct BfSample {
    float cx, cy, cz, dist;
    uint32_t flags;  // 1: the camera ray hit something, 2: a triangle, 4: the mirror probe was a ray (finite direction)
};
struct BfState {
    float ax, ay, az, aw;  // accum
    float lx, ly, lz, lw;  // lastSampleColour
    float dist;            // hitDistance of the last sample
    uint32_t n;            // samples taken
    uint32_t prim, sec, hits;
};
constexpr uint32_t kBfShort = 8;

__device__ __forceinline__ BfSample bf_sample(const SceneDev &sc, const FrameDev &fr, uint32_t p, uint32_t sample,
                                              uint2 *stk, uint2 *ovf, int lds_entries, Cnt &cnt) {
    BfSample r = {0.f, 0.f, 0.f, 0.f, 0u};
    Rng rng;
    rng_init(rng, fr.seed, p, sample);
    const float jx = rng_jitter(rng), jy = rng_jitter(rng);
    const float hX = (float)((((double)((float)(p % fr.width) + jx) - 0.25) / (double)fr.width) * 2.0 - 1.0);   // :65
    const float hY = (float)((((double)((float)(p / fr.width) + jy) - 0.25) / (double)fr.height) * 2.0 - 1.0);  // :66
    const float bx = (float)((double)(hX * fr.sensor_x) * 0.5), by = (float)((double)(hY * fr.sensor_y) * 0.5);  // :73-74
    const float gx = bx, gy = -by, gz = -fr.film_dist;  // :76
    float dx = (fr.m[0] * gx + fr.m[3] * gy) + (fr.m[6] * gz + 0.0f);  // :79, GLM mat4*vec4 order
    float dy = (fr.m[1] * gx + fr.m[4] * gy) + (fr.m[7] * gz + 0.0f);
    float dz = (fr.m[2] * gx + fr.m[5] * gy) + (fr.m[8] * gz + 0.0f);
    normalize3(dx, dy, dz);  // :81 (w == 1)
    CastResult c;
    ray_cast<false>(sc, fr.px, fr.py, fr.pz, dx, dy, dz, stk, c, cnt, ovf, lds_entries);
    r.dist = c.nearest;  // *pHitDistance (meshEngine.cpp:507)
    if (c.nearest < kInf) {
        r.flags = 1u | (c.slot >= 0 ? 2u : 0u);
        const float hx = fr.px + (dx * c.nearest), hy = fr.py + (dy * c.nearest), hz = fr.pz + (dz * c.nearest);
        float nx = c.nx, ny = c.ny, nz = c.nz;
        float Lx = 500.f - hx, Ly = 1100.f - hy, Lz = 2000.f - hz;  // :16,83
        normalize3(Lx, Ly, Lz);                                     // :84
        float unx = nx, uny = ny, unz = nz;
        normalize3(unx, uny, unz);
        float vNDL = dot3(Lx, Ly, Lz, unx, uny, unz);  // :88
        if (sc.tex) {                                   // :98-106
            const float4 t = tex_sample(sc, c.uvx, c.uvy);
            nx = nx + t.x, ny = ny + t.y, nz = nz + t.z;
            unx = nx, uny = ny, unz = nz;
            normalize3(unx, uny, unz);
            vNDL = dot3(Lx, Ly, Lz, unx, uny, unz);
        }
        // :119  -L - 2.f * N * dot(N, -L)
        const float mLx = -Lx, mLy = -Ly, mLz = -Lz;
        const float k = dot3(nx, ny, nz, mLx, mLy, mLz);
        const float sx = mLx - (2.f * nx) * k, sy = mLy - (2.f * ny) * k, sz = mLz - (2.f * nz) * k;
        if (finite3(sx, sy, sz)) r.flags |= 4u;
        // :121 — the probe's RayCast is only asked WHETHER it hit (:133).  RayCast returns nearest < INFINITY
        // (meshEngine.cpp:508), which is true iff the BVH query hit or some sphere of the table gave 0 < t < inf
        // (:378 with nearest still infinite) — whatever the order.  The spheres are asked first: inside the reference's
        // room (six 5e7-radius wall spheres) every probe with a finite direction hits one, and the BVH traversal —
        // half of this integrator's rays, the incoherent half — is only run for the probes that hit no sphere.
        bool probe_hit = false;
        for (uint32_t i = 0; i < sc.nspheres && !probe_hit; ++i) {
            const SphereDev &q = sc.spheres[i];
            const float th = sphere_hit(hx, hy, hz, sx, sy, sz, make_float4(q.cx, q.cy, q.cz, q.rad2), kInf);
            probe_hit = th > 0.f && th < kInf;
        }
        if (!probe_hit) {
            float best2;
            int slot2;
            bvh_nearest<false>(sc, hx, hy, hz, sx, sy, sz, stk, best2, slot2, cnt, ovf, lds_entries);
            probe_hit = slot2 >= 0;
        }
        if (!probe_hit) {                                            // :133-137
            vNDL = vNDL * 0.9f;
            vNDL = vNDL + 0.1f;
        }
        if (sc.tex1) {  // :141-147
            const float4 t = tex_sample_of(sc.tex1, sc.tex1_w, sc.tex1_h, sc.tex1_c, c.uvx, c.uvy);
            r.cx = t.x * vNDL, r.cy = t.y * vNDL, r.cz = t.z * vNDL;
        } else {  // :148-156
            r.cx = 0.890196078f * vNDL, r.cy = 0.258823529f * vNDL, r.cz = 0.203921569f * vNDL;
        }
    }
    return r;
}

// the sequential part of the sample loop (:59,158-173) for one sample; true = the convergence break fired
__device__ __forceinline__ bool bf_fold(BfState &st, const BfSample &r, uint32_t flags) {
    ++st.n;
    ++st.prim;
    st.dist = r.dist;
    if (r.flags & 1u) {
        if (r.flags & 2u) ++st.hits;
        if (r.flags & 4u) ++st.sec;
        st.ax = st.ax + r.cx, st.ay = st.ay + r.cy, st.az = st.az + r.cz, st.aw = st.aw + 1.0f;  // :158 (cw = 1)
    }
    const float fn = (float)st.n;
    if (st.n > 2) {  // :167-172
        st.lx = st.lx - st.ax / fn, st.ly = st.ly - st.ay / fn, st.lz = st.lz - st.az / fn, st.lw = st.lw - st.aw / fn;
        const float sum = ((st.lx + st.ly) + st.lz) + st.lw;
        float mag;
        if (flags & 1u) {  // VMX_BF_ABS_INT: abs(int), the float truncated to int first
            const double tr = trunc((double)sum);
            mag = (tr >= -2147483648.0 && tr <= 2147483647.0) ? (float)abs((int)tr) : 0.f;
        } else {
            mag = fabsf(sum);
        }
        if (mag < 0.001f) return true;
    }
    st.lx = st.ax / fn, st.ly = st.ay / fn, st.lz = st.az / fn, st.lw = st.aw / fn;  // :173
    return false;
}

__device__ __forceinline__ void bf_write_pixel(const BfState &st, float *__restrict__ out, uint32_t lp) {
    const float fn = (float)st.n;
    float *o5 = out + (size_t)lp * 5;  // :176-183
    o5[0] = sel_max(sel_min(st.ax / fn, 1.f), 0.f);  // std::max(std::min(x, 1.f), 0.f), NaN and all
    o5[1] = sel_max(sel_min(st.ay / fn, 1.f), 0.f);
    o5[2] = sel_max(sel_min(st.az / fn, 1.f), 0.f);
    o5[3] = st.aw / fn;
    o5[4] = st.dist;
}

// a pixel handed from k_bruteforce to k_bruteforce_long: 16 dwords
struct BfLong {
    uint32_t lp, n;
    float ax, ay, az, aw, lx, ly, lz, lw, dist;
    uint32_t pad[5];
};

__global__ void __launch_bounds__(256)
k_bruteforce(SceneDev sc, FrameDev fr, const unsigned int *__restrict__ order, uint32_t npix, uint32_t flags,
             float *__restrict__ out, DevCounters *ctr, BfLong *__restrict__ longs, unsigned int *long_count,
             uint32_t lds_levels, uint32_t overflow_entries, void *overflow_stack) {
    // (stack: lds_levels per lane in LDS, deeper entries in the global slab, as in the traversal kernels — the
    // whole stack in LDS leaves room for 3 waves per SIMD)
    extern __shared__ uint2 lds_stack[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int lds_entries = (int)lds_levels;
    uint2 *stk = lds_stack + (size_t)wave * (lds_levels + 1) * 64 + lane;
    uint2 *ovf = (uint2 *)overflow_stack + ((size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * overflow_entries) * 64 + lane;
    Cnt cnt = {0, 0};
    uint32_t n_prim = 0, n_sec = 0, n_hits = 0, n_samples = 0;
    const uint32_t cap = min(fr.spp, kBfShort);
    for (uint32_t base = blockIdx.x * blockDim.x; base < npix; base += gridDim.x * blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        if (i >= npix) continue;
        const uint32_t lp = order[i];
        const uint32_t p = global_pixel(fr, lp);
        BfState st = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u, 0u, 0u};
        bool broke = false;
        for (uint32_t sample = 0; sample < cap && !broke; ++sample)  // :59
            broke = bf_fold(st, bf_sample(sc, fr, p, sample, stk, ovf, lds_entries, cnt), flags);
        n_prim += st.prim, n_sec += st.sec, n_hits += st.hits;
        if (broke || st.n >= fr.spp) {
            n_samples += st.n;
            bf_write_pixel(st, out, lp);
        } else {  // not converged yet: a wave of k_bruteforce_long goes on from here
            const uint32_t at = atomicAdd(long_count, 1u);
            BfLong rec = {lp, st.n, st.ax, st.ay, st.az, st.aw, st.lx, st.ly, st.lz, st.lw, st.dist, {0, 0, 0, 0, 0}};
            longs[at] = rec;
        }
    }
    const uint32_t wp = wave_sum(n_prim), ws = wave_sum(n_sec), wh = wave_sum(n_hits), wn = wave_sum(n_samples);
    if (lane == 0) {
        if (wp) atomicAdd(&ctr->stage[0].rays, (unsigned long long)wp);
        if (ws) atomicAdd(&ctr->stage[1].rays, (unsigned long long)ws);
        if (wh) atomicAdd(&ctr->stage[0].tri_hits, (unsigned long long)wh);
        if (wn) atomicAdd(&ctr->samples, (unsigned long long)wn);
    }
}

__global__ void __launch_bounds__(256)
k_bruteforce_long(SceneDev sc, FrameDev fr, uint32_t flags, float *__restrict__ out, DevCounters *ctr,
                  const BfLong *__restrict__ longs, const unsigned int *long_count, uint32_t lds_levels,
                  uint32_t overflow_entries, void *overflow_stack) {
    extern __shared__ uint2 lds_stack[];
    __shared__ BfSample s_res[4][64];
    __shared__ uint32_t s_stop[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int lds_entries = (int)lds_levels;
    uint2 *stk = lds_stack + (size_t)wave * (lds_levels + 1) * 64 + lane;
    uint2 *ovf = (uint2 *)overflow_stack + ((size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * overflow_entries) * 64 + lane;
    Cnt cnt = {0, 0};
    const uint32_t total = *long_count, waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t e = blockIdx.x * (blockDim.x >> 6) + wave; e < total; e += waves) {
        const BfLong rec = longs[e];
        const uint32_t p = global_pixel(fr, rec.lp);
        BfState st = {rec.ax, rec.ay, rec.az, rec.aw, rec.lx, rec.ly, rec.lz, rec.lw, rec.dist, rec.n, 0u, 0u, 0u};
        uint32_t next = rec.n;  // the next sample index (= samples taken so far)
        bool broke = false;
        while (!broke && next < fr.spp) {
            const uint32_t count = min(64u, fr.spp - next);
            if (lane < count) s_res[wave][lane] = bf_sample(sc, fr, p, next + lane, stk, ovf, lds_entries, cnt);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) {
                uint32_t j = 0;
                for (; j < count && !broke; ++j) broke = bf_fold(st, s_res[wave][j], flags);
                s_stop[wave] = broke ? 1u : 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            broke = s_stop[wave] != 0u;
            next += count;
        }
        if (lane == 0) {
            bf_write_pixel(st, out, rec.lp);
            if (st.prim) atomicAdd(&ctr->stage[0].rays, (unsigned long long)st.prim);
            if (st.sec) atomicAdd(&ctr->stage[1].rays, (unsigned long long)st.sec);
            if (st.hits) atomicAdd(&ctr->stage[0].tri_hits, (unsigned long long)st.hits);
            atomicAdd(&ctr->samples, (unsigned long long)st.n);
        }
    }
}


__global__ void k_assemble(const float *__restrict__ gathered, uint64_t rank_stride, uint32_t width,
                           uint32_t height, uint32_t stripe_rows, uint32_t world, float *__restrict__ frame) {
    const uint64_t total = (uint64_t)width * height * 5;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(i / ((uint64_t)width * 5));
        const uint64_t in_row = i - (uint64_t)row * width * 5;
        const uint32_t gs = row / stripe_rows, r = row - gs * stripe_rows;
        const uint32_t rank = gs % world, ls = gs / world;
        const uint64_t lrow = (uint64_t)ls * stripe_rows + r;
        frame[i] = gathered[rank * rank_stride + lrow * width * 5 + in_row];
    }
}

// Camera::saveFrame's conversion loop (camera.cpp:159-163): floor(x*255) -> u8, depth plane
__global__ void k_quantize(const float *__restrict__ frame, uint64_t npix, uchar4 *__restrict__ rgba8,
                           float *__restrict__ depth) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (uint64_t)gridDim.x * blockDim.x) {
        const float *f = frame + p * 5;
        uchar4 o;
        o.x = (unsigned char)floorf(f[0] * 255.0f);
        o.y = (unsigned char)floorf(f[1] * 255.0f);
        o.z = (unsigned char)floorf(f[2] * 255.0f);
        o.w = (unsigned char)floorf(f[3] * 255.0f);
        rgba8[p] = o;
        if (depth) depth[p] = f[4];
    }
}

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
int launch_trace(const SceneDev &sc, const float *o, const float *d, uint32_t n, int32_t *tri_id, float *t,
                 DevCounters *counters, bool count, LaunchCfg cfg, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (count)
        hipLaunchKernelGGL(k_trace<true>, dim3(cfg.grid), dim3(cfg.block), cfg.lds_bytes, s, sc, o, d, n, tri_id, t,
                           counters);
    else
        hipLaunchKernelGGL(k_trace<false>, dim3(cfg.grid), dim3(cfg.block), cfg.lds_bytes, s, sc, o, d, n, tri_id, t,
                           counters);
    return launch_status();
}

int launch_raycast(const SceneDev &sc, const float *o, const float *d, uint32_t n, void *out, LaunchCfg cfg,
                   void *stream) {
    hipLaunchKernelGGL(k_raycast, dim3(cfg.grid), dim3(cfg.block), cfg.lds_bytes, (hipStream_t)stream, sc, o, d, n,
                       (float4 *)out);
    return launch_status();
}

int launch_primary_ids(const SceneDev &sc, const FrameDev &fr, uint32_t k, int32_t *tri_id, float *t, LaunchCfg cfg,
                       void *stream) {
    hipLaunchKernelGGL(k_primary_ids, dim3(cfg.grid), dim3(cfg.block), cfg.lds_bytes, (hipStream_t)stream, sc, fr, k,
                       tri_id, t);
    return launch_status();
}

int launch_trig(const float *x, uint32_t n, float *cs, float *sn, void *stream) {
    hipLaunchKernelGGL(k_trig, dim3(std::min<uint32_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, x, n, cs, sn);
    return launch_status();
}

int launch_init_pixels(PixelStateDev px, uint32_t npix, void *stream) {
    hipLaunchKernelGGL(k_init_pixels, dim3((npix + 255) / 256), dim3(256), 0, (hipStream_t)stream, px, npix);
    return launch_status();
}

int launch_zero_u32(unsigned int *p, uint32_t n, void *stream) {
    hipLaunchKernelGGL(k_zero_u32, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, n);
    return launch_status();
}

int launch_paths(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PixelStateDev px, void *rad,
                 DevCounters *counters, bool count, LaunchCfg cfg, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    dim3 g(cfg.grid), b(cfg.block);
#define VMX_GO(C, T) \
    hipLaunchKernelGGL((k_paths<C, 0, T>), g, b, cfg.lds_bytes, s, sc, fr, wk, px, (float4 *)rad, pa, counters)
    PathArrays pa{};
    if (sc.tex) {
        if (count) VMX_GO(true, true);
        else VMX_GO(false, true);
    } else {
        if (count) VMX_GO(true, false);
        else VMX_GO(false, false);
    }
#undef VMX_GO
    return launch_status();
}

int launch_tail(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PathArrays pa, DevCounters *counters,
                bool count, LaunchCfg cfg, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    dim3 g(cfg.grid), b(cfg.block);
    PixelStateDev px{nullptr, nullptr, nullptr};
#define VMX_GO(C, T) \
    hipLaunchKernelGGL((k_paths<C, 2, T>), g, b, cfg.lds_bytes, s, sc, fr, wk, px, (float4 *)pa.rad, pa, counters)
    if (sc.tex) {
        if (count) VMX_GO(true, true);
        else VMX_GO(false, true);
    } else {
        if (count) VMX_GO(true, false);
        else VMX_GO(false, false);
    }
#undef VMX_GO
    return launch_status();
}

int launch_camera_tables(const SceneDev &sc, uint32_t n_inner, float ox, float oy, float oz, void *cam_inner,
                         void *cam_tris, void *stream) {
    const uint32_t total = n_inner + sc.ntris;
    uint32_t grid = std::min<uint32_t>((total + 255) / 256, 4096);
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(k_camera_tables, dim3(grid), dim3(256), 0, (hipStream_t)stream, sc, n_inner, ox, oy, oz,
                       (float4 *)cam_inner, (float4 *)cam_tris);
    return launch_status();
}

int launch_raygen(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PixelStateDev px, PathArrays pa, void *stream) {
    const uint64_t total = (uint64_t)wk.samples * wk.n_pad;
    uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256u * 32u);
    if (grid == 0) grid = 1;
    if (wk.live_mask) hipLaunchKernelGGL(k_raygen<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sc, fr, wk, px, pa);
    else hipLaunchKernelGGL(k_raygen<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sc, fr, wk, px, pa);
    return launch_status();
}

int launch_raygen_live(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PixelStateDev px, PathArrays pa, void *stream) {
    // the list's length is known on the device only: a grid for a third of the pass's paths, striding over the rest
    const uint64_t total = ((uint64_t)wk.samples * wk.n_pad + 2) / 3;
    uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256u * 32u);
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(k_raygen_live, dim3(grid), dim3(256), 0, (hipStream_t)stream, sc, fr, wk, px, pa);
    return launch_status();
}

int launch_trace_q(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PixelStateDev px, PathArrays pa,
                   DevCounters *counters, bool count, bool from_queue, LaunchCfg cfg, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    dim3 g(cfg.grid), b(cfg.block);
    // production form: k_trace_w; with counters: the first form k_trace_q (same tests per ray)
    if (!count) {
        if (from_queue) {
            if (wk.out_rec) hipLaunchKernelGGL((k_trace_w<1, false, true>), g, b, cfg.lds_bytes, s, sc, fr, wk, pa);
            else hipLaunchKernelGGL((k_trace_w<1>), g, b, cfg.lds_bytes, s, sc, fr, wk, pa);
        } else if (wk.out_rec) {
            if (wk.live_ids) hipLaunchKernelGGL((k_trace_w<0, true, true>), g, b, cfg.lds_bytes, s, sc, fr, wk, pa);
            else hipLaunchKernelGGL((k_trace_w<0, false, true>), g, b, cfg.lds_bytes, s, sc, fr, wk, pa);
        } else if (wk.live_ids) hipLaunchKernelGGL((k_trace_w<0, true>), g, b, cfg.lds_bytes, s, sc, fr, wk, pa);
        else hipLaunchKernelGGL((k_trace_w<0>), g, b, cfg.lds_bytes, s, sc, fr, wk, pa);
    } else {
        if (from_queue) hipLaunchKernelGGL((k_trace_q<true, 1>), g, b, cfg.lds_bytes, s, sc, fr, wk, px, pa, counters);
        else hipLaunchKernelGGL((k_trace_q<true, 0>), g, b, cfg.lds_bytes, s, sc, fr, wk, px, pa, counters);
    }
    return launch_status();
}

// occupancy of the exact instantiation launch_trace_q selects (sorted: k_trace_w<.., SORT>; live: the ELIDE_DEAD list
// form of the camera kernel): the persistent grid and the record-list padding are sized from it
int query_trace_q_blocks_per_cu(uint32_t block, uint32_t lds_bytes, bool count, bool from_queue, bool sorted, bool live, int *blocks) {
    int a = 0;
    hipError_t e;
#define VMX_OCC(K) hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, K, (int)block, lds_bytes)
    if (count) {
        e = from_queue ? VMX_OCC((k_trace_q<true, 1>)) : VMX_OCC((k_trace_q<true, 0>));
    } else if (from_queue) {
        e = sorted ? VMX_OCC((k_trace_w<1, false, true>)) : VMX_OCC((k_trace_w<1>));
    } else if (sorted) {
        e = live ? VMX_OCC((k_trace_w<0, true, true>)) : VMX_OCC((k_trace_w<0, false, true>));
    } else {
        e = live ? VMX_OCC((k_trace_w<0, true>)) : VMX_OCC((k_trace_w<0>));
    }
#undef VMX_OCC
    if (blocks) *blocks = a;
    return (int)e;
}

// from_queue: bounce paths (wk.qids), else the camera paths of the pass.  wk.flat_ids: the second phase after
// launch_shade_ends + launch_live_compact — only the listed positions.
int launch_shade(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PixelStateDev px, PathArrays pa,
                 IdQueue qout, uint32_t max_chunks, DevCounters *counters, bool from_queue, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    const bool flat = wk.flat_ids != nullptr || wk.out_rec != nullptr;  // a list whose length is known on the device only -> a full grid
    // 7 blocks per CU are resident (VMX_SHADE_WPS); more blocks only add end-of-block counter atomics, which a small pass
    // feels (early-stop frame, first pass: 1.70 ms with 4096 blocks, 1.05 with 1792; the bench frame's 530.8 M paths: no change)
    constexpr uint32_t kShadeGrid = 256u * VMX_SHADE_WPS;
    const uint64_t items = flat ? kShadeGrid : (from_queue ? (uint64_t)max_chunks * kSubQueues : ((uint64_t)wk.samples * wk.n_pad + 255) / 256);
    uint32_t grid = (uint32_t)std::min<uint64_t>(items, kShadeGrid);
    if (grid == 0) grid = 1;
#define VMX_GO(S, T, E, Q) \
    hipLaunchKernelGGL((k_shade<S, T, E, Q>), dim3(grid), dim3(256), 0, s, sc, fr, wk, px, pa, qout, max_chunks, counters)
#define VMX_GO_T(S, E, Q)                \
    do {                                 \
        if (sc.tex) VMX_GO(S, true, E, Q); \
        else VMX_GO(S, false, E, Q);     \
    } while (0)
    // camera paths are shaded from the live list exactly when render_impl built one (wk.live_ids)
    const bool elide = from_queue ? fr.elide_dead != 0 : wk.live_ids != nullptr;
    if (from_queue) {
        if (wk.out_rec) { if (elide) VMX_GO_T(1, true, 2); else VMX_GO_T(1, false, 2); }
        else if (flat) { if (elide) VMX_GO_T(1, true, 1); else VMX_GO_T(1, false, 1); }
        else { if (elide) VMX_GO_T(1, true, 0); else VMX_GO_T(1, false, 0); }
    } else if (wk.out_rec) {
        if (elide) VMX_GO_T(0, true, 2);
        else VMX_GO_T(0, false, 2);
    } else if (flat) {
        if (elide) VMX_GO_T(0, true, 1);
        else VMX_GO_T(0, false, 1);
    } else {
        if (elide) VMX_GO_T(0, true, 0);
        else VMX_GO_T(0, false, 0);
    }
#undef VMX_GO_T
#undef VMX_GO
    return launch_status();
}

// first phase of the two-phase shading (k_shade_ends): finishes the steps that end by their draws alone, queues the
// others' positions in full_mask / full_cnt (one word per wave of 64 positions) for launch_live_compact + launch_shade
int launch_shade_ends(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PathArrays pa, unsigned long long *full_mask,
                      unsigned int *full_cnt, uint32_t max_chunks, DevCounters *counters, bool from_queue, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    const uint64_t items = from_queue ? (uint64_t)max_chunks * kSubQueues : ((uint64_t)wk.samples * wk.n_pad + 255) / 256;
    constexpr uint32_t kEndsGrid = 256u * 8u;  // 8 waves per SIMD resident; see launch_shade
    uint32_t grid = (uint32_t)std::min<uint64_t>(items, kEndsGrid);
    if (grid == 0) grid = 1;
#define VMX_GO(S, T, L) \
    hipLaunchKernelGGL((k_shade_ends<S, T, L>), dim3(grid), dim3(256), 0, s, sc, fr, wk, pa, full_mask, full_cnt, max_chunks, counters)
    if (from_queue) {
        if (sc.tex) VMX_GO(1, true, false);
        else VMX_GO(1, false, false);
    } else if (wk.live_ids) {
        if (sc.tex) VMX_GO(0, true, true);
        else VMX_GO(0, false, true);
    } else {
        if (sc.tex) VMX_GO(0, true, false);
        else VMX_GO(0, false, false);
    }
#undef VMX_GO
    return launch_status();
}

int launch_radiance_init_ids(const float *o, const float *d, uint32_t n, uint64_t seed, PathArrays pa, IdQueue qout,
                             void *stream) {
    uint32_t grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_radiance_init_ids, dim3(grid), dim3(256), 0, (hipStream_t)stream, o, d, n, seed, pa, qout);
    return launch_status();
}

int query_paths_blocks_per_cu(uint32_t block, uint32_t lds_bytes, bool count, int *blocks) {
    int a = 0, b = 0;
    hipError_t e;
    if (count) {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_paths<true, 0, true>, (int)block, lds_bytes);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_paths<true, 2, true>, (int)block, lds_bytes);
    } else {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_paths<false, 0, true>, (int)block, lds_bytes);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_paths<false, 2, true>, (int)block, lds_bytes);
    }
    if (blocks) *blocks = a < b ? a : b;
    return (int)e;
}

int launch_resolve(const FrameDev &fr, const unsigned int *active, uint32_t n_active, uint32_t samples,
                   bool pixel_major, const void *rad, const unsigned long long *rad_mask, PixelStateDev px, unsigned int *next_active,
                   unsigned int *next_count, float *out_rgbaz, DevCounters *counters, void *stream) {
    const uint32_t n_pad = (n_active + 63u) & ~63u;
    hipLaunchKernelGGL(k_resolve, dim3((n_active + 255) / 256), dim3(256), 0, (hipStream_t)stream, fr, active,
                       n_active, n_pad, samples, pixel_major ? 1u : 0u, (const float4 *)rad, pixel_major ? rad_mask : nullptr, px, next_active,
                       next_count, out_rgbaz, counters);
    return launch_status();
}

int launch_bruteforce(const SceneDev &sc, const FrameDev &fr, const unsigned int *order, uint32_t npix, uint32_t flags,
                      float *out, DevCounters *counters, void *longs, unsigned int *long_count, const WorkDev &stack,
                      LaunchCfg cfg, void *stream) {
    hipLaunchKernelGGL(k_bruteforce, dim3(cfg.grid), dim3(cfg.block), cfg.lds_bytes, (hipStream_t)stream, sc, fr, order,
                       npix, flags, out, counters, (BfLong *)longs, long_count, stack.lds_entries, stack.overflow_entries,
                       stack.overflow_stack);
    return launch_status();
}
// the pixels k_bruteforce handed over (*long_count of them, known to the host as `count`): one wave each
int launch_bruteforce_long(const SceneDev &sc, const FrameDev &fr, uint32_t flags, float *out, DevCounters *counters,
                           const void *longs, const unsigned int *long_count, uint32_t count, const WorkDev &stack,
                           LaunchCfg cfg, void *stream) {
    const uint32_t blocks = std::min<uint32_t>(cfg.grid, (count + (cfg.block / 64) - 1) / (cfg.block / 64));
    hipLaunchKernelGGL(k_bruteforce_long, dim3(std::max(blocks, 1u)), dim3(cfg.block), cfg.lds_bytes, (hipStream_t)stream, sc, fr,
                       flags, out, counters, (const BfLong *)longs, long_count, stack.lds_entries, stack.overflow_entries,
                       stack.overflow_stack);
    return launch_status();
}

int launch_assemble(const float *gathered, uint64_t rank_stride_floats, uint32_t width, uint32_t height,
                    uint32_t stripe_rows, uint32_t world, float *frame, void *stream) {
    hipLaunchKernelGGL(k_assemble, dim3(2048), dim3(256), 0, (hipStream_t)stream, gathered, rank_stride_floats, width,
                       height, stripe_rows, world, frame);
    return launch_status();
}

int launch_quantize(const float *frame, uint64_t npix, void *rgba8, float *depth, void *stream) {
    uint32_t grid = (uint32_t)std::min<uint64_t>((npix + 255) / 256, 8192);
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(k_quantize, dim3(grid), dim3(256), 0, (hipStream_t)stream, frame, npix, (uchar4 *)rgba8, depth);
    return launch_status();
}

#ifdef VMX_AB_KERNELS
// first-generation kernels (pipeline forms 2, 3): only in the A/B library of `make ab`, never in the product
#include "vmx_kernels_ab.inc"
#include "vmx_trace_pool.inc"
#endif

}  // namespace vmx
