/*
 * vmx_oracle.cpp — CPU ORACLE for the Vermilion path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under vermilion_amd/ or in
 * libvermilion_hip.so may include, link, import or execute this file; it is
 * loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg, and there only as the checker / the timed CPU baseline.
 *
 * What it is: an independent restatement, in plain C++ with its own small
 * vector types, of the reference's algorithm for
 *   BBox::intersect                 core/accelerators/bbox.cpp:70-83
 *   Triangle::getIntersection       core/accelerators/triangle.cpp:4-54
 *   Triangle::getNormal             core/accelerators/triangle.cpp:56-87
 *   BVH::build / getIntersection    core/accelerators/bvh.cpp:179-279 / 47-145
 *   sphereIntersect                 core/engines/meshEngine.cpp:182-194
 *   MeshEngine::RayCast             core/engines/meshEngine.cpp:239-509
 *   Radiance                        core/integrators/pathtracer.cpp:21-198
 *   PathTracer::Render              core/integrators/pathtracer.cpp:200-328
 *   BruteForceTracer::Render        core/integrators/integrators.cpp:9-186
 *   Camera ctor / setPixelValue     core/camera/camera.cpp:34-81, 88-124
 * Each function below cites the lines it follows.
 *
 * PARITY UNPINNED.  The reference ships no tests, golden vectors or fixtures
 * (SURVEY.md §4), and its hot path cannot be compiled in this image: every
 * translation unit needs GLM (extern/glm is an empty, un-pinned submodule,
 * .gitmodules:7-9) and meshEngine/camera additionally need Assimp and
 * OpenImageIO headers; building it would need hand-written stand-ins for
 * those headers, which is not a reference build.  GLM's arithmetic is
 * therefore restated from its published generic (non-SIMD) code path
 * (0.9.9 series: dot = (x*x + y*y) + z*z, normalize = v * (1/sqrt(dot)),
 * cross, mat4*vec4 as (m0*x + m1*y) + (m2*z + m3*w), gtc rotate); libstdc++'s
 * <random> is used directly where the reference's RNG is wanted.  The only
 * reference-run observations available are the probe results recorded in
 * SURVEY.md (Appendix A-1/A-2/A-9, §8a-6); tests/test_oracle.py checks
 * this file against those.
 *
 * Build: see oracle/Makefile.  The parity build is -O2 -ffp-contract=off
 * (every float operation rounds once, no FMA), which is what the HIP kernels
 * are compiled to match.
 */
#include "vmx_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

/* ------------------------------------------------------------------ */
/* vector arithmetic (GLM generic path, restated)                      */
/* ------------------------------------------------------------------ */
struct V2 {
    float x, y;
};
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct V4 {
    float x, y, z, w;
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V2 operator*(V2 a, float s) { return {a.x * s, a.y * s}; }
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator*(V4 a, V4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }

/* glm::dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z */
inline float dot(V3 a, V3 b) {
    V3 t = a * b;
    return t.x + t.y + t.z;
}
/* glm::cross */
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
/* glm::length = sqrt(dot(v,v)) */
inline float length(V3 v) { return std::sqrt(dot(v, v)); }
/* glm::normalize = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x) */
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
/* glm::min / glm::max: (b<a)?b:a and (a<b)?b:a per component */
inline float gmin(float a, float b) { return (b < a) ? b : a; }
inline float gmax(float a, float b) { return (a < b) ? b : a; }

/* ------------------------------------------------------------------ */
/* scene                                                               */
/* ------------------------------------------------------------------ */
struct Tri { /* core/accelerators/triangle.h:5-9 */
    V3 v0, v1, v2;
    V3 n0, n1, n2;
    V2 t0, t1, t2;
    uint32_t id; /* createBVH push order, meshEngine.cpp:712 */
};

struct Box { /* core/accelerators/bbox.h:6-7 */
    V3 lo, hi, extent;
};

inline Box box_of(V3 lo, V3 hi) { return Box{lo, hi, hi - lo}; } /* bbox.cpp:5-6 */
inline Box box_of(V3 p) { return Box{p, p, p - p}; }             /* bbox.cpp:8-9 */
inline void grow(Box &b, V3 p) {                                 /* bbox.cpp:29-33 */
    b.lo = {std::min(b.lo.x, p.x), std::min(b.lo.y, p.y), std::min(b.lo.z, p.z)};
    b.hi = {std::max(b.hi.x, p.x), std::max(b.hi.y, p.y), std::max(b.hi.z, p.z)};
    b.extent = b.hi - b.lo;
}
inline void grow(Box &b, const Box &o) { /* bbox.cpp:35-39 */
    b.lo = {std::min(b.lo.x, o.lo.x), std::min(b.lo.y, o.lo.y), std::min(b.lo.z, o.lo.z)};
    b.hi = {std::max(b.hi.x, o.hi.x), std::max(b.hi.y, o.hi.y), std::max(b.hi.z, o.hi.z)};
    b.extent = b.hi - b.lo;
}
/* bbox.cpp:41-46 — note: z is compared with y only */
inline uint32_t widest_axis(const Box &b) {
    uint32_t r = 0;
    if (b.extent.y > b.extent.x) r = 1;
    if (b.extent.z > b.extent.y) r = 2;
    return r;
}
/* triangle.cpp:107-114 */
inline Box tri_box(const Tri &t) {
    V3 lo = {std::min(std::min(t.v0.x, t.v1.x), t.v2.x), std::min(std::min(t.v0.y, t.v1.y), t.v2.y),
             std::min(std::min(t.v0.z, t.v1.z), t.v2.z)};
    V3 hi = {std::max(std::max(t.v0.x, t.v1.x), t.v2.x), std::max(std::max(t.v0.y, t.v1.y), t.v2.y),
             std::max(std::max(t.v0.z, t.v1.z), t.v2.z)};
    return box_of(lo, hi);
}
/* triangle.cpp:116-119 */
inline V3 tri_centroid(const Tri &t) { return (t.v0 + t.v1 + t.v2) * 0.333f; }

struct FlatNode { /* bvh.h:11-14 */
    Box box;
    uint32_t start, nprims, right_offset;
};

struct RayQ { /* Ray.h:4-11 */
    V3 o, d, inv_d;
};
inline RayQ make_ray(V3 o, V3 d) { return RayQ{o, d, v3(1, 1, 1) / d}; }

struct Counters {
    uint64_t inner_visits = 0, tri_tests = 0, pops = 0, max_stack = 0;
};

} // namespace

struct orc_scene {
    std::vector<Tri> tris;          /* storage, createBVH order */
    std::vector<const Tri *> prims; /* build_prims, permuted by build (bvh.cpp:252) */
    std::vector<FlatNode> nodes;
    std::vector<vmx_sphere> spheres;
    uint32_t leaf_size = 4, n_leaves = 0, max_depth = 0;
    /* boundTextures[0] (meshEngine.h:62): VermiTexture{nWidth,nHeight,nChannels,pData}, meshEngine.cpp:7-19 */
    std::vector<float> tex;
    uint16_t tex_w = 0, tex_h = 0, tex_c = 0;
    /* boundTextures[1]: read only by BruteForceTracer (integrators.cpp:141-147) */
    std::vector<float> tex1;
    uint16_t tex1_w = 0, tex1_h = 0, tex1_c = 0;
    uint32_t n_textures = 0;
};

namespace {

/* BVH::build, bvh.cpp:179-279 */
void build_bvh(orc_scene &sc) {
    struct Entry {
        uint32_t parent, start, end, depth;
    };
    const uint32_t kRoot = 0xfffffffcu, kUntouched = 0xffffffffu, kTouchedTwice = 0xfffffffdu;
    std::vector<Entry> todo;
    todo.push_back({kRoot, 0, (uint32_t)sc.prims.size(), 0});
    std::vector<FlatNode> &out = sc.nodes;
    out.clear();
    out.reserve(sc.prims.size() * 2);
    uint32_t n_nodes = 0;
    sc.n_leaves = 0;
    sc.max_depth = 0;
    while (!todo.empty()) {
        Entry e = todo.back();
        todo.pop_back();
        uint32_t start = e.start, end = e.end, np = end - start;
        n_nodes++;
        FlatNode node;
        node.start = start;
        node.nprims = np;
        node.right_offset = kUntouched;
        Box bb = tri_box(*sc.prims[start]);           /* :209 */
        Box bc = box_of(tri_centroid(*sc.prims[start])); /* :210 */
        for (uint32_t p = start + 1; p < end; ++p) {
            grow(bb, tri_box(*sc.prims[p]));
            grow(bc, tri_centroid(*sc.prims[p]));
        }
        node.box = bb;
        if (np <= sc.leaf_size) { /* :219 */
            node.right_offset = 0;
            sc.n_leaves++;
        }
        out.push_back(node);
        sc.max_depth = std::max(sc.max_depth, e.depth);
        if (e.parent != kRoot) { /* :228-236 */
            out[e.parent].right_offset--;
            if (out[e.parent].right_offset == kTouchedTwice)
                out[e.parent].right_offset = n_nodes - 1 - e.parent;
        }
        if (node.right_offset == 0) continue;
        uint32_t dim = widest_axis(bc);                            /* :243 */
        float split = .5f * (bc.lo[dim] + bc.hi[dim]);             /* :246 */
        uint32_t mid = start;
        for (uint32_t i = start; i < end; ++i) {                   /* :250-255 */
            if (tri_centroid(*sc.prims[i])[dim] < split) {
                std::swap(sc.prims[i], sc.prims[mid]);
                ++mid;
            }
        }
        if (mid == start || mid == end) mid = start + (end - start) / 2; /* :258-260 */
        todo.push_back({n_nodes - 1, mid, end, e.depth + 1});   /* right first, :263-266 */
        todo.push_back({n_nodes - 1, start, mid, e.depth + 1}); /* left on top, :269-272 */
    }
}

/* BBox::intersect, bbox.cpp:70-83 */
inline bool box_hit(const Box &b, const RayQ &r, float *tnear, float *tfar) {
    V3 t0 = (b.lo - r.o) * r.inv_d;
    V3 t1 = (b.hi - r.o) * r.inv_d;
    V3 ts = {gmin(t0.x, t1.x), gmin(t0.y, t1.y), gmin(t0.z, t1.z)};
    V3 tl = {gmax(t0.x, t1.x), gmax(t0.y, t1.y), gmax(t0.z, t1.z)};
    *tnear = std::max(std::max(ts.x, ts.y), ts.z);
    *tfar = std::min(std::min(tl.x, tl.y), tl.z);
    return *tnear <= *tfar;
}

/* Triangle::getIntersection, triangle.cpp:4-54 */
inline bool tri_hit(const Tri &tr, const RayQ &ray, float *t_out) {
    V3 rot = ray.d, pos = ray.o;
    V3 e1 = tr.v1 - tr.v0;
    V3 e2 = tr.v2 - tr.v0;
    V3 pvec = cross(rot, e2);
    float det = dot(e1, pvec);
    if (det < 1e-8 && det > -1e-8) return false; /* double literals, :25 */
    float inv_det = 1 / det;
    V3 tvec = pos - tr.v0;
    float u = dot(tvec, pvec) * inv_det;
    if (u < 0 || u > 1) return false;
    V3 qvec = cross(tvec, e1);
    float v = dot(rot, qvec) * inv_det;
    if (v < 0 || u + v > 1) return false;
    float dist = dot(e2, qvec) * inv_det;
    if (dist > 0.0f) {
        *t_out = dist;
        return true;
    }
    return false;
}

/* Triangle::getNormal, triangle.cpp:56-87 (lines 58-65 are dead code: their
 * results are overwritten before use) */
inline V3 tri_normal(const Tri &tr, V3 hit, V2 *uv) {
    V3 f0 = tr.v1 - tr.v0;
    V3 f1 = tr.v2 - tr.v0;
    V3 f2 = hit - tr.v0;
    float d00 = dot(f0, f0);
    float d01 = dot(f0, f1);
    float d11 = dot(f1, f1);
    float d20 = dot(f2, f0);
    float d21 = dot(f2, f1);
    float denom = d00 * d11 - d01 * d01;
    float w1 = (d11 * d20 - d01 * d21) / denom;
    float w2 = (d00 * d21 - d01 * d20) / denom;
    float w0 = 1 - w1 - w2;
    V3 n = tr.n0 * w0 + tr.n1 * w1 + tr.n2 * w2;
    if (uv) *uv = tr.t0 * w0 + tr.t1 * w1 + tr.t2 * w2;
    return -n;
}

/* BVH::getIntersection (occlusion == false), bvh.cpp:47-145 */
inline bool bvh_nearest(const orc_scene &sc, const RayQ &ray, float *t_out, const Tri **obj_out,
                        Counters *cnt) {
    float best = 999999999.f;
    const Tri *obj = nullptr;
    struct Todo {
        uint32_t i;
        float mint;
    } todo[64];
    int sp = 0;
    todo[0] = {0, -9999999.f};
    float bb[4] = {0, 0, 0, 0};
    while (sp >= 0) {
        int ni = (int)todo[sp].i;
        float near = todo[sp].mint;
        sp--;
        if (cnt) cnt->pops++;
        const FlatNode &node = sc.nodes[ni];
        if (near > best) continue; /* :69 */
        if (node.right_offset == 0) {
            for (uint32_t o = 0; o < node.nprims; ++o) {
                const Tri *tr = sc.prims[node.start + o];
                float tt;
                if (cnt) cnt->tri_tests++;
                if (tri_hit(*tr, ray, &tt)) {
                    if (tt < best) { /* strict: first tested wins ties, :90 */
                        best = tt;
                        obj = tr;
                    }
                }
            }
        } else {
            if (cnt) cnt->inner_visits++;
            bool h0 = box_hit(sc.nodes[ni + 1].box, ray, bb, bb + 1);
            bool h1 = box_hit(sc.nodes[ni + node.right_offset].box, ray, bb + 2, bb + 3);
            if (h0 && h1) {
                int closer = ni + 1, other = ni + (int)node.right_offset;
                if (bb[2] < bb[0]) { /* :110 */
                    std::swap(bb[0], bb[2]);
                    std::swap(bb[1], bb[3]);
                    std::swap(closer, other);
                }
                todo[++sp] = {(uint32_t)other, bb[2]};
                todo[++sp] = {(uint32_t)closer, bb[0]};
            } else if (h0) {
                todo[++sp] = {(uint32_t)(ni + 1), bb[0]};
            } else if (h1) {
                todo[++sp] = {(uint32_t)(ni + node.right_offset), bb[2]};
            }
            if (cnt && (uint64_t)(sp + 1) > cnt->max_stack) cnt->max_stack = (uint64_t)(sp + 1);
        }
    }
    *t_out = best;
    *obj_out = obj;
    return obj != nullptr;
}

/* sphereIntersect, meshEngine.cpp:182-194 — note the float dots and the
 * float rad*rad feeding a double discriminant */
inline float sphere_hit(V3 pos, V3 rot, V3 p, const float rad) {
    V3 op = p - pos;
    double t;
    double eps = 1e-4;
    double b = dot(op, rot);
    double det = b * b - dot(op, op) + rad * rad;
    if (det < 0)
        return 0;
    else
        det = std::sqrt(det);
    return (float)((t = b - det) > eps ? t : ((t = b + det) > eps ? t : 0));
}

const vmx_sphere kDefaultSpheres[8] = {
    /* meshEngine.cpp:377-387  light 1 */
    {{15.f, 140.f, 25.f}, 3.5f, {0.f * 15.f, .5f * 15.f, 1.0f * 15.f}, VMX_SPHERE_EMIT, {-55.f, 350.f, -150.f}, -1.f},
    /* meshEngine.cpp:410-420  light 2 */
    {{0.f, 3300.f, 1300.f}, 250.f, {1.0f * 15.2f, 1.0f * 15.2f, 1.0f * 15.2f}, VMX_SPHERE_EMIT, {500.f, 800.f, 1300.f}, 1.f},
    /* meshEngine.cpp:444-450  floor */
    {{0.f, (float)(-1e7 * 5), 0.f}, (float)(1e7 * 5), {0, 0, 0}, 0u, {0.f, (float)(-1e7 * 5), 0.f}, 1.f},
    /* :452-459 ceiling */
    {{0.f, (float)(1e7 * 5 + 1000), 0.f}, (float)(1e7 * 5), {0, 0, 0}, 0u, {0.f, (float)(1e7 * 5 + 1000), 0.f}, 1.f},
    /* :463-470 -x wall */
    {{(float)(-1e7 * 5 + 2000), 0.f, 0.f}, (float)(1e7 * 5), {0, 0, 0}, 0u, {(float)(-1e7 * 5 + 2000), 0.f, 0.f}, -1.f},
    /* :472-479 +x wall */
    {{(float)(1e7 * 5 - 2000), 0.f, 0.f}, (float)(1e7 * 5), {0, 0, 0}, 0u, {(float)(1e7 * 5 - 2000), 0.f, 0.f}, -1.f},
    /* :483-490 -z wall */
    {{0.f, 0.f, (float)(-1e7 * 5 + 2000)}, (float)(1e7 * 5), {0, 0, 0}, 0u, {0.f, 0.f, (float)(-1e7 * 5 + 2000)}, -1.f},
    /* :492-499 +z wall */
    {{0.f, 0.f, (float)(1e7 * 5 - 2000)}, (float)(1e7 * 5), {0, 0, 0}, 0u, {0.f, 0.f, (float)(1e7 * 5 - 2000)}, 1.f},
};

struct CastOut {
    bool hit;      /* return value */
    bool material; /* *ppImpactMaterial != nullptr */
    V3 location, normal, colour;
    V2 uv;
    float distance;
    int32_t tri_id;
    float tri_t;
};

/* MeshEngine::RayCast, meshEngine.cpp:239-509 */
inline CastOut ray_cast(const orc_scene &sc, V3 o, V3 d, Counters *cnt) {
    CastOut r;
    r.uv = {0, 0};
    r.normal = {0, 0, 0};
    r.colour = {0, 0, 0};
    r.material = false;
    r.location = {0, 0, 0};
    r.distance = 0.f;
    r.tri_id = -1;
    float nearest = INFINITY; /* :271 */
    float test = 0.f;
    int hit_mesh = -1;

    RayQ ray = make_ray(o, d); /* :361 */
    float bt;
    const Tri *obj;
    bool bh = bvh_nearest(sc, ray, &bt, &obj, cnt); /* :364 */
    r.tri_t = bt;
    if (bh) {
        nearest = bt;
        V3 hitp = ray.o + ray.d * bt;                    /* bvh.cpp:140 */
        r.normal = normalize(tri_normal(*obj, hitp, &r.uv)); /* :369 */
        hit_mesh = 0;
        r.tri_id = (int32_t)obj->id;
    }
    for (const vmx_sphere &s : sc.spheres) { /* :377-499, table order */
        test = sphere_hit(o, d, v3(s.centre[0], s.centre[1], s.centre[2]), s.radius);
        if (test > 0.f && test < nearest) {
            nearest = test;
            if (s.flags & VMX_SPHERE_EMIT) r.colour = v3(s.colour[0], s.colour[1], s.colour[2]);
            V3 nc = v3(s.normal_centre[0], s.normal_centre[1], s.normal_centre[2]);
            V3 n = normalize(o + (d * nearest) - nc);
            r.normal = (s.normal_sign < 0.f) ? -n : n;
        }
    }
    if (hit_mesh >= 0) r.material = true; /* :502-503 */
    r.location = o + (d * nearest);       /* :505 */
    r.distance = nearest;                 /* :507 */
    r.hit = nearest < INFINITY;           /* :508 */
    return r;
}

/* ------------------------------------------------------------------ */
/* RNG                                                                 */
/* ------------------------------------------------------------------ */
inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t splitmix64(uint64_t &x) {
    x += 0x9E3779B97F4A7C15ull;
    return mix64(x);
}
inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* keyed xoshiro256** stream: state = 4 splitmix64 outputs from
 * x0 = mix64(seed ^ (pixel << 32 | k)).  (DESIGN.md "RNG") */
struct Xoshiro {
    uint64_t s0, s1, s2, s3;
    /* Stream of sample k of pixel p — this build's own definition (the reference seeds a thread-local mt19937_64 from
     * std::random_device, pathtracer.cpp:231): two mix64 per pixel, three 64-bit multiplies per sample (the kernels key
     * the pixel half once per wave; vmx_kernels.hip: rng_pixel_key / rng_init_keyed) */
    void init(uint64_t seed, uint32_t pixel, uint32_t k) {
        const uint64_t a = mix64(seed ^ ((uint64_t)pixel << 32));
        const uint64_t b = mix64(a + 0x9E3779B97F4A7C15ull);
        s0 = mix64(a + (uint64_t)k);
        const uint64_t t = (s0 ^ b) * 0xD6E8FEB86659FD93ull;
        s1 = t ^ (t >> 32);
        s2 = rotl64(s0, 24) ^ b;
        s3 = rotl64(s1, 37) ^ a;
    }
    uint64_t next() {
        uint64_t r = rotl64(s1 * 5, 7) * 9;
        uint64_t t = s1 << 17;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = rotl64(s3, 45);
        return r;
    }
    /* uniform double in [0,1): 52 mantissa bits */
    double u01() {
        uint64_t b = 0x3FF0000000000000ull | (next() >> 12);
        double dd;
        std::memcpy(&dd, &b, 8);
        return dd - 1.0;
    }
    /* uniform_real_distribution<float>(0, 0.5) stand-in: 23 mantissa bits */
    float jitter() {
        uint32_t b = 0x3F800000u | (uint32_t)(next() >> 41);
        float f;
        std::memcpy(&f, &b, 4);
        return (f - 1.0f) * 0.5f;
    }
};

/* the reference's generator: std::mt19937_64 + std distributions, pathtracer.cpp:23,230-231 */
struct MtRng {
    std::mt19937_64 eng;
    std::uniform_real_distribution<double> dd{0.0, 1.0};
    std::uniform_real_distribution<float> df{0, 0.5};
    bool seeded = false;
    double u01() { return dd(eng); }
    float jitter() { return df(eng); }
};

/* Audit of the argument behind the kernels' two-phase shading and VMX_SAMPLING_ELIDE_DEAD (DESIGN_HISTORY.md 5.1), made on the
 * oracle's own Radiance: before every step, from a COPY of the stream, predict "this is the path's last step whatever
 * it hits" per value of the material flag, and "no light sphere can colour it"; after the step, check what happened. */
struct ElisionAudit {
    uint64_t steps = 0;             /* Radiance steps audited */
    uint64_t predicted_last = 0;    /* steps predicted to be the last one for the material flag the hit turned out to have */
    uint64_t predicted_dead = 0;    /* steps predicted last for BOTH values, with no light sphere in reach: the ray is not needed */
    uint64_t not_last = 0;          /* violations: predicted last, but the path went on with a direction that is not all NaN */
    uint64_t dead_changed = 0;      /* violations: predicted dead, but accumColour.rgb changed */
    uint64_t colour_mismatch = 0;   /* violations: predicted last, but accumColour is not before + accumRadiance * (hitColour
                                       as decided by the sphere table up to its last light) */
};
struct PathStats {
    uint64_t rays_primary = 0, rays_secondary = 0, tri_hits = 0, continued = 0;
    Counters cnt;
    ElisionAudit *audit = nullptr;
};

/* bit test, so that the -Ofast (finite-math-only) baseline build cannot fold it away */
inline bool finite1(float f) {
    uint32_t b;
    std::memcpy(&b, &f, 4);
    return (b & 0x7F800000u) != 0x7F800000u;
}
inline bool finite3(V3 v) { return finite1(v.x) && finite1(v.y) && finite1(v.z); }

/* VermiTexture::Sample, meshEngine.cpp:21-46: wrap by x - floor(x), nearest by round(x*(W-1)) */
inline void texture_sample_of(const float *tex, uint32_t tex_w, uint32_t tex_h, uint32_t tex_c, V2 uv, V4 *out) {
    float sx = uv.x - std::floor(uv.x);
    float sy = uv.y - std::floor(uv.y);
    uint32_t mx = (uint32_t)std::round(sx * (tex_w - 1));
    uint32_t my = (uint32_t)std::round(sy * (tex_h - 1));
    const float *ptr = &tex[((size_t)my * tex_w + mx) * tex_c];
    switch (tex_c) {
        case 1: *out = V4{ptr[0], ptr[0], ptr[0], ptr[0]}; break;
        case 2: *out = V4{ptr[0], ptr[1], 0, 0}; break;
        case 3: *out = V4{ptr[0], ptr[1], ptr[2], 0}; break;
        case 4: *out = V4{ptr[0], ptr[1], ptr[2], ptr[3]}; break;
        default:;
    }
}
inline void texture_sample(const orc_scene &sc, V2 uv, V4 *out) {
    texture_sample_of(sc.tex.data(), sc.tex_w, sc.tex_h, sc.tex_c, uv, out);
}

/* ---- cos / sin of a float argument -------------------------------------------------------------------
 * pathtracer.cpp:155,162 call the UNQUALIFIED cos(r1) / sin(r1) with `float r1`.  Which function that names
 * depends on whether <cmath>'s float overloads are visible in the global namespace — the same question as the
 * unqualified abs(float) of integrators.cpp:170 (render_bruteforce below).  One premise, one default: the
 * overloads are visible, so abs -> std::abs(float) and cos/sin -> cosf/sinf; the other reading (C's
 * `double cos(double)` on the widened argument, abs(int)) is selected by VMX_SAMPLING_LIBM_DOUBLE /
 * VMX_BF_ABS_INT.  sqrt(float) and fabs(float) at :157,160,162 give the same value under both readings
 * (sqrt is correctly rounded: narrowing the double root of a float is the float root).
 *
 * cosf / sinf are third-party arithmetic: the reference links the platform's libm (glibc on the Linux it is
 * built for, .travis.yml).  Restated here is glibc's algorithm since 2.28 (sysdeps/ieee754/flt-32/s_sinf.c,
 * s_cosf.c, sincosf.h = ARM optimized-routines' sinf/cosf): argument < pi/4: polynomial in double; else
 * n = round(x * 2/pi) via the 2^24-scaled product, x - n * (pi/2) in double, sign and polynomial by quadrant.
 * Only the range r1 can take, [0, float(2 pi)], is restated (the |x| >= 120 path is not).  Pinned:
 * orc_trig_compare_libm() compares every float of that range with THIS image's libm (glibc 2.35) — 0 of
 * 1,086,918,620 differ for both functions (tests/test_oracle.py); the kernels run the same operations. */
struct SinCosTab {
    double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3;
};
const SinCosTab kSinCos[2] = {
    {0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
     -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
     0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
inline float sincosf_poly(double x, double x2, const SinCosTab &p, int n) {
    if ((n & 1) == 0) {
        const double x3 = x * x2, s1 = p.s2 + x2 * p.s3, x7 = x3 * x2, s = x + x3 * p.s1;
        return (float)(s + x7 * s1);
    }
    const double x4 = x2 * x2, c2 = p.c3 + x2 * p.c4, c1 = p.c0 + x2 * p.c1, x6 = x4 * x2, c = c1 + x4 * p.c2;
    return (float)(c + x6 * c2);
}
inline uint32_t abstop12(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return (u >> 20) & 0x7ffu;
}
/* which = 0: sinf, 1: cosf; 0 <= y <= 2 pi */
inline float libm_sincosf(float y, int which) {
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return which ? 1.0f : y;
        return sincosf_poly(x, x * x, kSinCos[0], which);
    }
    const double r = x * kSinCos[0].hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - n * kSinCos[0].hpi;
    static const double sign[4] = {1.0, -1.0, -1.0, 1.0};
    return sincosf_poly(x * sign[n & 3], x * x, kSinCos[(n >> 1) & 1], n ^ which);
}
inline float restated_sinf(float y) { return libm_sincosf(y, 0); }
inline float restated_cosf(float y) { return libm_sincosf(y, 1); }

/* The prediction half of ElisionAudit: everything here is derived from the path's state BEFORE the step — the stream
 * (copied), the depth, the ray — exactly what the kernels have when they decide (vmx_kernels.hip: step_is_dead,
 * camera_step_ends, k_shade_ends). */
struct StepPrediction {
    bool last_mat, last_nomat, light_in_reach, throughput_finite;
    V4 before, radiance_before;
    V3 o, d;
    template <class Rng>
    void make(const orc_scene &sc, const Rng &stream, short depth, double r2scale, V3 ro, V3 rd, V4 accumColour,
              V4 accumRadiance) {
        Rng r = stream; /* a copy: the path's own stream is not advanced */
        bool rr_end = false;
        const int d1 = depth + 1;
        if (d1 > 5) rr_end = r.u01() > 0.95f || d1 > 1000; /* :56-59 */
        const double a = r.u01(), b = r.u01(), c = r.u01();
        /* material (:98-165): draw a picks the mirror branch, then r1 = b, r2 = float(r2scale * c); sqrt(1 - r2) is NaN */
        last_mat = rr_end || (!(a >= 0.96) && (1 - (float)(r2scale * c)) < 0);
        /* no material (:166-196): r1 = a, r2 = r2scale * b in double */
        last_nomat = rr_end || (1 - r2scale * b) < 0;
        light_in_reach = false;
        for (const vmx_sphere &s : sc.spheres)
            if ((s.flags & VMX_SPHERE_EMIT) && sphere_hit(ro, rd, v3(s.centre[0], s.centre[1], s.centre[2]), s.radius) > 0.f)
                light_in_reach = true;
        throughput_finite = finite1(accumRadiance.x) && finite1(accumRadiance.y) && finite1(accumRadiance.z);
        before = accumColour, radiance_before = accumRadiance, o = ro, d = rd;
    }
    static bool same_rgb(V4 p, V4 q) {
        return std::memcmp(&p.x, &q.x, 4) == 0 && std::memcmp(&p.y, &q.y, 4) == 0 && std::memcmp(&p.z, &q.z, 4) == 0;
    }
    /* ended: Radiance returned in this step; else next_dir is the direction it goes on with */
    void check(const orc_scene &sc, ElisionAudit &au, const CastOut &c, V3, V3, V4 after, bool ended, V3 next_dir) const {
        au.steps++;
        const bool last = c.material ? last_mat : last_nomat;
        const bool all_nan = next_dir.x != next_dir.x && next_dir.y != next_dir.y && next_dir.z != next_dir.z;
        if (last) {
            au.predicted_last++;
            /* an all-NaN direction misses the tree and every sphere: the next RayCast returns false, Radiance accumColour */
            if (!ended && !all_nan) au.not_last++;
            /* hitColour as the sphere table's entries up to the last light decide it, nothing else of the hit */
            size_t prefix = 0;
            for (size_t i = 0; i < sc.spheres.size(); ++i)
                if (sc.spheres[i].flags & VMX_SPHERE_EMIT) prefix = i + 1;
            float nearest = c.tri_id >= 0 ? c.tri_t : INFINITY;
            V3 col = {0, 0, 0};
            for (size_t i = 0; i < prefix; ++i) {
                const vmx_sphere &s = sc.spheres[i];
                const float t = sphere_hit(o, d, v3(s.centre[0], s.centre[1], s.centre[2]), s.radius);
                if (t > 0.f && t < nearest) {
                    nearest = t;
                    if (s.flags & VMX_SPHERE_EMIT) col = v3(s.colour[0], s.colour[1], s.colour[2]);
                }
            }
            if (throughput_finite) {
                const V4 expect = before + radiance_before * V4{col.x, col.y, col.z, 0.f};
                if (!same_rgb(expect, after)) au.colour_mismatch++;
            }
        }
        if (last_mat && last_nomat && !light_in_reach && throughput_finite) {
            au.predicted_dead++;
            if (!same_rgb(before, after)) au.dead_changed++;
            if (!ended && !all_nan) au.not_last++;
        }
    }
};

/* Radiance, pathtracer.cpp:21-198.  No texture is bound in any configuration
 * (pathtracer.cpp:63-66 not taken), so sampleColour is (1,1,1,1) (:75-79) unless a
 * texture was bound with orc_scene_bind_texture (then :63-66, VermiTexture::Sample). */
template <class Rng>
V4 radiance(const orc_scene &sc, V3 rStart, V3 rDir, Rng &rng, uint32_t sampling, PathStats *st,
            bool count_nodes) {
    V4 accumColour = {0, 0, 0, -100};
    V4 accumRadiance = {1, 1, 1, 1};
    short depth = 0;
    const double r2scale = ((sampling & VMX_SAMPLING_MODE_MASK) == VMX_SAMPLING_CORRECTED) ? 1.0 : 10.0;
    const bool libm_double = (sampling & VMX_SAMPLING_LIBM_DOUBLE) != 0;
    ElisionAudit *const audit = st ? st->audit : nullptr;
    StepPrediction pred{};
    while (1) {
        bool is_ray = finite3(rDir); /* NaN directions are not rays */
        if (st) {
            if (depth == 0)
                st->rays_primary++;
            else if (is_ray)
                st->rays_secondary++;
        }
        if (audit) pred.make(sc, rng, depth, r2scale, rStart, rDir, accumColour, accumRadiance);
        CastOut c = ray_cast(sc, rStart, rDir, (st && count_nodes && is_ray) ? &st->cnt : nullptr);
        if (!c.hit) {
            if (audit) pred.check(sc, *audit, c, rStart, rDir, accumColour, true, rDir);
            return accumColour; /* :36-41 */
        }
        if (st && c.tri_id >= 0) st->tri_hits++;
        accumColour = accumColour + accumRadiance * V4{c.colour.x, c.colour.y, c.colour.z, 0.f}; /* :43 */
        if (depth == 0) accumColour.w = c.distance;                                            /* :44-47 */
        if (length(c.colour) > 1.f) {
            if (audit) pred.check(sc, *audit, c, rStart, rDir, accumColour, true, rDir);
            return accumColour; /* :52 */
        }
        if (++depth > 5 && (rng.u01() > 0.95f || depth > 1000)) {
            if (audit) pred.check(sc, *audit, c, rStart, rDir, accumColour, true, rDir);
            return accumColour; /* :56-59 */
        }
        V4 sampleColour = {0.f, 0.f, 0.f, 0.f}; /* :62 */
        if (c.material && sc.n_textures > 0)
            texture_sample(sc, c.uv, &sampleColour); /* :63-66 */
        else
            sampleColour = V4{1.f, 1.f, 1.f, 1.0}; /* :75-79 */
        V3 n = c.normal;
        V3 next_dir;
        if (c.material && rng.u01() >= 0.96) { /* :98-109 specular */
            (void)rng.u01();
            (void)rng.u01();
            (void)rng.u01(); /* :101-103, unused noise */
            rStart = c.location - rDir * 0.001f;
            next_dir = normalize(rDir - n * 2.f * dot(n, rDir));
        } else if (c.material) { /* :151-165 */
            accumRadiance = accumRadiance * sampleColour;
            float r1 = (float)(2 * M_PI * rng.u01());
            float r2 = (float)(r2scale * rng.u01());
            float r2s = std::sqrt(r2);
            V3 w = dot(n, rDir) < 0.f ? n : n * -1.f;
            V3 u = normalize(cross(std::fabs(w.x) > .1 ? v3(0, 1, 0) : v3(1, 0, 0), w));
            V3 v = cross(w, u);
            /* float(cos(r1)), float(sin(r1)) with float r1 (:162): cosf/sinf, or C's double functions (see above) */
            const float cs = libm_double ? (float)std::cos((double)r1) : restated_cosf(r1);
            const float sn = libm_double ? (float)std::sin((double)r1) : restated_sinf(r1);
            V3 dd = normalize(u * cs * r2s + v * sn * r2s + w * (float)std::sqrt(1 - r2));
            rStart = c.location - rDir * 0.001f;
            next_dir = dd;
        } else { /* :166-196, nearest hit is a sphere and the BVH hit nothing */
            double r1 = 2 * M_PI * rng.u01();
            double r2 = r2scale * rng.u01();
            float r2s = (float)std::sqrt(r2);
            (void)rng.u01();
            (void)rng.u01();
            (void)rng.u01(); /* :173-175, unused noise */
            V3 w = dot(n, rDir) < 0.f ? n : n * -1.f;
            V3 u = normalize(cross(std::fabs(w.x) > .1 ? v3(0, 1, 0) : v3(1, 0, 0), w));
            V3 v = cross(w, u);
            V3 dd = normalize(u * (float)std::cos(r1) * r2s + v * (float)std::sin(r1) * r2s +
                              w * (float)std::sqrt(1 - r2));
            rStart = c.location - rDir * 0.001f;
            next_dir = dd;
        }
        if (audit) pred.check(sc, *audit, c, rStart, rDir, accumColour, false, next_dir);
        rDir = next_dir;
        if (st && finite3(rDir)) st->continued++;
    }
}

/* Camera ctor + camera matrix: camera.cpp:43-47, pathtracer.cpp:216-221 with
 * glm::rotate (gtc/matrix_transform) restated.  Column-major 3x3. */
struct Mat3 {
    V3 c0, c1, c2;
};
inline Mat3 rotate(const Mat3 &m, float angle, V3 axis_in) {
    float c = std::cos(angle), s = std::sin(angle);
    V3 axis = normalize(axis_in);
    V3 temp = axis * (1.0f - c);
    float r00 = c + temp.x * axis.x, r01 = temp.x * axis.y + s * axis.z, r02 = temp.x * axis.z - s * axis.y;
    float r10 = temp.y * axis.x - s * axis.z, r11 = c + temp.y * axis.y, r12 = temp.y * axis.z + s * axis.x;
    float r20 = temp.z * axis.x + s * axis.y, r21 = temp.z * axis.y - s * axis.x, r22 = c + temp.z * axis.z;
    Mat3 out;
    out.c0 = m.c0 * r00 + m.c1 * r01 + m.c2 * r02;
    out.c1 = m.c0 * r10 + m.c1 * r11 + m.c2 * r12;
    out.c2 = m.c0 * r20 + m.c1 * r21 + m.c2 * r22;
    return out;
}
inline Mat3 camera_matrix(const vmx_camera &cam) {
    /* camera.cpp:43-47: double arithmetic narrowed to float */
    float rx = (float)(-cam.rotation_deg[0] * 3.1415926535 / 180);
    float ry = (float)(-cam.rotation_deg[1] * 3.1415926535 / 180);
    float rz = (float)(cam.rotation_deg[2] * 3.1415926535 / 180);
    /* VMX_ROTATION_RADIANS: the caller holds Camera::mRotation itself (what pathtracer.cpp:219-221 reads) */
    if (cam.rotation_units == VMX_ROTATION_RADIANS) rx = cam.rotation_rad[0], ry = cam.rotation_rad[1], rz = cam.rotation_rad[2];
    Mat3 m = {v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)};
    m = rotate(m, ry, v3(0, 1, 0)); /* pathtracer.cpp:219 */
    m = rotate(m, rx, v3(1, 0, 0)); /* :220 */
    m = rotate(m, rz, v3(0, 0, 1)); /* :221 */
    return m;
}

/* pathtracer.cpp:251-280 */
inline V3 primary_dir(const vmx_camera &cam, const Mat3 &M, uint64_t p, uint32_t sampleX,
                      uint32_t sampleY, float jx, float jy) {
    uint32_t W = cam.image_res[0], H = cam.image_res[1];
    float hx = (float)(((float(p % W) + (sampleX * 0.5f - 0.5f) + jx - 0.25) / W) - 0.5);
    float hy = (float)(((float(p / W) + (sampleY * 0.5f - 0.5f) + jy - 0.25) / H) - 0.5);
    float bx = hx * cam.back_size[0];
    float by = hy * cam.back_size[1];
    float gx = bx, gy = -by, gz = -cam.back_distance; /* gridPane, w = 1 */
    /* mat4*vec4 = (m0*x + m1*y) + (m2*z + m3*w); m3 = (0,0,0,1) */
    V3 raw = (M.c0 * gx + M.c1 * gy) + (M.c2 * gz + v3(0, 0, 0) * 1.0f);
    float w = (0.f * gx + 0.f * gy) + (0.f * gz + 1.f * 1.0f);
    return normalize(raw / w);
}

template <class Rng, class InitFn>
void render_rows(const orc_scene &sc, const vmx_camera &cam, const vmx_opts &opts, float *out,
                 vmx_stats *stats, int threads, InitFn init_rng) {
    const uint32_t W = cam.image_res[0], H = cam.image_res[1], spp = cam.rays_per_pixel;
    /* Pixel subset (test infrastructure: the reference always renders every pixel).  With opts.world > 1 only the rows
     * of the stripes s with s % world == rank are rendered — the sharding of vmx_render — and `out` holds those rows
     * packed in ascending order; pixels are independent (pathtracer.cpp:226-227), sample streams are keyed by the GLOBAL
     * pixel index, so a row here equals the same row of the whole frame.  This is what lets the tests put 1/16 of a
     * full-size frame (BASELINE.json configs 3, 4) against the HIP path in seconds. */
    const uint32_t world = opts.world <= 1 ? 1u : opts.world, rank = opts.world <= 1 ? 0u : opts.rank;
    const uint32_t stripe = opts.stripe_rows ? opts.stripe_rows : 16u;
    std::vector<uint32_t> rows;
    for (uint32_t y = 0; y < H; ++y)
        if ((y / stripe) % world == rank) rows.push_back(y);
    const uint64_t npix = (uint64_t)W * rows.size();
    const Mat3 M = camera_matrix(cam);
    const V3 origin = v3(cam.position[0], cam.position[1], cam.position[2]);
    const bool count_nodes = opts.collect_counters != 0;
    uint64_t t_prim = 0, t_sec = 0, t_samples = 0, t_inner = 0, t_tris = 0, t_hits = 0, t_cont = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel reduction(+ : t_prim, t_sec, t_samples, t_inner, t_tris, t_hits, t_cont)
    {
        Rng rng;
        PathStats st;
#pragma omp for schedule(dynamic, 1)
        for (uint64_t lp = 0; lp < npix; ++lp) { /* pathtracer.cpp:226-227 */
            const uint64_t p = (uint64_t)rows[lp / W] * W + lp % W; /* global pixel index (all of them when world <= 1) */
            V4 accum = {0, 0, 0, 0};
            uint32_t nTotal = 0;
            for (uint16_t sx = 0; sx < 2; ++sx) {
                for (uint16_t sy = 0; sy < 2; ++sy) {
                    for (uint32_t sz = 0; sz < (spp / 4); ++sz) { /* :242-247 */
                        ++nTotal;
                        uint32_t k = (uint32_t)(sx * 2 + sy) * (spp / 4) + sz;
                        init_rng(rng, p, k);
                        float jx = rng.jitter(); /* :251 */
                        float jy = rng.jitter(); /* :252 */
                        V3 dir = primary_dir(cam, M, p, sx, sy, jx, jy);
                        V4 s = radiance(sc, origin, dir, rng, opts.sampling, &st, count_nodes);
                        accum = accum + s; /* :283 */
                        if (opts.early_stop && nTotal > std::sqrt((double)spp)) { /* :290-311 */
                            V3 a = v3(accum.x, accum.y, accum.z) / float(nTotal);
                            V3 b = v3(accum.x + s.x, accum.y + s.y, accum.z + s.z) / float(nTotal + 1);
                            if (std::fabs(length(a - b)) < 0.00001f) break;
                        }
                    }
                }
            }
            float *px = out + lp * 5; /* :318-324, camera.cpp:106-113 */
            px[0] = std::max(std::min(accum.x / nTotal, 1.f), 0.f);
            px[1] = std::max(std::min(accum.y / nTotal, 1.f), 0.f);
            px[2] = std::max(std::min(accum.z / nTotal, 1.f), 0.f);
            px[3] = 1.f;
            px[4] = (float)nTotal;
            t_samples += nTotal;
        }
        t_prim += st.rays_primary;
        t_sec += st.rays_secondary;
        t_inner += st.cnt.inner_visits;
        t_tris += st.cnt.tri_tests;
        t_hits += st.tri_hits;
        t_cont += st.continued;
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays_primary = t_prim;
        stats->rays_secondary = t_sec;
        stats->samples = t_samples;
        /* the oracle does not split counters by stage: totals go to `primary` */
        stats->primary.rays = t_prim + t_sec;
        stats->primary.inner_visits = t_inner;
        stats->primary.tri_tests = t_tris;
        stats->primary.tri_hits = t_hits;
        stats->primary.continued = t_cont;
    }
}

/* BruteForceTracer::Render, core/integrators/integrators.cpp:9-186 — the engine's default
 * integrator (renderEngine.cpp:51): N.L from a point light at (500,1100,2000), one mirror probe,
 * optional normal perturbation by boundTextures[0] and albedo from boundTextures[1], and a
 * convergence break once more than 2 samples are in.
 *
 * Choices where the reference leaves behaviour open (stated in DESIGN_HISTORY.md §8 f-4 as well):
 *  - RNG.  The reference shares ONE std::mt19937 seeded with time(0) between all OpenMP threads
 *    without synchronisation (:30,65-66): not reproducible even run to run.  As for PathTracer, sample
 *    `s` of pixel `p` draws its two jitters from the keyed stream (seed, p, s).
 *  - `hitMaterial->mNumProperties` (:93-96) is read although RayCast leaves the pointer null when only
 *    a sphere was hit (meshEngine.cpp:250-251,502-503); the loop body is empty, so no value of it can
 *    reach the image and an optimising build drops the read.  Not restated.
 *  - unqualified `abs(float)` (:170) binds to std::abs(float) when <cmath>'s overloads are visible in
 *    the global namespace (they are with libstdc++ once <stdlib.h>/<math.h> come in, as they do through
 *    GLM/Assimp) and to C's abs(int) otherwise: VMX_BF_ABS_INT selects the second reading.
 *  - `resp` (:117-131) is computed and never used (:144 has it commented out): not restated.
 * depth = hitDistance of the pixel's LAST sample (:181; INFINITY after a miss, meshEngine.cpp:507). */
void render_bruteforce(const orc_scene &sc, const vmx_camera &cam, const vmx_opts &opts, uint32_t flags, float *out,
                       vmx_stats *stats, int threads) {
    const uint32_t W = cam.image_res[0], H = cam.image_res[1], spp = cam.rays_per_pixel;
    const uint64_t npix = (uint64_t)W * H;
    const Mat3 M = camera_matrix(cam); /* :23-28, same three rotations as pathtracer.cpp:216-221 */
    const V3 origin = v3(cam.position[0], cam.position[1], cam.position[2]);
    const V3 lightLocation = v3(500, 1100, 2000); /* :16 */
    uint64_t t_prim = 0, t_sec = 0, t_samples = 0, t_hits = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : t_prim, t_sec, t_samples, t_hits)
    for (uint64_t p = 0; p < npix; ++p) { /* :32-33 */
        float hitDistance = 0.f;
        V4 accum = {0, 0, 0, 0};
        uint32_t nTotalSamples = 0;
        V4 lastSampleColour = {0, 0, 0, 0};
        V4 currentSampleColour = {0, 0, 0, 0};
        for (uint16_t sample = 0; sample < spp; ++sample) { /* :59 */
            ++nTotalSamples;
            Xoshiro rng;
            rng.init(opts.seed, (uint32_t)p, sample);
            const float jx = rng.jitter(), jy = rng.jitter();
            float homogenousX = (float)(((float(p % W) + jx - 0.25) / W) * 2 - 1); /* :65 */
            float homogenousY = (float)(((float(p / W) + jy - 0.25) / H) * 2 - 1); /* :66 */
            float cameraBackXCm = (float)(homogenousX * cam.back_size[0] * 0.5);   /* :73 */
            float cameraBackYCm = (float)(homogenousY * cam.back_size[1] * 0.5);   /* :74 */
            float gx = cameraBackXCm, gy = -cameraBackYCm, gz = -cam.back_distance; /* :76, w = 1 */
            V3 raw = (M.c0 * gx + M.c1 * gy) + (M.c2 * gz + v3(0, 0, 0) * 1.0f);  /* :79 */
            float w = (0.f * gx + 0.f * gy) + (0.f * gz + 1.f * 1.0f);
            V3 dir = normalize(raw / w); /* :81 */
            ++t_prim;
            CastOut c = ray_cast(sc, origin, dir, nullptr); /* :81 */
            hitDistance = c.distance;
            if (c.hit) {
                if (c.tri_id >= 0) ++t_hits;
                V3 hitNormal = c.normal;
                V3 lightDirectionVector = lightLocation - c.location; /* :83 */
                V3 lightDirection = normalize(lightDirectionVector);   /* :84 */
                float vNDL = dot(lightDirection, normalize(hitNormal)); /* :88 */
                if (sc.n_textures > 0) { /* :98-106 */
                    V4 normal = {0, 0, 0, 0};
                    texture_sample(sc, c.uv, &normal);
                    hitNormal = hitNormal + v3(normal.x, normal.y, normal.z);
                    vNDL = dot(lightDirection, normalize(hitNormal));
                }
                /* :119  -L - 2.f * N * dot(N, -L): (2.f * N) is formed first, then scaled */
                V3 negL = -lightDirection;
                V3 secondBounceDirection = negL - (v3(2.f * hitNormal.x, 2.f * hitNormal.y, 2.f * hitNormal.z) * dot(hitNormal, negL));
                if (finite3(secondBounceDirection)) ++t_sec;
                CastOut c2 = ray_cast(sc, c.location, secondBounceDirection, nullptr); /* :121 */
                if (!c2.hit) { /* :133-137 */
                    vNDL *= 0.9f;
                    vNDL += 0.1f;
                }
                if (sc.n_textures > 1) { /* :141-147 */
                    texture_sample_of(sc.tex1.data(), sc.tex1_w, sc.tex1_h, sc.tex1_c, c.uv, &currentSampleColour);
                    currentSampleColour = V4{currentSampleColour.x * vNDL, currentSampleColour.y * vNDL,
                                             currentSampleColour.z * vNDL, currentSampleColour.w * vNDL};
                    currentSampleColour.w = 1.0;
                } else { /* :148-156 */
                    currentSampleColour = V4{0.890196078f * vNDL, 0.258823529f * vNDL, 0.203921569f * vNDL, 1.0};
                }
                accum = accum + currentSampleColour; /* :158 */
            }
            if (nTotalSamples > 2) { /* :167-172 */
                const float n = float(nTotalSamples); /* glm::vec4(nTotalSamples) */
                lastSampleColour = V4{lastSampleColour.x - accum.x / n, lastSampleColour.y - accum.y / n,
                                      lastSampleColour.z - accum.z / n, lastSampleColour.w - accum.w / n};
                const float sum = lastSampleColour.x + lastSampleColour.y + lastSampleColour.z + lastSampleColour.w;
                float mag;
                if (flags & VMX_BF_ABS_INT) {
                    /* abs(int): the float converts to int first (values outside int's range: undefined there, 0 here) */
                    const double tr = std::trunc((double)sum);
                    mag = (tr >= -2147483648.0 && tr <= 2147483647.0) ? (float)std::abs((int)tr) : 0.f;
                } else {
                    mag = std::fabs(sum);
                }
                if (mag < 0.001f) break;
            }
            {
                const float n = float(nTotalSamples); /* :173 */
                lastSampleColour = V4{accum.x / n, accum.y / n, accum.z / n, accum.w / n};
            }
        }
        float *px = out + p * 5; /* :176-183, camera.cpp:106-113 (RGBAZ) */
        px[0] = std::max(std::min(accum.x / nTotalSamples, 1.f), 0.f);
        px[1] = std::max(std::min(accum.y / nTotalSamples, 1.f), 0.f);
        px[2] = std::max(std::min(accum.z / nTotalSamples, 1.f), 0.f);
        px[3] = accum.w / nTotalSamples;
        px[4] = hitDistance;
        t_samples += nTotalSamples;
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays_primary = t_prim;
        stats->rays_secondary = t_sec;
        stats->samples = t_samples;
        stats->primary.rays = t_prim + t_sec;
        stats->primary.tri_hits = t_hits;
    }
}

} // namespace

/* ------------------------------------------------------------------ */
/* C API                                                               */
/* ------------------------------------------------------------------ */
extern "C" {

const char *orc_build_flags(void) {
#ifdef ORC_BUILD_FLAGS
    return ORC_BUILD_FLAGS;
#else
    return "unknown";
#endif
}

const vmx_sphere *orc_default_spheres(uint32_t *count) {
    if (count) *count = 8;
    return kDefaultSpheres;
}

orc_scene *orc_scene_create(const float *pos, const float *nrm, const float *uv, uint32_t ntris,
                            const vmx_sphere *spheres, uint32_t nspheres, uint32_t leaf_size) {
    if (!pos || !nrm || ntris == 0) return nullptr;
    orc_scene *sc = new orc_scene();
    sc->leaf_size = leaf_size ? leaf_size : 4;
    sc->tris.resize(ntris);
    for (uint32_t i = 0; i < ntris; ++i) {
        Tri &t = sc->tris[i];
        const float *p = pos + (size_t)i * 9, *n = nrm + (size_t)i * 9;
        t.v0 = v3(p[0], p[1], p[2]);
        t.v1 = v3(p[3], p[4], p[5]);
        t.v2 = v3(p[6], p[7], p[8]);
        t.n0 = v3(n[0], n[1], n[2]);
        t.n1 = v3(n[3], n[4], n[5]);
        t.n2 = v3(n[6], n[7], n[8]);
        if (uv) {
            const float *q = uv + (size_t)i * 6;
            t.t0 = {q[0], q[1]};
            t.t1 = {q[2], q[3]};
            t.t2 = {q[4], q[5]};
        } else {
            t.t0 = t.t1 = t.t2 = {0, 0};
        }
        t.id = i;
    }
    sc->prims.resize(ntris);
    for (uint32_t i = 0; i < ntris; ++i) sc->prims[i] = &sc->tris[i];
    if (spheres == nullptr && nspheres == 0)
        sc->spheres.assign(kDefaultSpheres, kDefaultSpheres + 8);
    else
        sc->spheres.assign(spheres, spheres + nspheres);
    build_bvh(*sc);
    return sc;
}

/* A scene whose flat tree is given (reference layout, bvh.h:11-14) instead of built: lets the tests
 * run the reference's *traversal* over another builder's tree (SURVEY §8 f-1 quality builder). */
orc_scene *orc_scene_create_from_tree(const float *pos, const float *nrm, const float *uv, uint32_t ntris,
                                      const vmx_sphere *spheres, uint32_t nspheres, uint32_t n_nodes,
                                      const uint32_t *start, const uint32_t *nprims, const uint32_t *right_offset,
                                      const float *bbox, const uint32_t *prim_order) {
    orc_scene *sc = orc_scene_create(pos, nrm, uv, ntris, spheres, nspheres, 4);
    if (!sc) return nullptr;
    sc->nodes.assign(n_nodes, FlatNode{});
    sc->n_leaves = 0;
    for (uint32_t i = 0; i < n_nodes; ++i) {
        FlatNode &n = sc->nodes[i];
        n.start = start[i], n.nprims = nprims[i], n.right_offset = right_offset[i];
        const float *b = bbox + (size_t)i * 6;
        n.box = box_of(v3(b[0], b[1], b[2]), v3(b[3], b[4], b[5]));
        if (n.right_offset == 0) sc->n_leaves++;
    }
    for (uint32_t i = 0; i < ntris; ++i) sc->prims[i] = &sc->tris[prim_order[i]];
    return sc;
}

void orc_scene_destroy(orc_scene *sc) { delete sc; }

/* MeshEngine::bindTexture (meshEngine.cpp:74-93) without the OIIO read: only boundTextures[0] is
 * ever sampled (pathtracer.cpp:65) */
int orc_scene_bind_texture(orc_scene *sc, const float *data, uint32_t w, uint32_t h, uint32_t c) {
    if (!sc || !data || w == 0 || h == 0 || c == 0 || c > 4 || w > 65535 || h > 65535) return 1;
    if (sc->n_textures == 0) {
        sc->tex.assign(data, data + (size_t)w * h * c);
        sc->tex_w = (uint16_t)w, sc->tex_h = (uint16_t)h, sc->tex_c = (uint16_t)c;
    } else if (sc->n_textures == 1) { /* boundTextures[1]: BruteForceTracer's albedo (integrators.cpp:141-147) */
        sc->tex1.assign(data, data + (size_t)w * h * c);
        sc->tex1_w = (uint16_t)w, sc->tex1_h = (uint16_t)h, sc->tex1_c = (uint16_t)c;
    }
    sc->n_textures++;
    return 0;
}

void orc_scene_describe(const orc_scene *sc, uint32_t *n_nodes, uint32_t *n_leaves,
                        uint32_t *max_depth) {
    if (n_nodes) *n_nodes = (uint32_t)sc->nodes.size();
    if (n_leaves) *n_leaves = sc->n_leaves;
    if (max_depth) *max_depth = sc->max_depth;
}

void orc_scene_bvh(const orc_scene *sc, uint32_t *start, uint32_t *nprims, uint32_t *right_offset,
                   float *bbox, uint32_t *prim_order) {
    for (size_t i = 0; i < sc->nodes.size(); ++i) {
        const FlatNode &n = sc->nodes[i];
        if (start) start[i] = n.start;
        if (nprims) nprims[i] = n.nprims;
        if (right_offset) right_offset[i] = n.right_offset;
        if (bbox) {
            float *b = bbox + i * 6;
            b[0] = n.box.lo.x, b[1] = n.box.lo.y, b[2] = n.box.lo.z;
            b[3] = n.box.hi.x, b[4] = n.box.hi.y, b[5] = n.box.hi.z;
        }
    }
    if (prim_order)
        for (size_t i = 0; i < sc->prims.size(); ++i) prim_order[i] = sc->prims[i]->id;
}

void orc_trace(const orc_scene *sc, const float *o, const float *d, uint32_t n, int32_t *tri_id,
               float *t, orc_trace_counters *counters) {
    uint64_t iv = 0, tt = 0, pp = 0, ms = 0;
#pragma omp parallel for schedule(static) reduction(+ : iv, tt, pp) reduction(max : ms)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        RayQ r = make_ray(v3(o[i * 3], o[i * 3 + 1], o[i * 3 + 2]), v3(d[i * 3], d[i * 3 + 1], d[i * 3 + 2]));
        Counters c;
        float bt;
        const Tri *obj;
        bool h = bvh_nearest(*sc, r, &bt, &obj, counters ? &c : nullptr);
        tri_id[i] = h ? (int32_t)obj->id : -1;
        t[i] = bt;
        iv += c.inner_visits, tt += c.tri_tests, pp += c.pops;
        ms = std::max(ms, c.max_stack);
    }
    if (counters) {
        counters->inner_visits = iv;
        counters->tri_tests = tt;
        counters->pops = pp;
        counters->max_stack = ms;
    }
}

void orc_raycast(const orc_scene *sc, const float *o, const float *d, uint32_t n, vmx_rayhit *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        CastOut c = ray_cast(*sc, v3(o[i * 3], o[i * 3 + 1], o[i * 3 + 2]),
                             v3(d[i * 3], d[i * 3 + 1], d[i * 3 + 2]), nullptr);
        vmx_rayhit &h = out[i];
        std::memset(&h, 0, sizeof(h));
        h.location[0] = c.location.x, h.location[1] = c.location.y, h.location[2] = c.location.z;
        h.distance = c.distance;
        h.normal[0] = c.normal.x, h.normal[1] = c.normal.y, h.normal[2] = c.normal.z;
        h.tri_id = c.tri_id;
        h.uv[0] = c.uv.x, h.uv[1] = c.uv.y;
        h.tri_t = c.tri_t;
        h.flags = (c.hit ? 1u : 0u) | (c.material ? 2u : 0u);
        h.colour[0] = c.colour.x, h.colour[1] = c.colour.y, h.colour[2] = c.colour.z;
    }
}

void orc_radiance(const orc_scene *sc, const float *o, const float *d, uint32_t n,
                  const vmx_opts *opts, float *out4, vmx_stats *stats) {
    uint64_t t_prim = 0, t_sec = 0, t_inner = 0, t_tris = 0, t_hits = 0, t_cont = 0;
    const bool count_nodes = opts->collect_counters != 0;
#pragma omp parallel for schedule(static) reduction(+ : t_prim, t_sec, t_inner, t_tris, t_hits, t_cont)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        Xoshiro rng;
        rng.init(opts->seed, (uint32_t)i, 0);
        (void)rng.next(); /* the two pixel-jitter draws of the stream are skipped */
        (void)rng.next();
        PathStats st;
        V4 r = radiance(*sc, v3(o[i * 3], o[i * 3 + 1], o[i * 3 + 2]),
                        v3(d[i * 3], d[i * 3 + 1], d[i * 3 + 2]), rng, opts->sampling, &st, count_nodes);
        out4[i * 4] = r.x, out4[i * 4 + 1] = r.y, out4[i * 4 + 2] = r.z, out4[i * 4 + 3] = r.w;
        t_prim += st.rays_primary, t_sec += st.rays_secondary, t_inner += st.cnt.inner_visits;
        t_tris += st.cnt.tri_tests, t_hits += st.tri_hits, t_cont += st.continued;
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays_primary = t_prim, stats->rays_secondary = t_sec;
        stats->primary.rays = t_prim + t_sec, stats->primary.inner_visits = t_inner;
        stats->primary.tri_tests = t_tris, stats->primary.tri_hits = t_hits;
        stats->primary.continued = t_cont;
        stats->samples = n;
    }
}

/* ElisionAudit over n explicit camera rays (paths keyed like orc_radiance): out[6] = steps, predicted_last, predicted_dead,
 * not_last, dead_changed, colour_mismatch; the radiance is written too and must equal orc_radiance's */
void orc_audit_elision(const orc_scene *sc, const float *o, const float *d, uint32_t n, const vmx_opts *opts, float *out4,
                       uint64_t *out6) {
    uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
#pragma omp parallel for schedule(static) reduction(+ : a0, a1, a2, a3, a4, a5)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        Xoshiro rng;
        rng.init(opts->seed, (uint32_t)i, 0);
        (void)rng.next();
        (void)rng.next();
        ElisionAudit au;
        PathStats st;
        st.audit = &au;
        V4 r = radiance(*sc, v3(o[i * 3], o[i * 3 + 1], o[i * 3 + 2]), v3(d[i * 3], d[i * 3 + 1], d[i * 3 + 2]), rng,
                        opts->sampling, &st, false);
        out4[i * 4] = r.x, out4[i * 4 + 1] = r.y, out4[i * 4 + 2] = r.z, out4[i * 4 + 3] = r.w;
        a0 += au.steps, a1 += au.predicted_last, a2 += au.predicted_dead, a3 += au.not_last, a4 += au.dead_changed,
            a5 += au.colour_mismatch;
    }
    out6[0] = a0, out6[1] = a1, out6[2] = a2, out6[3] = a3, out6[4] = a4, out6[5] = a5;
}

void orc_radiance_mt(const orc_scene *sc, const float *o, const float *d, uint32_t n,
                     const uint64_t *seeds, uint32_t sampling, float *out4) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        MtRng rng;
        rng.eng.seed(seeds[i]);
        V4 r = radiance(*sc, v3(o[i * 3], o[i * 3 + 1], o[i * 3 + 2]),
                        v3(d[i * 3], d[i * 3 + 1], d[i * 3 + 2]), rng, sampling, nullptr, false);
        out4[i * 4] = r.x, out4[i * 4 + 1] = r.y, out4[i * 4 + 2] = r.z, out4[i * 4 + 3] = r.w;
    }
}

void orc_camera_matrix(const vmx_camera *cam, float m9[9]) {
    Mat3 m = camera_matrix(*cam);
    m9[0] = m.c0.x, m9[1] = m.c0.y, m9[2] = m.c0.z;
    m9[3] = m.c1.x, m9[4] = m.c1.y, m9[5] = m.c1.z;
    m9[6] = m.c2.x, m9[7] = m.c2.y, m9[8] = m.c2.z;
}

void orc_primary_rays(const vmx_camera *cam, const vmx_opts *opts, uint32_t k, float *o, float *d) {
    const uint32_t W = cam->image_res[0], H = cam->image_res[1], Q = cam->rays_per_pixel / 4;
    const Mat3 M = camera_matrix(*cam);
    const uint32_t s = Q ? k / Q : 0, sx = s >> 1, sy = s & 1;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < (int64_t)W * H; ++p) {
        Xoshiro rng;
        rng.init(opts->seed, (uint32_t)p, k);
        float jx = rng.jitter(), jy = rng.jitter();
        V3 dir = primary_dir(*cam, M, (uint64_t)p, sx, sy, jx, jy);
        o[p * 3] = cam->position[0], o[p * 3 + 1] = cam->position[1], o[p * 3 + 2] = cam->position[2];
        d[p * 3] = dir.x, d[p * 3 + 1] = dir.y, d[p * 3 + 2] = dir.z;
    }
}

void orc_stream(uint64_t seed, uint32_t pixel, uint32_t k, uint32_t n, uint64_t *out) {
    Xoshiro r;
    r.init(seed, pixel, k);
    for (uint32_t i = 0; i < n; ++i) out[i] = r.next();
}

uint64_t orc_splitmix64(uint64_t *state) { return splitmix64(*state); }

/* Camera::saveFrame conversion loop, camera.cpp:159-163 (RGBAZ) */
/* the restated cosf / sinf (see libm_sincosf) for explicit arguments in [0, 2 pi] */
void orc_trig(const float *x, uint32_t n, float *cos_out, float *sin_out) {
    for (uint32_t i = 0; i < n; ++i) {
        cos_out[i] = restated_cosf(x[i]);
        sin_out[i] = restated_sinf(x[i]);
    }
}
/* how many floats with bit patterns in [lo_bits, hi_bits] (non-negative floats) the restatement and the HOST's
 * libm disagree on: the pin of the restatement against the platform library it restates */
void orc_trig_compare_libm(uint32_t lo_bits, uint32_t hi_bits, uint64_t *cos_diff, uint64_t *sin_diff) {
    uint64_t dc = 0, ds = 0;
    float (*volatile host_cosf)(float) = cosf; /* through pointers: the compiler must not fold the libm calls */
    float (*volatile host_sinf)(float) = sinf;
    float (*const hc)(float) = host_cosf;
    float (*const hs)(float) = host_sinf;
#pragma omp parallel for reduction(+ : dc, ds) schedule(static)
    for (uint64_t b = lo_bits; b <= (uint64_t)hi_bits; ++b) {
        const uint32_t bb = (uint32_t)b;
        float x;
        std::memcpy(&x, &bb, 4);
        const float c1 = hc(x), c2 = restated_cosf(x), s1 = hs(x), s2 = restated_sinf(x);
        if (std::memcmp(&c1, &c2, 4)) dc++;
        if (std::memcmp(&s1, &s2, 4)) ds++;
    }
    *cos_diff = dc;
    *sin_diff = ds;
}

/* The kernels divide by the image width / height (pathtracer.cpp:251-252, in double) with a reciprocal and one FMA
 * correction (vmx_kernels.hip: div_by_count).  For every float fx with bit pattern in [lo_bits, hi_bits] and both
 * signs: how often float(((double)fx - 0.25) / n - 0.5) — what the reference computes — differs from the same with
 * the three-operation quotient; also counts the cases where the double quotients themselves differ.  Must be 0 / 0. */
void orc_check_div_by_count(uint32_t n, uint32_t lo_bits, uint32_t hi_bits, uint64_t *float_diff, uint64_t *double_diff) {
    uint64_t df = 0, dd = 0;
    const double y = 1.0 / (double)n;
#pragma omp parallel for reduction(+ : df, dd) schedule(static)
    for (uint64_t b = lo_bits; b <= (uint64_t)hi_bits; ++b) {
        for (uint32_t sign = 0; sign < 2; ++sign) {
            const uint32_t bb = (uint32_t)b | (sign << 31);
            float fx;
            std::memcpy(&fx, &bb, 4);
            const double a = (double)fx - 0.25;
            const double ref = a / (double)n;
            const double q0 = a * y;
            const double e = std::fma(-q0, (double)n, a);
            const double q = std::fma(e, y, q0);
            if (std::memcmp(&ref, &q, 8)) dd++;
            const float h1 = (float)(ref - 0.5), h2 = (float)(q - 0.5);
            if (std::memcmp(&h1, &h2, 4)) df++;
        }
    }
    *float_diff = df;
    *double_diff = dd;
}

void orc_quantize(const float *frame, uint64_t npix, unsigned char *rgba8, float *depth) {
    for (uint64_t p = 0; p < npix; ++p) {
        rgba8[p * 4 + 0] = static_cast<unsigned char>(std::floor(frame[p * 5 + 0] * 255));
        rgba8[p * 4 + 1] = static_cast<unsigned char>(std::floor(frame[p * 5 + 1] * 255));
        rgba8[p * 4 + 2] = static_cast<unsigned char>(std::floor(frame[p * 5 + 2] * 255));
        rgba8[p * 4 + 3] = static_cast<unsigned char>(std::floor(frame[p * 5 + 3] * 255));
        if (depth) depth[p] = frame[p * 5 + 4];
    }
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_render(const orc_scene *sc, const vmx_camera *cam, const vmx_opts *opts, int rng_mode,
                int threads, float *out_rgbaz, vmx_stats *stats) {
    if (rng_mode == ORC_RNG_MT19937_64) {
        const uint64_t seed = opts->seed;
        /* one engine per thread, seeded once and carried across pixels,
         * pathtracer.cpp:231 (the reference seeds from std::random_device;
         * here seed + thread id) */
        render_rows<MtRng>(*sc, *cam, *opts, out_rgbaz, stats, threads,
                           [seed](MtRng &r, uint64_t, uint32_t) {
                               if (r.seeded) return;
#ifdef _OPENMP
                               r.eng.seed(seed + (uint64_t)omp_get_thread_num());
#else
                               r.eng.seed(seed);
#endif
                               r.seeded = true;
                           });
    } else {
        const uint64_t seed = opts->seed;
        render_rows<Xoshiro>(*sc, *cam, *opts, out_rgbaz, stats, threads,
                             [seed](Xoshiro &r, uint64_t p, uint32_t k) { r.init(seed, (uint32_t)p, k); });
    }
}

/* BruteForceTracer::Render (integrators.cpp:9-186), keyed jitter streams; W*H*5 floats
 * (r, g, b, alpha = hit fraction, depth = last sample's hit distance) */
void orc_render_bruteforce(const orc_scene *sc, const vmx_camera *cam, const vmx_opts *opts, uint32_t flags,
                           int threads, float *out_rgbaz, vmx_stats *stats) {
    render_bruteforce(*sc, *cam, *opts, flags, out_rgbaz, stats, threads);
}

} /* extern "C" */
