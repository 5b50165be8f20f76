// vmx_api.cpp — implementation of the C ABI in include/vermilion_hip.h.
//
// Host orchestration only: BVH build (bvh_build.cpp), uploads, the pass loop
// of the wavefront pipeline, statistics.  All arithmetic that defines results
// runs in the gfx950 kernels (vmx_kernels.hip); there is no CPU rendering
// path here — without a HIP device every compute entry point fails.
#include "../../include/vermilion_hip.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bvh_build.h"
#include "vmx_device.h"
#include "vmx_kernels.h"

#ifndef VMX_LDS_PRIMARY
#define VMX_LDS_PRIMARY 8  // LDS stack levels of the camera-ray kernel (A/B builds: make EXTRA=-DVMX_LDS_PRIMARY=n)
#endif

using namespace vmx;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(VMX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

#define LAUNCH_TRY(expr)                                                                           \
    do {                                                                                           \
        int e_ = (expr);                                                                           \
        if (e_ != 0)                                                                               \
            return fail(VMX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString((hipError_t)e_)); \
    } while (0)

// The reference's eight spheres, core/engines/meshEngine.cpp:377-500.
// sizeOfSpheres = 1e7*5 is a double narrowed to float by glm::vec3 / the float
// `rad` parameter (meshEngine.cpp:182, 425).
const vmx_sphere kReferenceSpheres[8] = {
    {{15.f, 140.f, 25.f}, 3.5f, {0.f * 15.f, .5f * 15.f, 1.0f * 15.f}, VMX_SPHERE_EMIT, {-55.f, 350.f, -150.f}, -1.f},
    {{0.f, 3300.f, 1300.f}, 250.f, {1.0f * 15.2f, 1.0f * 15.2f, 1.0f * 15.2f}, VMX_SPHERE_EMIT, {500.f, 800.f, 1300.f}, 1.f},
    {{0.f, (float)(-5e7), 0.f}, (float)5e7, {0, 0, 0}, 0u, {0.f, (float)(-5e7), 0.f}, 1.f},
    {{0.f, (float)(5e7 + 1000), 0.f}, (float)5e7, {0, 0, 0}, 0u, {0.f, (float)(5e7 + 1000), 0.f}, 1.f},
    {{(float)(-5e7 + 2000), 0.f, 0.f}, (float)5e7, {0, 0, 0}, 0u, {(float)(-5e7 + 2000), 0.f, 0.f}, -1.f},
    {{(float)(5e7 - 2000), 0.f, 0.f}, (float)5e7, {0, 0, 0}, 0u, {(float)(5e7 - 2000), 0.f, 0.f}, -1.f},
    {{0.f, 0.f, (float)(-5e7 + 2000)}, (float)5e7, {0, 0, 0}, 0u, {0.f, 0.f, (float)(-5e7 + 2000)}, -1.f},
    {{0.f, 0.f, (float)(5e7 - 2000)}, (float)5e7, {0, 0, 0}, 0u, {0.f, 0.f, (float)(5e7 - 2000)}, 1.f},
};

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int ensure(size_t count) {
        if (count <= n && p) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) return (int)e;
        n = count;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

struct EventPool {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    hipEvent_t get() {
        if (used == ev.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            ev.push_back(e);
        }
        return ev[used++];
    }
    void reset() { used = 0; }
    void release() {
        for (auto e : ev) (void)hipEventDestroy(e);
        ev.clear();
        used = 0;
    }
};

// per-scene reusable device workspace for the render pipeline
struct Workspace {
    DevBuf<unsigned char> queue_planes[2];  // first-generation kernels (A/B library) only
    DevBuf<unsigned int> queue_counts;      // 2 * kSubQueues * 32
    DevBuf<unsigned int> heads;         // kSubQueues * 32 reservation heads of k_paths
    DevBuf<unsigned char> overflow_stack;  // k_paths: stack levels beyond the LDS part
    DevBuf<unsigned char> rayA, state, hit, thr;  // split wavefront: per-path state
    DevBuf<unsigned int> ids[3], id_counts;             // split wavefront: live path ids ([2]: two-phase shading)
    DevBuf<unsigned int> sort_keys[2], ids_sorted;      // bounce reordering (path_sort.hip)
    DevBuf<unsigned char> sort_tmp;
    DevBuf<unsigned char> cam_inner;                    // per-frame camera-relative scene tables: 8 node copies, then the triangles
    DevBuf<unsigned char> rad;          // float4 per path of a pass
    DevBuf<unsigned long long> rad_mask;  // split pipeline: one bit per path, "its radiance was stored" (PathArrays::rad_mask)
    // VMX_SAMPLING_ELIDE_DEAD: live bits per 64 paths; [popcounts | their exclusive scan | list length]; the list; scan scratch
    DevBuf<unsigned long long> live_mask;
    DevBuf<unsigned int> live_u32, live_ids;
    DevBuf<unsigned char> live_tmp;
    // two-phase shading (k_shade_ends -> k_shade): the same three for the positions left to k_shade; their list is ids[2]
    DevBuf<unsigned long long> full_mask;
    DevBuf<unsigned int> full_u32;
    DevBuf<unsigned char> full_tmp;
    size_t full_words = 0, full_tmp_bytes = 0;
    DevBuf<unsigned char> out_rec;   // k_trace_w<.., SORT>: 32-byte records of the camera rays that still need shading
    DevBuf<unsigned int> out_count;
    size_t out_capacity = 0;         // entries out_rec was sized for
    DevBuf<unsigned char> accum;        // float4 per local pixel
    DevBuf<unsigned int> count, cursor, active[2], next_count;
    DevBuf<DevCounters> counters;
    DevBuf<float> out;  // frame buffer for host-output renders
    std::vector<unsigned int> order;
    uint32_t order_w = 0, order_rows = 0;
    EventPool events;
    void release() {
        queue_planes[0].release(), queue_planes[1].release(), queue_counts.release(), rad.release(), rad_mask.release(), heads.release(), overflow_stack.release();
        rayA.release(), state.release(), hit.release(), thr.release();
        ids[0].release(), ids[1].release(), ids[2].release(), id_counts.release(), cam_inner.release();
        sort_keys[0].release(), sort_keys[1].release(), ids_sorted.release(), sort_tmp.release();
        accum.release(), count.release(), cursor.release(), active[0].release(), active[1].release();
        next_count.release(), counters.release(), out.release(), events.release();
        live_mask.release(), live_u32.release(), live_ids.release(), live_tmp.release();
        full_mask.release(), full_u32.release(), full_tmp.release(), out_rec.release(), out_count.release();
    }
};

}  // namespace

struct vmx_scene {
    int device = 0;
    int num_cus = 0;
    HostBvh bvh;        // host-built trees: flat layout + device records; device-built (LBVH): filled on demand
    LbvhDevice lbvh;    // VMX_BVH_LBVH: the tree was built and flattened on the device (lbvh_build.hip)
    bool device_built = false;
    std::atomic<bool> flat_ready{true};
    uint32_t n_inner = 0;  // inner record slots on the device
    uint32_t ntris = 0, leaf_size = 4;
    std::vector<vmx_sphere> spheres;
    // inner records, then (64-byte aligned) the triangle records, in ONE allocation: a lane of the bounce
    // traversal kernel addresses either kind of record with a 32-bit byte offset from `dev.inner`
    // (SceneDev::tri_off), so inner-node lanes and leaf lanes of a wave fetch in one set of loads
    DevBuf<unsigned char> d_geom;
    DevBuf<AttrRecord> d_attrs;
    DevBuf<SphereDev> d_spheres;
    DevBuf<float> d_tex, d_tex1;
    uint32_t n_textures = 0;
    SceneDev dev{};
    hipStream_t stream = nullptr;
    std::mutex mu;
    Workspace ws;
    uint32_t block = 256;
    float bounds_lo[3] = {0, 0, 0}, bounds_hi[3] = {1, 1, 1};  // vertex bounds (origin cells of the bounce reordering)
    vmx_timings timings{};  // per-kernel durations of the last render on this scene
};

namespace {

uint32_t local_rows_of(uint32_t height, uint32_t stripe_rows, uint32_t rank, uint32_t world) {
    if (world <= 1) return height;
    uint32_t rows = 0;
    const uint32_t n_stripes = (height + stripe_rows - 1) / stripe_rows;
    for (uint32_t s = rank; s < n_stripes; s += world)
        rows += std::min(stripe_rows, height - s * stripe_rows);
    return rows;
}

// Camera ctor conversion (camera.cpp:43-47) + camera matrix (pathtracer.cpp:216-221).
// glm::rotate (gtc/matrix_transform) on the upper-left 3x3; column-major.
struct M3 {
    float c[3][3];
};
M3 rotate_axis(const M3 &m, float angle, float ax, float ay, float az) {
    const float c = std::cos(angle), s = std::sin(angle);
    const float inv = 1.0f / std::sqrt((ax * ax + ay * ay) + az * az);  // glm::normalize
    const float a[3] = {ax * inv, ay * inv, az * inv};
    const float t[3] = {a[0] * (1.0f - c), a[1] * (1.0f - c), a[2] * (1.0f - c)};
    float r[3][3];
    r[0][0] = c + t[0] * a[0];
    r[0][1] = t[0] * a[1] + s * a[2];
    r[0][2] = t[0] * a[2] - s * a[1];
    r[1][0] = t[1] * a[0] - s * a[2];
    r[1][1] = c + t[1] * a[1];
    r[1][2] = t[1] * a[2] + s * a[0];
    r[2][0] = t[2] * a[0] + s * a[1];
    r[2][1] = t[2] * a[1] - s * a[0];
    r[2][2] = c + t[2] * a[2];
    M3 out;
    for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row)
            out.c[col][row] = (m.c[0][row] * r[col][0] + m.c[1][row] * r[col][1]) + m.c[2][row] * r[col][2];
    return out;
}

int make_frame(const vmx_camera &cam, const vmx_opts &o, FrameDev &fr) {
    std::memset(&fr, 0, sizeof(fr));  // (fields a caller sets later — lead, bounce_bits — start defined: a fixed-count frame reads lead)
    const uint32_t W = cam.image_res[0], H = cam.image_res[1], spp = cam.rays_per_pixel;
    if (W == 0 || H == 0) return fail(VMX_ERR_INVALID, "image resolution must be non-zero");
    if ((uint64_t)W * H > 0x7fffffffull / 8) return fail(VMX_ERR_INVALID, "image too large");
    if (spp < 4)
        return fail(VMX_ERR_INVALID,
                    "rays_per_pixel < 4 renders no sample (uSamplesPerPixel/4 == 0, pathtracer.cpp:247)");
    if ((o.sampling & VMX_SAMPLING_MODE_MASK) > VMX_SAMPLING_CORRECTED || (o.sampling & ~(VMX_SAMPLING_MODE_MASK | VMX_SAMPLING_LIBM_DOUBLE | VMX_SAMPLING_ELIDE_DEAD)))
        return fail(VMX_ERR_INVALID, "unknown sampling mode");
    if (cam.rotation_units > VMX_ROTATION_RADIANS) return fail(VMX_ERR_INVALID, "unknown rotation_units");
    // Camera ctor (camera.cpp:43-47): mRotation = (-rx, -ry, +rz) * 3.1415926535 / 180, double arithmetic narrowed to
    // float; with VMX_ROTATION_RADIANS the caller hands over mRotation itself
    const bool rad = cam.rotation_units == VMX_ROTATION_RADIANS;
    const float rx = rad ? cam.rotation_rad[0] : (float)(-cam.rotation_deg[0] * 3.1415926535 / 180);
    const float ry = rad ? cam.rotation_rad[1] : (float)(-cam.rotation_deg[1] * 3.1415926535 / 180);
    const float rz = rad ? cam.rotation_rad[2] : (float)(cam.rotation_deg[2] * 3.1415926535 / 180);
    M3 m = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
    m = rotate_axis(m, ry, 0, 1, 0);
    m = rotate_axis(m, rx, 1, 0, 0);
    m = rotate_axis(m, rz, 0, 0, 1);
    for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row) fr.m[col * 3 + row] = m.c[col][row];
    fr.px = cam.position[0], fr.py = cam.position[1], fr.pz = cam.position[2];
    fr.film_dist = cam.back_distance;
    fr.sensor_x = cam.back_size[0], fr.sensor_y = cam.back_size[1];
    fr.width = W, fr.height = H;
    fr.inv_width = 1.0 / (double)W, fr.inv_height = 1.0 / (double)H;
    fr.div_width = make_fastdiv(W);
    fr.spp = spp, fr.quarter = spp / 4, fr.kmax = 4 * (spp / 4);
    fr.nmin = (uint32_t)std::floor(std::sqrt((double)spp));
    fr.early_stop = o.early_stop ? 1u : 0u;
    fr.r2scale = (o.sampling & VMX_SAMPLING_MODE_MASK) == VMX_SAMPLING_CORRECTED ? 1.0f : 10.0f;
    fr.libm_double = (o.sampling & VMX_SAMPLING_LIBM_DOUBLE) ? 1u : 0u;
    fr.elide_dead = (o.sampling & VMX_SAMPLING_ELIDE_DEAD) ? 1u : 0u;  // split passes of vmx_render only (k_raygen)
    fr.bounce_bits = 0;  // render_impl
    fr.world = o.world <= 1 ? 1u : o.world;
    fr.rank = o.world <= 1 ? 0u : o.rank;
    fr.stripe_rows = o.stripe_rows ? o.stripe_rows : 16u;
    fr.div_stripe = make_fastdiv(fr.stripe_rows);
    if (fr.rank >= fr.world) return fail(VMX_ERR_INVALID, "rank must be < world");
    fr.local_rows = local_rows_of(H, fr.stripe_rows, fr.rank, fr.world);
    fr.seed = o.seed;
    return VMX_OK;
}

int bind_device(const vmx_scene *sc) {
    HIP_TRY(hipSetDevice(sc->device));
    return VMX_OK;
}

LaunchCfg trace_cfg(const vmx_scene *sc, uint32_t work_items, int blocks_per_cu) {
    LaunchCfg c;
    c.block = sc->block;
    c.lds_bytes = (sc->block / 64) * sc->dev.stack_entries * 512;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    uint32_t grid = (uint32_t)sc->num_cus * (uint32_t)blocks_per_cu;
    grid = std::max(8u, grid & ~7u);
    if (work_items < grid) grid = std::max(1u, work_items);
    c.grid = grid;
    return c;
}

// 8x8-pixel tiles, tile-major: consecutive slots are neighbouring pixels, so a
// wave's 64 primary rays are coherent.
void tile_order(uint32_t W, uint32_t rows, std::vector<unsigned int> &order) {
    order.clear();
    order.reserve((size_t)W * rows);
    for (uint32_t ty = 0; ty < rows; ty += 8)
        for (uint32_t tx = 0; tx < W; tx += 8)
            for (uint32_t y = ty; y < std::min(ty + 8, rows); ++y)
                for (uint32_t x = tx; x < std::min(tx + 8, W); ++x) order.push_back(y * W + x);
}

void stage_out(vmx_stage_stats &dst, const StageCounters &c) {
    dst.rays = c.rays, dst.inner_visits = c.inner_visits, dst.tri_tests = c.tri_tests;
    dst.tri_hits = c.tri_hits, dst.continued = c.continued;
}

struct TimedLaunch {
    hipEvent_t a, b;
    int stage;   // vmx_stats bucket: 0 primary, 1 bounce, 2 shade
    int kernel;  // VMX_K_* of vmx_timings (per-kernel durations of the last call)
};

// reads the 16 sub-queue tails; returns total and the largest
int read_counts(vmx_scene *sc, unsigned int *d_counts, hipStream_t s, uint64_t &total, uint32_t &largest,
                uint32_t sub_capacity = 0xffffffffu, uint32_t *per_queue = nullptr) {
    unsigned int h[kSubQueues * 32];
    HIP_TRY(hipMemcpyAsync(h, d_counts, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    total = 0, largest = 0;
    for (uint32_t q = 0; q < kSubQueues; ++q) {
        // an append that would have written past its sub-list wrote nothing and left the tail high (id_append)
        if (h[q * 32] > sub_capacity) return fail(VMX_ERR_NOMEM, "live-path list overflow (sub-list " + std::to_string(q) + ")");
        const uint32_t c = h[q * 32];
        total += c;
        largest = std::max(largest, c);
        if (per_queue) per_queue[q] = c;
    }
    (void)sc;
    return VMX_OK;
}

#ifdef VMX_AB_KERNELS
int run_queue(vmx_scene *sc, const FrameDev &fr, QueueDev q[2], int cur, void *rad, DevCounters *ctr, bool count,
              uint32_t tail_threshold, hipStream_t s, std::vector<TimedLaunch> &timed, uint64_t &launches,
              int bounce_blocks) {
    Workspace &ws = sc->ws;
    for (;;) {
        uint64_t total;
        uint32_t largest;
        int rc = read_counts(sc, q[cur].counts, s, total, largest);
        if (rc) return rc;
        if (total == 0) break;
        const uint32_t max_chunks = (largest + sc->block - 1) / sc->block;
        const bool tail = total <= tail_threshold;
        LaunchCfg cfg = trace_cfg(sc, max_chunks * kSubQueues, bounce_blocks);
        if (!tail) LAUNCH_TRY(launch_zero_u32(q[cur ^ 1].counts, kSubQueues * 32, s));
        TimedLaunch tl{ws.events.get(), ws.events.get(), 1, VMX_K_OTHER};
        if (!tl.a || !tl.b) return fail(VMX_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(tl.a, s));
        LAUNCH_TRY(launch_bounce(sc->dev, fr.r2scale, fr.libm_double, q[cur], max_chunks, q[cur ^ 1], rad, ctr, count, tail, false,
                                 cfg, s));
        HIP_TRY(hipEventRecord(tl.b, s));
        timed.push_back(tl);
        launches += tail ? 1 : 2;
        if (tail) break;
        cur ^= 1;
    }
    return VMX_OK;
}

#endif

struct Tuning {
    uint32_t refill_min, refill_primary, shade_min, leaf_min, tail_threshold;
    uint32_t lds_entries, lds_primary, lds_bounce;  // LDS stack levels: fused kernels, camera-ray trace, bounce trace
    uint32_t sort_mode;  // bounce reordering: obits | dbits << 4 | dir_major << 8 | chunk_log2 << 12 | shade_sorted << 20
    bool two_phase;      // split passes of vmx_render: k_shade_ends + k_shade on what it queues (render_impl)
    bool sorted;         // ... and the camera rays sorted by the trace kernel itself (k_trace_w<0, .., SORT>): no k_shade_ends
    bool bounce_records; // one-phase shading of a render pass's bounce generations: the traversal kernel hands every ray on as a
                         // dense (t, leaf slot, path id) record (k_trace_w<1, .., SORT> with WorkDev::keep_all) instead of a
                         // scattered 8-byte hit[pid] store that k_shade<1> then gathers — same shading, every ray a full RayCast
    bool pool;           // A/B library, reserved[0] bit 10: the bounce generations of a pass through k_trace_pool (phase-pure
                         // steps, ray state in LDS; profiles/r04_state_pool.txt) — unsorted passes only
};

Tuning make_tuning(const vmx_scene *sc, const vmx_opts *o) {
    Tuning tn;
    // coherent camera rays do best when a wave starts 64 of them together; incoherent bounce rays
    // when finished lanes are replaced early: with the quad-cooperative record fetch an idle lane still
    // costs its share of every fetch, so lanes are refilled as soon as 8 are idle (measured 4/8: 47.7 ms,
    // 16: 49.4, 32: 55.5 for the bounce stage of the bench frame)
    tn.refill_min = o->reserved[3] ? o->reserved[3] : 8u;
    tn.refill_primary = o->reserved[3] ? o->reserved[3] : 64u;
    tn.shade_min = o->reserved[4] ? o->reserved[4] : 16u;
    tn.leaf_min = 0xFFFFFFFFu;
    tn.sort_mode = o->reserved[5];
    tn.two_phase = false;
    tn.sorted = false;
    tn.bounce_records = false;
    tn.pool = false;
    // LDS stack levels per lane (+1 scratch level), 512 B per level and wave.  Measured on the Sponza
    // stand-in: camera rays rarely go deep and gain from the 8th wave per SIMD that 8 levels leave
    // room for (52.6 -> 50.6 ms); the bounce kernel, once its record fetch is quad-cooperative, prefers
    // waves to LDS levels as well (8: 50.0, 9: 49.8, 10: 50.5, 12: 51.5, 13: 54.6 ms; with the per-lane
    // fetch it preferred 13 levels at 5-6 waves); the fused kernels keep 10 (7 waves).
    const uint32_t cap = o->reserved[6];
    tn.lds_entries = std::min(sc->dev.stack_entries, cap ? cap : 10u);
    tn.lds_primary = std::min(sc->dev.stack_entries, cap ? cap : (uint32_t)VMX_LDS_PRIMARY);
    tn.lds_bounce = std::min(sc->dev.stack_entries, cap ? cap : 9u);
    // bounce generations with fewer live paths than this finish in one fused launch (measured on the
    // Sponza stand-in: 512 K -> 16 M = 155.7 -> 152.9 ms fixed spp, 23.6 -> 20.8 ms with early stop; round 3, with the
    // traversal kernel sorting its rays: 16 M / 8 M / 4 M / 2 M = 86.0 / 85.8 / 85.6 / 85.8 ms, early stop 12.0 / 11.8 /
    // 12.0 / 11.8 — no difference; under VMX_SAMPLING_ELIDE_DEAD, where the first bounce generation is 16 M rays:
    // 32.3 / 31.7 / 31.6 / 32.3, so render_impl takes 8 M there)
    tn.tail_threshold = o->reserved[2] ? o->reserved[2] : (16u << 20);
    return tn;
}

constexpr uint32_t kPathsBlock = 256;

// fills the stack fields of a WorkDev and makes sure the global overflow slab is large enough
int bind_stack(vmx_scene *sc, const Tuning &tn, uint32_t entries, uint32_t grid, uint64_t items, WorkDev &wk) {
    // items a wave reserves from its work source per atomic: 256 when every lane will take >= 256 of them, else 128
    // (measured: early-stop frame 15.6 -> 14.3 ms with 128, whole 256-spp frame unchanged; 64 costs the camera-ray
    // kernel of the big frame 2 ms), 64 for launches of fewer than 64 items per lane: what a wave still holds privately
    // when the list runs empty is part of the launch's drain (round 4, tools/shard_kernels.py: the first bounce
    // generation of one rank of 8 — 9 M rays — 5.09 -> 4.94 ms; nothing measurable on larger launches)
    const uint64_t per_lane = items / ((uint64_t)grid * kPathsBlock);
    wk.reserve = per_lane >= 256 ? 256u : (per_lane >= 64 ? 128u : 64u);
    wk.lds_entries = entries;
    wk.leaf_min = tn.leaf_min;
    // + 1: k_trace_w keeps its bottom entry in LDS level 0
    wk.overflow_entries = sc->dev.stack_entries + 1 > entries ? sc->dev.stack_entries + 1 - entries : 1u;
    const size_t waves = (size_t)grid * (kPathsBlock / 64);
    if (sc->ws.overflow_stack.ensure(waves * wk.overflow_entries * 64 * 8))
        return fail(VMX_ERR_NOMEM, "hipMalloc failed for the overflow stack");
    wk.overflow_stack = sc->ws.overflow_stack.p;
    return VMX_OK;
}

LaunchCfg paths_cfg(const vmx_scene *sc, uint32_t entries, uint64_t items, int blocks_per_cu) {
    LaunchCfg c;
    c.block = kPathsBlock;
    c.lds_bytes = (kPathsBlock / 64) * (entries + 1) * 512;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    uint64_t grid = (uint64_t)sc->num_cus * (uint64_t)blocks_per_cu;
    const uint64_t need = (items + kPathsBlock - 1) / kPathsBlock;
    if (need < grid) grid = std::max<uint64_t>(1, need);
    c.grid = (uint32_t)grid;
    return c;
}

// ---- split wavefront (pipeline 0): k_trace_q + k_shade + id queues ---------------------------
int ensure_paths(vmx_scene *sc, size_t nslots, PathArrays &pa, IdQueue q[3]) {
    Workspace &ws = sc->ws;
    // k_shade appends the survivors of work item i to sub-list i % 16, at most 256 per item.  Its items cover the pass's
    // path slots — or, when the traversal kernels hand their rays on as records (k_trace_w<.., SORT>), the padded record
    // list, which is up to 256 entries per wave of the launch longer (at most 8 blocks of 4 waves per CU): each sub-list
    // holds its share of that, so no append can reach past it by construction
    const size_t list_pad = (size_t)sc->num_cus * 8 * (kPathsBlock / 64) * 256;
    const uint32_t sub_cap = (uint32_t)((nslots + list_pad) / kSubQueues + 1024);
    if (ws.rayA.ensure(nslots * 16) || ws.state.ensure(nslots * 64) || ws.hit.ensure(nslots * 8) ||
        ws.rad.ensure(nslots * 16) ||
        ws.ids[0].ensure((size_t)sub_cap * kSubQueues) || ws.ids[1].ensure((size_t)sub_cap * kSubQueues) ||
        ws.ids[2].ensure((size_t)sub_cap * kSubQueues) || ws.id_counts.ensure(3 * kSubQueues * 32) || ws.heads.ensure(kSubQueues * 32) || ws.counters.ensure(1))
        return fail(VMX_ERR_NOMEM, "hipMalloc failed for the path arrays");
    pa.rayA = ws.rayA.p, pa.state = ws.state.p;
    pa.hit = ws.hit.p, pa.rad = ws.rad.p;
    pa.thr = nullptr;
    pa.rad_mask = nullptr;  // render_impl switches it on for its split passes
    if (sc->dev.tex) {
        if (ws.thr.ensure(nslots * 16)) return fail(VMX_ERR_NOMEM, "hipMalloc failed for the throughput plane");
        pa.thr = ws.thr.p;
    }
    for (int i = 0; i < 3; ++i) {  // [0], [1]: the generations' ping-pong; [2]: k_shade_ends -> k_shade
        q[i].ids = ws.ids[i].p;
        q[i].counts = ws.id_counts.p + (size_t)i * kSubQueues * 32;
        q[i].sub_capacity = sub_cap;
        q[i].pad = 0;
    }
    return VMX_OK;
}

// two-phase shading of one generation: k_shade_ends finishes the steps that end by their draws alone and marks the other
// positions, launch_live_compact lists them in order, k_shade takes the list (wk.flat_ids) in dense waves.
// nwords: words of 64 positions k_shade_ends covers; clear_first: it does not write all of them (live-path list)
int shade_two_phase(vmx_scene *sc, const FrameDev &fr, WorkDev wk, PixelStateDev px, PathArrays pa, IdQueue q[3], IdQueue qout,
                    uint32_t max_chunks, size_t nwords, bool clear_first, DevCounters *ctr, bool from_queue, hipStream_t s) {
    Workspace &ws = sc->ws;
    if (nwords > ws.full_words) return fail(VMX_ERR_INVALID, "two-phase shading: more positions than the workspace was sized for");
    unsigned int *cnt = ws.full_u32.p, *offs = cnt + ws.full_words, *len = offs + ws.full_words;
    if (clear_first) {
        HIP_TRY(hipMemsetAsync(cnt, 0, nwords * 4, s));
        HIP_TRY(hipMemsetAsync(ws.full_mask.p, 0, nwords * 8, s));
    }
    LAUNCH_TRY(launch_shade_ends(sc->dev, fr, wk, pa, ws.full_mask.p, cnt, max_chunks, ctr, from_queue, s));
    HIP_TRY((hipError_t)launch_live_compact(ws.full_mask.p, cnt, (uint32_t)nwords, offs, q[2].ids, len, ws.full_tmp.p,
                                            ws.full_tmp_bytes, s));
    wk.flat_ids = q[2].ids, wk.flat_count = len;
    LAUNCH_TRY(launch_shade(sc->dev, fr, wk, px, pa, qout, max_chunks, ctr, from_queue, s));
    return VMX_OK;
}

// bounce generations of the split wavefront: trace + shade per generation while many paths are
// alive, then one launch that follows the remaining paths to their end
int run_ids(vmx_scene *sc, const FrameDev &fr, PathArrays pa, IdQueue q[3], int cur, DevCounters *ctr, bool count,
            const Tuning &tn, hipStream_t s, std::vector<TimedLaunch> &timed, uint64_t &launches, int trace_blocks,
            int tail_blocks) {
    Workspace &ws = sc->ws;
    PixelStateDev nopx{nullptr, nullptr, nullptr};
    for (;;) {
        uint64_t total;
        uint32_t largest;
        uint32_t per_queue[kSubQueues];
        int rc = read_counts(sc, q[cur].counts, s, total, largest, q[cur].sub_capacity, per_queue);
        if (rc) return rc;
        if (total == 0) break;
        const bool tail = total <= tn.tail_threshold;
        WorkDev wk;
        std::memset(&wk, 0, sizeof(wk));
        wk.heads = ws.heads.p;
        wk.nsrc = kSubQueues;
        wk.refill_min = tn.refill_min, wk.shade_min = tn.shade_min;
        wk.qids = q[cur];
        IdQueue q_shade = q[cur];
#ifndef VMX_AB_KERNELS
        if (tn.sort_mode)
            return fail(VMX_ERR_INVALID, "bounce reordering (vmx_opts.reserved[5]) is an experiment of the A/B library (make ab): "
                                         "it never paid for its sort, profiles/r03_bounce_sort.txt");
#else
        if (!tail && tn.sort_mode) {
            // reorder the live ids by (origin cell, direction cell) of their next rays (path_sort.hip)
            const size_t qsize = (size_t)q[cur].sub_capacity * kSubQueues;
            const size_t tmp_bytes = path_sort_tmp_bytes(largest);
            if (ws.sort_keys[0].ensure(qsize) || ws.sort_keys[1].ensure(qsize) || ws.ids_sorted.ensure(qsize) ||
                ws.sort_tmp.ensure(tmp_bytes))
                return fail(VMX_ERR_NOMEM, "hipMalloc failed for the path sort");
            SortKeyCfg kc;
            for (int a = 0; a < 3; ++a) {
                kc.lo[a] = sc->bounds_lo[a];
                const float ext = sc->bounds_hi[a] - sc->bounds_lo[a];
                kc.inv[a] = ext > 0.f ? 1.0f / ext : 0.f;
            }
            kc.obits = tn.sort_mode & 15u, kc.dbits = (tn.sort_mode >> 4) & 15u;
            kc.dir_major = (tn.sort_mode >> 8) & 1u, kc.chunk_log2 = (tn.sort_mode >> 12) & 31u;
            if (kc.obits > 10 || kc.dbits > 8 || 3 * kc.obits + 2 * kc.dbits > 32)
                return fail(VMX_ERR_INVALID, "bounce reordering: key wider than 32 bits");
            TimedLaunch tso{ws.events.get(), ws.events.get(), -1, VMX_K_OTHER};
            if (!tso.a || !tso.b) return fail(VMX_ERR_HIP, "hipEventCreate failed");
            HIP_TRY(hipEventRecord(tso.a, s));
            LAUNCH_TRY(path_sort_ids(q[cur], per_queue, pa.state, kc, ws.sort_keys[0].p, ws.sort_keys[1].p, ws.ids_sorted.p,
                                     ws.sort_tmp.p, tmp_bytes, s));
            HIP_TRY(hipEventRecord(tso.b, s));
            timed.push_back(tso);
            wk.qids.ids = ws.ids_sorted.p;
            if ((tn.sort_mode >> 20) & 1u) q_shade.ids = ws.ids_sorted.p;
        }
#endif
        const uint32_t entries = tail ? tn.lds_entries : tn.lds_bounce;
        LaunchCfg cfg = paths_cfg(sc, entries, total, tail ? tail_blocks : trace_blocks);
        rc = bind_stack(sc, tn, entries, cfg.grid, total, wk);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(ws.heads.p, 0, kSubQueues * 32 * 4, s));
        TimedLaunch tl{ws.events.get(), ws.events.get(), 1, tail ? VMX_K_TAIL : VMX_K_TRACE_BOUNCE};
        if (!tl.a || !tl.b) return fail(VMX_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(tl.a, s));
        if (tail) {
            LAUNCH_TRY(launch_tail(sc->dev, fr, wk, pa, ctr, count, cfg, s));
            HIP_TRY(hipEventRecord(tl.b, s));
            timed.push_back(tl);
            launches += 2;
            break;
        }
#ifdef VMX_AB_KERNELS
        if (tn.pool) {
            // the probe: P ray slots per block of 256 threads (VMX_AB_POOL_SLOTS, default 512), L stack levels in LDS
            // (VMX_AB_POOL_LEVELS, default 8), as many blocks per CU as the LDS admits
            uint32_t P = 512, Lv = 8, lds = 0;
            if (const char *e = std::getenv("VMX_AB_POOL_SLOTS")) P = (uint32_t)std::atoi(e);
            if (const char *e = std::getenv("VMX_AB_POOL_LEVELS")) Lv = (uint32_t)std::atoi(e);
            if (P < 64 || P > 16384 || Lv < 1 || Lv > 64) return fail(VMX_ERR_INVALID, "k_trace_pool: bad VMX_AB_POOL_SLOTS / _LEVELS");
            int pbl = 0;
            HIP_TRY((hipError_t)query_trace_pool(kPathsBlock, P, Lv, &lds, &pbl));
            if (pbl < 1) return fail(VMX_ERR_INVALID, "k_trace_pool does not fit on a CU with these slots / levels");
            if (const char *e = std::getenv("VMX_AB_POOL_BLOCKS")) pbl = std::max(1, std::min(pbl, std::atoi(e)));
            LaunchCfg pc;
            pc.block = kPathsBlock, pc.lds_bytes = lds;
            pc.grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)sc->num_cus * pbl, (total + P - 1) / P));
            if (const char *e = std::getenv("VMX_AB_POOL_GRID")) pc.grid = (uint32_t)std::max(1, std::min((int)pc.grid, std::atoi(e)));
            wk.pool_slots = P, wk.lds_entries = Lv;
            wk.overflow_entries = sc->dev.stack_entries + 1 > Lv ? sc->dev.stack_entries + 1 - Lv : 1u;
            if (ws.overflow_stack.ensure((size_t)pc.grid * P * wk.overflow_entries * 8))
                return fail(VMX_ERR_NOMEM, "hipMalloc failed for the overflow stack");
            wk.overflow_stack = ws.overflow_stack.p;
            wk.reserve = 256;
            LAUNCH_TRY(launch_trace_pool(sc->dev, wk, pa, pc, s));
        } else
#endif
        {
        if (tn.sorted || tn.bounce_records) {  // the traversal kernel settles the rays whose step ends by its draws and hands the others on as records (bounce_records: hands all of them on)
            wk.out_rec = ws.out_rec.p, wk.out_count = ws.out_count.p, wk.out_ctr = ctr, wk.out_capacity = (uint32_t)ws.out_capacity;
            wk.keep_all = tn.sorted ? 0u : 1u;
            HIP_TRY(hipMemsetAsync(ws.out_count.p, 0, 4, s));
        }
        LAUNCH_TRY(launch_trace_q(sc->dev, fr, wk, nopx, pa, ctr, count, true, cfg, s));
        }
        HIP_TRY(hipEventRecord(tl.b, s));
        timed.push_back(tl);
        HIP_TRY(hipMemsetAsync(q[cur ^ 1].counts, 0, kSubQueues * 32 * 4, s));
        TimedLaunch ts{ws.events.get(), ws.events.get(), 2, VMX_K_SHADE_BOUNCE};
        if (!ts.a || !ts.b) return fail(VMX_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(ts.a, s));
        wk.qids = q_shade;
        const uint32_t max_chunks = (largest + 255) / 256;
        if (tn.sorted || tn.bounce_records) {
            LAUNCH_TRY(launch_shade(sc->dev, fr, wk, nopx, pa, q[cur ^ 1], max_chunks, ctr, true, s));
        } else if (tn.two_phase) {
            rc = shade_two_phase(sc, fr, wk, nopx, pa, q, q[cur ^ 1], max_chunks, (size_t)max_chunks * kSubQueues * 4, false, ctr, true, s);
            if (rc) return rc;
        } else {
            LAUNCH_TRY(launch_shade(sc->dev, fr, wk, nopx, pa, q[cur ^ 1], max_chunks, ctr, true, s));
        }
        HIP_TRY(hipEventRecord(ts.b, s));
        timed.push_back(ts);
        launches += 4;
        cur ^= 1;
    }
    return VMX_OK;
}

#ifdef VMX_AB_KERNELS
int ensure_queues(vmx_scene *sc, uint32_t sub_capacity, QueueDev q[2]) {
    Workspace &ws = sc->ws;
    const size_t cap = (size_t)sub_capacity * kSubQueues;
    for (int i = 0; i < 2; ++i) {
        int e = ws.queue_planes[i].ensure(cap * kPathBytes);
        if (e) return fail(VMX_ERR_NOMEM, std::string("path queue: ") + hipGetErrorString((hipError_t)e));
    }
    int e = ws.queue_counts.ensure(2 * kSubQueues * 32);
    if (e) return fail(VMX_ERR_NOMEM, "queue counters");
    if (ws.heads.ensure(kSubQueues * 32)) return fail(VMX_ERR_NOMEM, "work heads");
    for (int i = 0; i < 2; ++i) {
        q[i].planes = ws.queue_planes[i].p;
        q[i].counts = ws.queue_counts.p + (size_t)i * kSubQueues * 32;
        q[i].capacity = (uint32_t)cap;
        q[i].sub_capacity = sub_capacity;
    }
    return VMX_OK;
}

#endif

int finish_stats(vmx_scene *sc, hipStream_t s, std::vector<TimedLaunch> &timed, hipEvent_t ev0, hipEvent_t ev1,
                 vmx_stats *stats, uint64_t launches, uint64_t passes,
                 std::chrono::steady_clock::time_point t0) {
    Workspace &ws = sc->ws;
    HIP_TRY(hipStreamSynchronize(s));
    DevCounters h;
    HIP_TRY(hipMemcpy(&h, ws.counters.p, sizeof(h), hipMemcpyDeviceToHost));
    if (h.overflow) return fail(VMX_ERR_NOMEM, "path queue overflow (sub-queues full); lower max paths per pass");
    vmx_stats local_stats;
    if (!stats) stats = &local_stats;
    std::memset(stats, 0, sizeof(*stats));
    stage_out(stats->primary, h.stage[0]);
    stage_out(stats->bounce, h.stage[1]);
    stats->rays_primary = h.stage[0].rays;
    stats->rays_secondary = h.stage[1].rays;
    stats->samples = h.samples;
    stats->samples_discarded = h.discarded;
    stats->passes = passes;
    stats->kernel_launches = launches;
    std::memset(&sc->timings, 0, sizeof(sc->timings));
    for (auto &tl : timed) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, tl.a, tl.b));
        if (tl.stage >= 0) {
            vmx_stage_stats &st = tl.stage == 0 ? stats->primary : (tl.stage == 1 ? stats->bounce : stats->shade);
            st.ms += ms;
            st.launches++;
        }
        sc->timings.ms[tl.kernel] += ms;
        sc->timings.launches[tl.kernel]++;
        if (ms > sc->timings.longest_ms[tl.kernel]) sc->timings.longest_ms[tl.kernel] = ms;
    }
    float ms = 0.f;
    if (ev0 && ev1) HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    stats->ms_device = ms;
    stats->ms_total =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return VMX_OK;
}

int render_impl(vmx_scene *sc, const vmx_camera *cam, const vmx_opts *opts, float *d_out, hipStream_t s,
                vmx_stats *stats) {
    const auto t0 = std::chrono::steady_clock::now();
    FrameDev fr;
    int rc = make_frame(*cam, *opts, fr);
    if (rc) return rc;
    Workspace &ws = sc->ws;
    const uint32_t W = fr.width, rows = fr.local_rows;
    const uint32_t npix = W * rows;
    const bool count = opts->collect_counters != 0;
    if (count) fr.elide_dead = 0;  // the counting build traces every ray: its totals are the oracle's
    // pipeline forms (all produce the same frame, bit for bit):
    //   0 split wavefront: k_trace_q + k_shade + id compaction (default)
    //   1 k_paths: refilling lanes keep their path to its end (no queues)
    //   2 first-generation k_primary/k_bounce wavefront   3 k_primary following every path to the end
    //   4 split wavefront for every pass (form 0 hands passes of fewer than 4 M paths to form 1's
    //     kernel, which needs no per-generation host round trip)
    const uint32_t pipeline = opts->reserved[0] & 0xFFu;  // (bit 8: one-phase shading, see two_phase below)
    if (pipeline > 4 || (opts->reserved[0] & ~0x7FFu)) return fail(VMX_ERR_INVALID, "unknown pipeline form");
#ifdef VMX_AB_KERNELS
    if (sc->dev.tex && pipeline >= 2 && pipeline <= 3)
        return fail(VMX_ERR_INVALID, "the first-generation kernels (pipeline forms 2, 3) do not sample textures");
#else
    if (pipeline >= 2 && pipeline <= 3)
        return fail(VMX_ERR_INVALID, "pipeline forms 2 and 3 (first-generation kernels) are only in the A/B library (make ab)");
#endif
    const bool split_any = pipeline == 0 || pipeline == 4;
    const bool legacy = pipeline >= 2 && pipeline <= 3;
    // passes below this many path slots run in the fused kernel (no per-generation launches): 4 M (1, 16 and 64 M were
    // all slower on early-stop frames in round 1; a full 7.4 M-path pass takes 5.1 ms fused, 3.6 ms split).  The passes
    // that FOLLOW the first one of an early-stop frame are different: many slots, few of them valid (pixels that stopped
    // a stratum take 3 of up to 1024) — they run fused up to 32 M slots (round 3: 13.6 -> 12.9 ms before the pass merge)
    constexpr uint64_t kHybridPaths = 4ull << 20, kHybridLeftover = 32ull << 20;
    Tuning tn = make_tuning(sc, opts);
    // two-phase shading of the split passes (k_shade_ends, then k_shade on what it queues) unless reserved[0] bit 8 asks
    // for the one-phase form; the counting build keeps the one-phase form (an A/B of the two inside every test that
    // compares a counted with an uncounted call)
    // (r2 = U: hardly a step ends by its draws — only Russian roulette past depth 5 — so there the extra phase is pure
    // overhead: `corrected` 64-spp bench frame 1.80 s one-phase, 1.92 s two-phase)
    tn.two_phase = !count && !(opts->reserved[0] & 0x100u) && fr.r2scale == 10.0f;
    // camera rays: the traversal kernel settles the rays whose step ends by its draws (and that no light sphere can
    // colour) when they finish, and hands the others on as records; reserved[0] bit 9 keeps k_shade_ends for them
    tn.sorted = tn.two_phase && !(opts->reserved[0] & 0x200u);
#ifdef VMX_AB_KERNELS
    tn.pool = (opts->reserved[0] & 0x400u) != 0 && !tn.sorted && !count;
#else
    if (opts->reserved[0] & 0x400u)
        return fail(VMX_ERR_INVALID, "k_trace_pool (vmx_opts.reserved[0] bit 10) is a probe of the A/B library (make ab): "
                                     "profiles/r04_state_pool.txt");
#endif
    fr.bounce_bits = tn.sorted ? 1u : 0u;
    fr.camera_bits = tn.two_phase ? 1u : 0u;  // read by k_trace_w<0, .., SORT> / k_shade_ends<0> only
    if (!opts->reserved[2]) {
        // measured (tools/elide_probe.py, tools/shard_probe.py): the whole frame on one GPU does not care between 2 M and
        // 16 M (86.0 / 85.8 / 85.6 / 85.8 ms at 16 / 8 / 4 / 2 M); a rank of a sharded frame does — its first bounce
        // generation (9 M rays on one of 8 ranks) runs 0.5 ms faster split than fused (13.8 -> 13.3 ms) — and so does
        // VMX_SAMPLING_ELIDE_DEAD, whose first generation is 16 M rays (32.3 -> 31.7 ms)
        uint32_t thr = fr.elide_dead ? (8u << 20) : (16u << 20);
        thr /= std::min(fr.world, 4u);
        tn.tail_threshold = std::max(thr, 2u << 20);
    }
    if (npix == 0) {
        if (stats) std::memset(stats, 0, sizeof(*stats));
        return VMX_OK;
    }

    // pass sizing — paths in flight per pass: up to 640 M path slots = 67 GB of per-path state (288 GB of HBM per
    // GPU); large passes keep the small late-bounce launches few (16 M -> 640 M per pass: 1.8x on the
    // whole frame).  Fixed-spp passes are balanced: ceil(kmax / passes) samples each.
    // (the first-generation kernels move 96-byte records through two queues: 16 M paths there)
    // (path ids are 32-bit: never more than 2^31 slots per pass)
    const uint64_t max_paths = std::min<uint64_t>(
        opts->reserved[1] ? opts->reserved[1] : (legacy ? (16ull << 20) : (640ull << 20)), 1ull << 31);
    uint32_t smax = (uint32_t)std::max<uint64_t>(1, max_paths / npix);
    if (opts->samples_per_batch) smax = opts->samples_per_batch;
    smax = std::min(smax, fr.kmax);
    // ... and never more than the device can hold right now (a shared or partitioned GPU, a second scene): a pass
    // needs `per_path` bytes per path slot; the budget is the free memory plus what this scene's workspace already
    // holds for the same purpose, less a tenth.  Frames do not depend on the pass size, so smaller passes are free of
    // consequences other than time.  VMX_MEM_BUDGET_MB lowers the budget (tests, co-tenancy).
    uint64_t mem_budget = 0;
    const size_t n_pad_cap = ((size_t)npix + 63u) & ~(size_t)63u;
    if (!opts->samples_per_batch && !legacy) {
        const size_t per_path = 16 + 64 + 8 + 16 + 12 /* three id lists */ + (sc->dev.tex ? 16 : 0) + (tn.sort_mode ? 12 : 0) + (fr.elide_dead ? 5 : 0) + 32;
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const size_t held = ws.rayA.n + ws.state.n + ws.hit.n + ws.rad.n + ws.thr.n + (ws.ids[0].n + ws.ids[1].n + ws.ids[2].n) * 4 +
                            (ws.sort_keys[0].n + ws.sort_keys[1].n + ws.ids_sorted.n) * 4 + ws.out_rec.n + ws.live_ids.n * 4;
        mem_budget = (uint64_t)((free_b + held) / 10 * 9);
        if (const char *e = std::getenv("VMX_MEM_BUDGET_MB")) {
            const uint64_t cap = std::strtoull(e, nullptr, 10) << 20;
            if (cap) mem_budget = std::min(mem_budget, cap);
        }
        const uint64_t fixed = (uint64_t)npix * 48 + ((uint64_t)sc->n_inner * 512 + (uint64_t)sc->ntris * 64) + (64ull << 20);
        const uint64_t fit = mem_budget > fixed ? (mem_budget - fixed) / per_path : 0;
        const uint32_t s_fit = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, fit / n_pad_cap), fr.kmax);
        if (s_fit < smax) smax = s_fit;
    }
    // under early stop no group is larger than the first one: nmin + 1 samples plus the first samples of the following strata
    if (fr.early_stop) smax = std::min(smax, fr.nmin + 1 + (fr.quarter ? fr.kmax / fr.quarter - 1u : 0u));
    if (!fr.early_stop && !opts->samples_per_batch) {
        const uint32_t npass = (fr.kmax + smax - 1) / smax;
        smax = (fr.kmax + npass - 1) / npass;
    }
    const uint32_t smax_alloc = smax;  // buffers are sized for this many samples per pixel and pass

    int pb = 1, bb = 1;
    int rb = 1;
#ifdef VMX_AB_KERNELS
    HIP_TRY((hipError_t)query_blocks_per_cu(sc->block, (sc->block / 64) * sc->dev.stack_entries * 512, count, &pb, &bb));
#endif
    HIP_TRY((hipError_t)query_paths_blocks_per_cu(kPathsBlock, (kPathsBlock / 64) * (tn.lds_entries + 1) * 512, count, &rb));
    if (pb < 1 || bb < 1 || rb < 1) return fail(VMX_ERR_HIP, "kernel does not fit on a CU (LDS stack too deep?)");

    // buffers
    const uint32_t n_pad_max = (npix + 63u) & ~63u;
    const uint32_t tiles8_max = (((n_pad_max + sc->block - 1) / sc->block) + 7u) & ~7u;
    // static item->sub-queue map (pipeline 2): exact bound; dynamic refill (pipeline 0): 25 % slack
    // per sub-queue, overflow falls through to the next sub-queue and is reported if nothing fits
    uint32_t sub_cap = ((tiles8_max * smax) / kSubQueues + 2) * sc->block;
    sub_cap = sub_cap + sub_cap / 4 + 4096;
#ifdef VMX_AB_KERNELS
    QueueDev q[2];
#endif
    PathArrays pa{};
    IdQueue qi[3];
    int tb = 1, tbb = 1;  // blocks per CU of the trace kernel: camera rays, bounce rays
    if (split_any) {
        rc = ensure_paths(sc, (size_t)n_pad_max * smax, pa, qi);
        if (rc == VMX_ERR_NOMEM && smax > 1 && !opts->samples_per_batch)
            return fail(VMX_ERR_NOMEM, "hipMalloc failed for the path arrays although " + std::to_string(mem_budget >> 20) +
                                           " MB were reported free: set VMX_MEM_BUDGET_MB or vmx_opts.reserved[1] (paths per pass)");
        if (rc) return rc;
        HIP_TRY((hipError_t)query_trace_q_blocks_per_cu(kPathsBlock, (kPathsBlock / 64) * (tn.lds_primary + 1) * 512, count, false,
                                                        tn.sorted, fr.elide_dead != 0, &tb));
        HIP_TRY((hipError_t)query_trace_q_blocks_per_cu(kPathsBlock, (kPathsBlock / 64) * (tn.lds_bounce + 1) * 512, count, true,
                                                        tn.sorted || (!tn.two_phase && !count && !tn.pool && fr.r2scale == 10.0f), false, &tbb));
        if (tb < 1 || tbb < 1) return fail(VMX_ERR_HIP, "trace kernel does not fit on a CU");
#ifdef VMX_AB_KERNELS
        // A/B library only: cap the bounce kernel's blocks per CU (how much of its time is latency hiding:
        // profiles/r04_state_pool.txt)
        if (const char *e = std::getenv("VMX_AB_BOUNCE_BLOCKS")) tbb = std::max(1, std::min(tbb, std::atoi(e)));
#endif
    }
#ifdef VMX_AB_KERNELS
    else if (legacy) {
        rc = ensure_queues(sc, sub_cap, q);
        if (rc) return rc;
    }
#else
    (void)sub_cap, (void)pb, (void)bb;
#endif
    if (ws.heads.ensure(kSubQueues * 32)) return fail(VMX_ERR_NOMEM, "work heads");
    if (ws.rad_mask.ensure(((size_t)n_pad_max * smax + 63) / 64 + 8)) return fail(VMX_ERR_NOMEM, "hipMalloc failed for the radiance mask");
    if (split_any) pa.rad_mask = ws.rad_mask.p;
    const bool elide = split_any && fr.elide_dead;  // camera paths of the split passes: compacted live list
    const size_t live_words_max = ((size_t)n_pad_max * smax + 63) / 64;
    if (split_any && tn.two_phase) {
        const size_t words = live_words_max + 2048;  // camera passes: one word per 64 path ids; bounce generations: per 64 queue slots
        const size_t tmp = live_compact_tmp_bytes((uint32_t)words);
        if (ws.full_mask.ensure(words + 8) || ws.full_u32.ensure(2 * words + 16) || ws.full_tmp.ensure(tmp + 256))
            return fail(VMX_ERR_NOMEM, "hipMalloc failed for the two-phase shading lists");
        ws.full_words = words, ws.full_tmp_bytes = tmp;
    }
    // one-phase shading (no two_phase): the bounce generations still go through dense ray records, all of them kept
    // (reference sampling only: there one path in seven goes on, and the gathered 8-byte hit records were a quarter of
    // k_shade<1>'s traffic — 5.9 -> 5.2 ms on the bench frame.  Under `corrected` sampling nearly every path goes on, and
    // records arrive in the order the rays FINISH, which scatters the state reads and writes that the id queue's order
    // keeps nearly sequential: measured 243 -> 328 ms of shading per 64-spp frame, so that form keeps hit[pid])
    tn.bounce_records = split_any && !tn.two_phase && !count && !tn.pool && fr.r2scale == 10.0f;
    if (split_any && (tn.sorted || tn.bounce_records)) {
        // every path of a pass may need a record; each wave of the launch leaves at most the tail of one chunk of 256 unused
        // (camera-ray records are 32 bytes, bounce-ray records 16)
        const size_t waves = (size_t)sc->num_cus * (size_t)std::max(tb, tbb) * (kPathsBlock / 64);
        ws.out_capacity = (size_t)n_pad_max * smax + waves * 256 + 1024;
        if (ws.out_capacity > 0xFFFFFFFFull) return fail(VMX_ERR_INVALID, "pass too large for the sorted ray records");
        if (ws.out_rec.ensure(ws.out_capacity * (tn.sorted ? 32 : 16)) || ws.out_count.ensure(32))
            return fail(VMX_ERR_NOMEM, "hipMalloc failed for the ray records");
    }
    size_t live_tmp_bytes = 0;
    if (elide) {
        live_tmp_bytes = live_compact_tmp_bytes((uint32_t)live_words_max);
        if (ws.live_mask.ensure(live_words_max + 8) || ws.live_u32.ensure(2 * live_words_max + 16) ||
            ws.live_ids.ensure((size_t)n_pad_max * smax + 64) || ws.live_tmp.ensure(live_tmp_bytes + 256))
            return fail(VMX_ERR_NOMEM, "hipMalloc failed for the live-path list");
    }
    if (ws.rad.ensure((size_t)n_pad_max * smax * 16) || ws.accum.ensure((size_t)npix * 16) ||
        ws.count.ensure(npix) || ws.cursor.ensure(npix) || ws.active[0].ensure(npix) ||
        ws.active[1].ensure(npix) || ws.next_count.ensure(32) || ws.counters.ensure(1))
        return fail(VMX_ERR_NOMEM, "hipMalloc failed for the render workspace");
    if (ws.order_w != W || ws.order_rows != rows) {
        tile_order(W, rows, ws.order);
        ws.order_w = W, ws.order_rows = rows;
    }
    PixelStateDev px{ws.accum.p, ws.count.p, ws.cursor.p};

    ws.events.reset();
    std::vector<TimedLaunch> timed;
    uint64_t launches = 0, passes = 0;
    hipEvent_t ev0 = ws.events.get(), ev1 = ws.events.get();
    if (!ev0 || !ev1) return fail(VMX_ERR_HIP, "hipEventCreate failed");

    HIP_TRY(hipMemcpyAsync(ws.active[0].p, ws.order.data(), (size_t)npix * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(ev0, s));
    if (split_any) {
        // camera-relative copies of the node and triangle records for this frame's origin
        const size_t n_inner = std::max<size_t>(sc->n_inner, 1);
        // (one allocation, node copies first: the assembly loop of k_trace_w<0> addresses a triangle record by a
        // 32-bit offset from the node table's base)
        if (ws.cam_inner.ensure(n_inner * 64 * 8 + (size_t)sc->ntris * 64))
            return fail(VMX_ERR_NOMEM, "hipMalloc failed for the camera tables");
        LAUNCH_TRY(launch_camera_tables(sc->dev, sc->n_inner, fr.px, fr.py, fr.pz, ws.cam_inner.p,
                                        ws.cam_inner.p + n_inner * 64 * 8, s));
        launches++;
    }
    HIP_TRY(hipMemsetAsync(ws.counters.p, 0, sizeof(DevCounters), s));
    LAUNCH_TRY(launch_init_pixels(px, npix, s));
    launches++;

    uint32_t n_active = npix;
    int cur_list = 0;
    uint32_t n_uniform = 0;  // samples every active pixel has taken while no early stop was possible
    uint32_t k_fixed = 0;    // fixed-spp mode: samples issued so far
    uint64_t last_pass_pixels = 0, last_pass_breaks = 0;  // early-stop statistics of the previous pass
    const uint32_t tail_threshold = tn.tail_threshold;
    while (n_active > 0) {
        uint32_t S;
        if (fr.early_stop) {
            // no sample can trigger the early-stop rule before n = nmin+1 (pathtracer.cpp:292):
            // up to there whole groups of samples are issued without speculation, after that one at a time
            fr.lead = 0;
            if (n_uniform < fr.nmin + 1 && n_uniform < fr.kmax) {
                S = std::min({fr.nmin + 1 - n_uniform, smax, fr.kmax - n_uniform});
                n_uniform += S;
                // Most pixels stop on the first test (sample nmin) and then on the first sample of every
                // following stratum: when the whole group fits in one pass, those first samples ride
                // along in it (sample_index / k_resolve), which saves a pass (measured 20.0 -> 17.5 ms)
                const uint32_t strata_after = fr.quarter ? fr.kmax / fr.quarter - 1u : 0u;
                if (S == fr.nmin + 1 && S <= fr.quarter && strata_after > 0 && S + strata_after <= smax) {
                    fr.lead = S;
                    S += strata_after;
                }
            } else {
                // Past that point a pixel may stop after any sample.  While most pixels are still
                // active one sample per pass is issued (nothing speculative); once the active set is
                // small, many samples per pixel are issued at once and k_resolve discards what
                // follows an early stop (same frame, fewer launch-bound passes).  Pixels that keep
                // sampling almost never stop later, so the speculation is deep (up to 8 M paths).
                // Speculation only pays for pixels that keep sampling: it is used when fewer than
                // 1 in 8 of the active pixels stopped a stratum in the previous pass.
                constexpr uint64_t spec = 16ull << 20, cap = 1024;  // measured: 45 ms -> 31 ms vs (1 M, 16)
                // A pixel that has just stopped a stratum gets the first samples of all following
                // strata in one pass (sample_index in the kernels): 3 paths cover it to the end of
                // the frame if it keeps stopping, as most do (measured: 6 -> 4 passes, 26 -> 21 ms).
                S = std::max(1u, fr.kmax / std::max(fr.quarter, 1u) - 1u);
                // ... or when so few pixels are left that a pass of everything they can still take is small anyway (the
                // Sponza stand-in: 4,492 of 2 M pixels after the first pass — a pass of 3 paths each was an empty launch
                // chain of ~1 ms before the one that finished them; 13.8 -> 12.1 ms)
                if ((last_pass_pixels > 0 && last_pass_breaks * 8 < last_pass_pixels) || (uint64_t)n_active * 2 * cap <= spec)
                    S = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, spec / (2ull * n_active)), cap);
                S = (uint32_t)std::min<uint64_t>(S, ((uint64_t)n_pad_max * smax_alloc) / ((n_active + 63u) & ~63u));
                S = std::max(S, 1u);
            }
        } else {
            if (k_fixed >= fr.kmax) break;
            S = std::min(smax, fr.kmax - k_fixed);
            k_fixed += S;
        }
        const uint32_t n_pad = (n_active + 63u) & ~63u;
        const uint32_t tiles8 = (((n_pad + sc->block - 1) / sc->block) + 7u) & ~7u;
        // the form this pass runs in
        const bool split = pipeline == 4 || (pipeline == 0 && (uint64_t)n_pad * S >= ((fr.early_stop && passes > 0) ? kHybridLeftover : kHybridPaths));
        const bool refill = pipeline == 1 || (pipeline == 0 && !split);
        const bool mega = refill || pipeline == 3;
        (void)mega;
#ifdef VMX_AB_KERNELS
        if (!mega && !split) HIP_TRY(hipMemsetAsync(q[0].counts, 0, kSubQueues * 32 * 4, s));
#endif
        TimedLaunch tl{ws.events.get(), ws.events.get(), 0, VMX_K_OTHER};
        if (!tl.a || !tl.b) return fail(VMX_ERR_HIP, "hipEventCreate failed");
        if (split) {
            tl.kernel = VMX_K_TRACE_CAMERA;
            WorkDev wk;
            std::memset(&wk, 0, sizeof(wk));
            wk.heads = ws.heads.p;
            wk.nsrc = 8;
            wk.refill_min = tn.refill_primary, wk.shade_min = tn.shade_min;
            wk.active = ws.active[cur_list].p;
            wk.n_active = n_active, wk.n_pad = n_pad, wk.samples = S, wk.div_samples = make_fastdiv(S);
            wk.band_slots = (((n_pad + 7u) / 8u) + 63u) & ~63u;
            wk.band_items = wk.band_slots * S;
            wk.pixel_major = 1;
            wk.cam_inner = ws.cam_inner.p, wk.cam_tris = ws.cam_inner.p + std::max<size_t>(sc->n_inner, 1) * 64 * 8;
            wk.cam_n_inner = sc->n_inner;
            LaunchCfg cfg = paths_cfg(sc, tn.lds_primary, (uint64_t)n_pad * S, tb);
            rc = bind_stack(sc, tn, tn.lds_primary, cfg.grid, (uint64_t)n_pad * S, wk);
            if (rc) return rc;
            HIP_TRY(hipMemsetAsync(ws.heads.p, 0, kSubQueues * 32 * 4, s));
            TimedLaunch tg{ws.events.get(), ws.events.get(), -1, VMX_K_RAYGEN};
            if (!tg.a || !tg.b) return fail(VMX_ERR_HIP, "hipEventCreate failed");
            HIP_TRY(hipEventRecord(tg.a, s));
            if (elide) {
                // live bits -> ordered list of live path ids -> their rays, dense; trace and shade then work on the list
                const uint32_t nwords = (uint32_t)(((uint64_t)n_pad * S + 63) / 64);
                unsigned int *cnt = ws.live_u32.p, *offs = cnt + live_words_max, *len = offs + live_words_max;
                // (one kernel for live bits, list and rays — a wave compacting its pixel's live samples through LDS and
                // appending them to the list in chunks — was measured in round 4: the list then holds the pixels in the
                // order the waves finish, the traversal kernel's 8 image bands lose their XCDs' L2 locality, and
                // k_trace_w<0, LIVE> went 8.2 -> 15.9 ms (and the generation itself 4.2 -> 5.6 ms): the ORDERED list is worth its scan)
                wk.live_mask = ws.live_mask.p, wk.live_cnt = cnt;
                LAUNCH_TRY(launch_raygen(sc->dev, fr, wk, px, pa, s));
                HIP_TRY((hipError_t)launch_live_compact(ws.live_mask.p, cnt, nwords, offs, ws.live_ids.p, len, ws.live_tmp.p, live_tmp_bytes, s));
                wk.live_ids = ws.live_ids.p, wk.live_count = len;
                LAUNCH_TRY(launch_raygen_live(sc->dev, fr, wk, px, pa, s));
                HIP_TRY(hipMemsetAsync(pa.rad_mask, 0, (size_t)nwords * 8, s));
            } else {
                LAUNCH_TRY(launch_raygen(sc->dev, fr, wk, px, pa, s));
            }
            if (tn.sorted) {
                wk.out_rec = ws.out_rec.p, wk.out_count = ws.out_count.p, wk.out_ctr = ws.counters.p, wk.out_capacity = (uint32_t)ws.out_capacity;
                HIP_TRY(hipMemsetAsync(ws.out_count.p, 0, 4, s));
                if (!elide) HIP_TRY(hipMemsetAsync(pa.rad_mask, 0, (size_t)(((uint64_t)n_pad * S + 63) / 64) * 8, s));
            }
            HIP_TRY(hipEventRecord(tg.b, s));
            timed.push_back(tg);
            HIP_TRY(hipEventRecord(tl.a, s));
            LAUNCH_TRY(launch_trace_q(sc->dev, fr, wk, px, pa, ws.counters.p, count, false, cfg, s));
            HIP_TRY(hipEventRecord(tl.b, s));
            HIP_TRY(hipMemsetAsync(qi[0].counts, 0, kSubQueues * 32 * 4, s));
            TimedLaunch ts{ws.events.get(), ws.events.get(), 2, VMX_K_SHADE_CAMERA};
            if (!ts.a || !ts.b) return fail(VMX_ERR_HIP, "hipEventCreate failed");
            HIP_TRY(hipEventRecord(ts.a, s));
            // (under VMX_SAMPLING_ELIDE_DEAD the camera paths left are mostly those that go on: one phase — 0.56 against
            // 0.77 ms on the early-stop bench frame, no difference on the fixed-count one)
            if (tn.two_phase && !elide && !tn.sorted) {
                rc = shade_two_phase(sc, fr, wk, px, pa, qi, qi[0], 0, (size_t)(((uint64_t)n_pad * S + 63) / 64), false, ws.counters.p, false, s);
                if (rc) return rc;
            } else {
                LAUNCH_TRY(launch_shade(sc->dev, fr, wk, px, pa, qi[0], 0, ws.counters.p, false, s));
            }
            HIP_TRY(hipEventRecord(ts.b, s));
            timed.push_back(ts);
            launches += 2;
        } else if (refill) {
            tl.kernel = VMX_K_FUSED;
            WorkDev wk;
            std::memset(&wk, 0, sizeof(wk));
            wk.heads = ws.heads.p;
            wk.nsrc = 8;
            wk.refill_min = tn.refill_min, wk.shade_min = tn.shade_min;
            wk.active = ws.active[cur_list].p;
            wk.n_active = n_active, wk.n_pad = n_pad, wk.samples = S, wk.div_samples = make_fastdiv(S);
            wk.band_slots = (((n_pad + 7u) / 8u) + 63u) & ~63u;
            wk.band_items = wk.band_slots * S;
            wk.pixel_major = 1;
            LaunchCfg cfg = paths_cfg(sc, tn.lds_entries, (uint64_t)n_pad * S, rb);
            rc = bind_stack(sc, tn, tn.lds_entries, cfg.grid, (uint64_t)n_pad * S, wk);
            if (rc) return rc;
            HIP_TRY(hipMemsetAsync(ws.heads.p, 0, kSubQueues * 32 * 4, s));
            HIP_TRY(hipEventRecord(tl.a, s));
            LAUNCH_TRY(launch_paths(sc->dev, fr, wk, px, ws.rad.p, ws.counters.p, count, cfg, s));
            HIP_TRY(hipEventRecord(tl.b, s));
        }
#ifdef VMX_AB_KERNELS
        else {
            LaunchCfg cfg = trace_cfg(sc, tiles8 * S, pb);
            HIP_TRY(hipEventRecord(tl.a, s));
            LAUNCH_TRY(launch_primary(sc->dev, fr, ws.active[cur_list].p, n_active, S, px, q[0], ws.rad.p,
                                      ws.counters.p, count, mega, cfg, s));
            HIP_TRY(hipEventRecord(tl.b, s));
        }
#else
        (void)tiles8;
#endif
        timed.push_back(tl);
        launches += 2;
        if (split) {
            rc = run_ids(sc, fr, pa, qi, 0, ws.counters.p, count, tn, s, timed, launches, tbb, rb);
            if (rc) return rc;
        }
#ifdef VMX_AB_KERNELS
        else if (!mega) {
            rc = run_queue(sc, fr, q, 0, ws.rad.p, ws.counters.p, count, tail_threshold, s, timed, launches,
                           bb);
            if (rc) return rc;
        }
#else
        (void)tail_threshold;
#endif
        HIP_TRY(hipMemsetAsync(ws.next_count.p, 0, 8, s));
        TimedLaunch tr{ws.events.get(), ws.events.get(), -1, VMX_K_RESOLVE};
        if (!tr.a || !tr.b) return fail(VMX_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(tr.a, s));
        LAUNCH_TRY(launch_resolve(fr, ws.active[cur_list].p, n_active, S, split || refill, ws.rad.p, split ? pa.rad_mask : nullptr, px, ws.active[cur_list ^ 1].p,
                                  ws.next_count.p, d_out, ws.counters.p, s));
        HIP_TRY(hipEventRecord(tr.b, s));
        timed.push_back(tr);
        launches++;
        passes++;
        unsigned int h_next[2] = {0, 0};  // [0] pixels still active, [1] pixels that stopped a stratum early
        HIP_TRY(hipMemcpyAsync(h_next, ws.next_count.p, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        last_pass_pixels = n_active;
        last_pass_breaks = h_next[1];
        n_active = h_next[0];
        cur_list ^= 1;
    }
    HIP_TRY(hipEventRecord(ev1, s));
    return finish_stats(sc, s, timed, ev0, ev1, stats, launches, passes, t0);
}

}  // namespace

extern "C" {

int vmx_abi_version(void) { return VMX_ABI_VERSION; }

const char *vmx_last_error(void) { return g_err.c_str(); }

int vmx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const vmx_sphere *vmx_default_spheres(uint32_t *count) {
    if (count) *count = 8;
    return kReferenceSpheres;
}

int vmx_scene_create(const float *pos, const float *nrm, const float *uv, uint32_t ntris,
                     const vmx_sphere *spheres, uint32_t nspheres, uint32_t leaf_size, int device,
                     vmx_scene **out) {
    return vmx_scene_create_ex(pos, nrm, uv, ntris, spheres, nspheres, leaf_size, VMX_BVH_REFERENCE, device, out);
}

// device half of scene creation: uploads sc->bvh / sc->spheres to sc->device (used for the first scene
// and for the replicas of a multi-device scene, which share one host-side build)
static int scene_upload(vmx_scene *sc) {
    const int device = sc->device;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    sc->num_cus = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking));

    std::vector<SphereDev> sd(sc->spheres.size());
    for (size_t i = 0; i < sd.size(); ++i) {
        const vmx_sphere &s = sc->spheres[i];
        SphereDev &d = sd[i];
        std::memset(&d, 0, sizeof(d));
        d.cx = s.centre[0], d.cy = s.centre[1], d.cz = s.centre[2];
        d.rad = s.radius;
        d.rad2 = s.radius * s.radius;  // float product (meshEngine.cpp:188)
        d.colr = s.colour[0], d.colg = s.colour[1], d.colb = s.colour[2];
        d.ncx = s.normal_centre[0], d.ncy = s.normal_centre[1], d.ncz = s.normal_centre[2];
        d.nsign = s.normal_sign < 0.f ? -1.f : 1.f;
        d.flags = s.flags;
    }
    if (sc->d_spheres.ensure(std::max<size_t>(sd.size(), 1))) return fail(VMX_ERR_NOMEM, "hipMalloc failed for the scene");
    if (sd.size()) HIP_TRY(hipMemcpy(sc->d_spheres.p, sd.data(), sd.size() * sizeof(SphereDev), hipMemcpyHostToDevice));
    sc->dev.spheres = sc->d_spheres.p;
    sc->dev.nspheres = (uint32_t)sd.size();
    sc->dev.emit_prefix = 0;
    for (size_t i = 0; i < sd.size(); ++i)
        if (sd[i].flags & 1u) sc->dev.emit_prefix = (uint32_t)i + 1;
    sc->dev.ntris = sc->ntris;
    if (sc->device_built) {
        // records were written on the device (k_lbvh_emit_*): the scene takes the builder's buffers over
        const LbvhDevice &l = sc->lbvh;
        sc->dev.inner = l.geom;
        sc->dev.tris = (const unsigned char *)l.geom + l.tri_off;
        sc->dev.tri_off = l.tri_off;
        sc->dev.attrs = l.attrs;
        sc->dev.root_ref = l.root_ref;
        sc->dev.stack_entries = l.height + 2;
        sc->n_inner = l.n_inner;
    } else {
        const HostBvh &b = sc->bvh;
        const size_t inner_bytes = std::max<size_t>(b.inner.size(), 1) * sizeof(InnerRecord);
        const size_t tri_bytes = b.tris.size() * sizeof(TriRecord);
        // + 64: the quad-cooperative fetch reads 64 bytes from a 48-byte triangle record's start
        if (inner_bytes + tri_bytes + 64 > 0xFFFFFFFFull) return fail(VMX_ERR_INVALID, "scene too large for 32-bit record offsets");
        if (sc->d_geom.ensure(inner_bytes + tri_bytes + 64) || sc->d_attrs.ensure(b.attrs.size()))
            return fail(VMX_ERR_NOMEM, "hipMalloc failed for the scene");
        HIP_TRY(hipMemset(sc->d_geom.p, 0, inner_bytes + tri_bytes + 64));
        if (b.inner.size()) HIP_TRY(hipMemcpy(sc->d_geom.p, b.inner.data(), b.inner.size() * sizeof(InnerRecord), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(sc->d_geom.p + inner_bytes, b.tris.data(), tri_bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(sc->d_attrs.p, b.attrs.data(), b.attrs.size() * sizeof(AttrRecord), hipMemcpyHostToDevice));
        sc->dev.inner = sc->d_geom.p;
        sc->dev.tris = sc->d_geom.p + inner_bytes;
        sc->dev.tri_off = (uint32_t)inner_bytes;
        sc->dev.attrs = sc->d_attrs.p;
        sc->dev.root_ref = b.root_ref;
        sc->dev.stack_entries = b.max_depth + 2;
        sc->n_inner = (uint32_t)b.inner.size();
    }
    // LDS budget: shrink the block until one block's stacks fit in 64 KiB
    sc->block = 256;
    while (sc->block > 64 && (sc->block / 64) * sc->dev.stack_entries * 512 > 65536) sc->block /= 2;
    return VMX_OK;
}

int vmx_scene_create_ex(const float *pos, const float *nrm, const float *uv, uint32_t ntris,
                        const vmx_sphere *spheres, uint32_t nspheres, uint32_t leaf_size, uint32_t builder,
                        int device, vmx_scene **out) {
    if (!out) return fail(VMX_ERR_INVALID, "out is NULL");
    if (builder > VMX_BVH_PLOC) return fail(VMX_ERR_INVALID, "unknown BVH builder");
    *out = nullptr;
    if (!pos || !nrm || ntris == 0) return fail(VMX_ERR_INVALID, "scene needs positions, normals, ntris > 0");
    if (spheres == nullptr && nspheres != 0)
        return fail(VMX_ERR_INVALID, "spheres is NULL but nspheres > 0 (NULL,0 selects the reference table)");
    if (nspheres > kMaxSpheres) return fail(VMX_ERR_INVALID, "more than 16 spheres");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(VMX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(VMX_ERR_NO_DEVICE, "device ordinal out of range");

    vmx_scene *sc = new vmx_scene();
    sc->device = device;
    sc->ntris = ntris;
    sc->leaf_size = leaf_size ? leaf_size : 4;
    std::string err;
    bool built;
    if (builder == VMX_BVH_LBVH || builder == VMX_BVH_PLOC) {
        sc->device_built = true, sc->flat_ready.store(false);
        built = builder == VMX_BVH_PLOC ? build_bvh_ploc_device(pos, nrm, uv, ntris, sc->leaf_size, device, sc->lbvh, err)
                                        : build_bvh_lbvh_device(pos, nrm, uv, ntris, sc->leaf_size, device, sc->lbvh, err);
    } else {
        built = builder == VMX_BVH_SAH ? build_bvh_sah(pos, nrm, uv, ntris, sc->leaf_size, sc->bvh, err)
                                       : build_bvh(pos, nrm, uv, ntris, sc->leaf_size, sc->bvh, err);
    }
    if (!built) {
        lbvh_release(sc->lbvh);
        // the device builders report HIP failures as "LBVH builder: <call>: <hipGetErrorString>"
        int code = VMX_ERR_INVALID;
        if (err.find("deeper") != std::string::npos) code = VMX_ERR_DEPTH;
        else if (err.rfind("LBVH builder: hip", 0) == 0)
            code = (err.find("hipMalloc") != std::string::npos || err.find("out of memory") != std::string::npos) ? VMX_ERR_NOMEM : VMX_ERR_HIP;
        delete sc;
        return fail(code, err);
    }
    if (spheres)
        sc->spheres.assign(spheres, spheres + nspheres);
    else
        sc->spheres.assign(kReferenceSpheres, kReferenceSpheres + 8);
    for (int a = 0; a < 3; ++a) sc->bounds_lo[a] = sc->bounds_hi[a] = pos[a];
    for (size_t v = 0; v < (size_t)ntris * 3; ++v)
        for (int a = 0; a < 3; ++a) {
            sc->bounds_lo[a] = std::min(sc->bounds_lo[a], pos[v * 3 + a]);
            sc->bounds_hi[a] = std::max(sc->bounds_hi[a], pos[v * 3 + a]);
        }
    const int rc = scene_upload(sc);
    if (rc) {
        const std::string keep = g_err;
        vmx_scene_destroy(sc);
        return fail(rc, keep);
    }
    *out = sc;
    return VMX_OK;
}

int vmx_scene_destroy(vmx_scene *sc) {
    if (!sc) return VMX_OK;
    (void)hipSetDevice(sc->device);
    sc->ws.release();
    sc->d_geom.release(), sc->d_attrs.release(), sc->d_spheres.release();
    lbvh_release(sc->lbvh);
    sc->d_tex.release(), sc->d_tex1.release();
    if (sc->stream) (void)hipStreamDestroy(sc->stream);
    delete sc;
    return VMX_OK;
}

int vmx_scene_bind_texture(vmx_scene *sc, const float *data, uint32_t width, uint32_t height, uint32_t channels) {
    if (!sc || !data) return fail(VMX_ERR_INVALID, "NULL argument");
    if (width == 0 || height == 0 || channels == 0 || channels > 4 || width > 65535 || height > 65535)
        return fail(VMX_ERR_INVALID, "texture must be 1..65535 texels wide/high with 1..4 channels");
    std::lock_guard<std::mutex> lock(sc->mu);
    int rc = bind_device(sc);
    if (rc) return rc;
    if (sc->n_textures == 0) {  // only boundTextures[0] is sampled by PathTracer (pathtracer.cpp:65)
        const size_t n = (size_t)width * height * channels;
        if (sc->d_tex.ensure(n)) return fail(VMX_ERR_NOMEM, "hipMalloc failed for the texture");
        HIP_TRY(hipMemcpy(sc->d_tex.p, data, n * 4, hipMemcpyHostToDevice));
        sc->dev.tex = sc->d_tex.p;
        sc->dev.tex_w = width, sc->dev.tex_h = height, sc->dev.tex_c = channels;
    } else if (sc->n_textures == 1) {  // boundTextures[1]: BruteForceTracer's albedo (integrators.cpp:141-147)
        const size_t n = (size_t)width * height * channels;
        if (sc->d_tex1.ensure(n)) return fail(VMX_ERR_NOMEM, "hipMalloc failed for the texture");
        HIP_TRY(hipMemcpy(sc->d_tex1.p, data, n * 4, hipMemcpyHostToDevice));
        sc->dev.tex1 = sc->d_tex1.p;
        sc->dev.tex1_w = width, sc->dev.tex1_h = height, sc->dev.tex1_c = channels;
    }
    sc->n_textures++;
    return VMX_OK;
}

// device-built trees: the reference's flat layout is produced on the first request for it
static int ensure_flat(const vmx_scene *csc) {
    vmx_scene *sc = const_cast<vmx_scene *>(csc);
    if (sc->flat_ready.load(std::memory_order_acquire)) return VMX_OK;
    std::lock_guard<std::mutex> lock(sc->mu);
    if (sc->flat_ready.load(std::memory_order_relaxed)) return VMX_OK;
    std::string err;
    if (!lbvh_export_flat(sc->lbvh, sc->device, sc->bvh, err)) return fail(VMX_ERR_HIP, err);
    sc->flat_ready.store(true, std::memory_order_release);
    return VMX_OK;
}

int vmx_scene_describe(const vmx_scene *sc, vmx_scene_desc *out) {
    if (!sc || !out) return fail(VMX_ERR_INVALID, "NULL argument");
    if (int rc = ensure_flat(sc)) return rc;
    std::memset(out, 0, sizeof(*out));
    out->ntris = sc->ntris;
    out->nspheres = (uint32_t)sc->spheres.size();
    out->leaf_size = sc->leaf_size;
    out->n_nodes = (uint32_t)sc->bvh.start.size();
    out->n_leaves = sc->bvh.n_leaves;
    out->n_inner = sc->n_inner;
    out->max_depth = sc->bvh.max_depth;
    out->stack_entries = sc->dev.stack_entries;
    out->device_bytes = (size_t)sc->n_inner * sizeof(InnerRecord) + (size_t)sc->ntris * sizeof(TriRecord) +
                        (size_t)sc->ntris * sizeof(AttrRecord) + sc->spheres.size() * sizeof(SphereDev) +
                        (sc->device_built ? sc->lbvh.arena_bytes : 0);  // device-built trees keep their hierarchy arrays
    out->device = sc->device;
    return VMX_OK;
}

int vmx_scene_timings(const vmx_scene *sc, vmx_timings *out) {
    if (!sc || !out) return fail(VMX_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(const_cast<vmx_scene *>(sc)->mu);  // a render on another thread rewrites them
    *out = sc->timings;
    return VMX_OK;
}

int vmx_scene_bvh(const vmx_scene *sc, uint32_t *start, uint32_t *nprims, uint32_t *right_offset, float *bbox,
                  uint32_t *prim_order) {
    if (!sc) return fail(VMX_ERR_INVALID, "NULL scene");
    if (int rc = ensure_flat(sc)) return rc;
    const HostBvh &b = sc->bvh;
    const size_t n = b.start.size();
    if (start) std::memcpy(start, b.start.data(), n * 4);
    if (nprims) std::memcpy(nprims, b.nprims.data(), n * 4);
    if (right_offset) std::memcpy(right_offset, b.right_offset.data(), n * 4);
    if (bbox) std::memcpy(bbox, b.bbox.data(), n * 24);
    if (prim_order) std::memcpy(prim_order, b.prim_order.data(), b.prim_order.size() * 4);
    return VMX_OK;
}

int vmx_trace(const vmx_scene *csc, const float *origin, const float *dir, uint32_t n, int32_t *tri_id,
              float *t) {
    vmx_scene *sc = const_cast<vmx_scene *>(csc);
    if (!sc || !origin || !dir || !tri_id || !t) return fail(VMX_ERR_INVALID, "NULL argument");
    if (n == 0) return VMX_OK;
    std::lock_guard<std::mutex> lock(sc->mu);
    int rc = bind_device(sc);
    if (rc) return rc;
    DevBuf<float> d_o, d_d, d_t;
    DevBuf<int32_t> d_id;
    if (d_o.ensure((size_t)n * 3) || d_d.ensure((size_t)n * 3) || d_t.ensure(n) || d_id.ensure(n))
        return fail(VMX_ERR_NOMEM, "hipMalloc failed for the ray batch");
    hipStream_t s = sc->stream;
    auto cleanup = [&]() { d_o.release(), d_d.release(), d_t.release(), d_id.release(); };
    hipError_t e = hipMemcpyAsync(d_o.p, origin, (size_t)n * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_d.p, dir, (size_t)n * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        LaunchCfg cfg = trace_cfg(sc, (n + sc->block - 1) / sc->block, 4);
        e = (hipError_t)launch_trace(sc->dev, d_o.p, d_d.p, n, d_id.p, d_t.p, nullptr, false, cfg, s);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(tri_id, d_id.p, (size_t)n * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(t, d_t.p, (size_t)n * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    cleanup();
    if (e != hipSuccess) return fail(VMX_ERR_HIP, std::string("vmx_trace: ") + hipGetErrorString(e));
    return VMX_OK;
}

int vmx_raycast(const vmx_scene *csc, const float *origin, const float *dir, uint32_t n, vmx_rayhit *out) {
    vmx_scene *sc = const_cast<vmx_scene *>(csc);
    if (!sc || !origin || !dir || !out) return fail(VMX_ERR_INVALID, "NULL argument");
    if (n == 0) return VMX_OK;
    static_assert(sizeof(vmx_rayhit) == 64, "vmx_rayhit must be 64 bytes");
    std::lock_guard<std::mutex> lock(sc->mu);
    int rc = bind_device(sc);
    if (rc) return rc;
    DevBuf<float> d_o, d_d;
    DevBuf<vmx_rayhit> d_out;
    if (d_o.ensure((size_t)n * 3) || d_d.ensure((size_t)n * 3) || d_out.ensure(n))
        return fail(VMX_ERR_NOMEM, "hipMalloc failed for the ray batch");
    hipStream_t s = sc->stream;
    hipError_t e = hipMemcpyAsync(d_o.p, origin, (size_t)n * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_d.p, dir, (size_t)n * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        LaunchCfg cfg = trace_cfg(sc, (n + sc->block - 1) / sc->block, 4);
        e = (hipError_t)launch_raycast(sc->dev, d_o.p, d_d.p, n, d_out.p, cfg, s);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out.p, (size_t)n * sizeof(vmx_rayhit), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    d_o.release(), d_d.release(), d_out.release();
    if (e != hipSuccess) return fail(VMX_ERR_HIP, std::string("vmx_raycast: ") + hipGetErrorString(e));
    return VMX_OK;
}

int vmx_primary_ids(const vmx_scene *csc, const vmx_camera *cam, const vmx_opts *opts, uint32_t k,
                    int32_t *tri_id, float *t) {
    vmx_scene *sc = const_cast<vmx_scene *>(csc);
    if (!sc || !cam || !opts || !tri_id || !t) return fail(VMX_ERR_INVALID, "NULL argument");
    FrameDev fr;
    int rc = make_frame(*cam, *opts, fr);
    if (rc) return rc;
    if (k >= fr.kmax) return fail(VMX_ERR_INVALID, "sample index out of range");
    std::lock_guard<std::mutex> lock(sc->mu);
    rc = bind_device(sc);
    if (rc) return rc;
    const uint32_t n = fr.width * fr.height;
    DevBuf<float> d_t;
    DevBuf<int32_t> d_id;
    if (d_t.ensure(n) || d_id.ensure(n)) return fail(VMX_ERR_NOMEM, "hipMalloc failed");
    hipStream_t s = sc->stream;
    LaunchCfg cfg = trace_cfg(sc, (n + sc->block - 1) / sc->block, 4);
    hipError_t e = (hipError_t)launch_primary_ids(sc->dev, fr, k, d_id.p, d_t.p, cfg, s);
    if (e == hipSuccess) e = hipMemcpyAsync(tri_id, d_id.p, (size_t)n * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(t, d_t.p, (size_t)n * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    d_t.release(), d_id.release();
    if (e != hipSuccess) return fail(VMX_ERR_HIP, std::string("vmx_primary_ids: ") + hipGetErrorString(e));
    return VMX_OK;
}

int vmx_radiance(const vmx_scene *csc, const float *origin, const float *dir, uint32_t n, const vmx_opts *opts,
                 float *out, vmx_stats *stats) {
    vmx_scene *sc = const_cast<vmx_scene *>(csc);
    if (!sc || !origin || !dir || !opts || !out) return fail(VMX_ERR_INVALID, "NULL argument");
    if ((opts->sampling & VMX_SAMPLING_MODE_MASK) > VMX_SAMPLING_CORRECTED ||
        (opts->sampling & ~(VMX_SAMPLING_MODE_MASK | VMX_SAMPLING_LIBM_DOUBLE | VMX_SAMPLING_ELIDE_DEAD)))
        return fail(VMX_ERR_INVALID, "unknown sampling mode");
    // (bits 8-10 of reserved[0] select shading / traversal forms of vmx_render's split passes and mean nothing here:
    // accepted and ignored, as render_impl accepts them)
    const uint32_t pipeline = opts->reserved[0] & 0xFFu;
    if (pipeline > 4 || (opts->reserved[0] & ~0x7FFu)) return fail(VMX_ERR_INVALID, "unknown pipeline form");
#ifdef VMX_AB_KERNELS
    if (sc->dev.tex && pipeline >= 2 && pipeline <= 3)
        return fail(VMX_ERR_INVALID, "the first-generation kernels (pipeline forms 2, 3) do not sample textures");
#else
    if (pipeline >= 2 && pipeline <= 3)
        return fail(VMX_ERR_INVALID, "pipeline forms 2 and 3 (first-generation kernels) are only in the A/B library (make ab)");
#endif
    if (n == 0) return VMX_OK;
    const auto t0 = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> lock(sc->mu);
    int rc = bind_device(sc);
    if (rc) return rc;
    Workspace &ws = sc->ws;
    hipStream_t s = sc->stream;
    const bool count = opts->collect_counters != 0;
    const bool legacy = pipeline == 2 || pipeline == 3;  // first-generation kernels
    (void)legacy;
    Tuning tn = make_tuning(sc, opts);
#ifdef VMX_AB_KERNELS
    tn.pool = (opts->reserved[0] & 0x400u) != 0 && !count;  // the phase-pure probe (vmx_trace_pool.inc)
#endif
    FrameDev fr;
    std::memset(&fr, 0, sizeof(fr));
    fr.r2scale = (opts->sampling & VMX_SAMPLING_MODE_MASK) == VMX_SAMPLING_CORRECTED ? 1.0f : 10.0f;
    fr.libm_double = (opts->sampling & VMX_SAMPLING_LIBM_DOUBLE) ? 1u : 0u;
    fr.elide_dead = 0;
    PathArrays pa{};
    IdQueue qi[3];
    int tb = 1, rb = 1;
    const uint32_t lds_paths = (kPathsBlock / 64) * (tn.lds_entries + 1) * 512;
#ifdef VMX_AB_KERNELS
    QueueDev q[2];
    int bb = 1, pb = 1;
    if (legacy) {
        const uint32_t blocks = (n + 255) / 256;
        const uint32_t sub_cap0 = (blocks / kSubQueues + 2) * 256;
        rc = ensure_queues(sc, sub_cap0 + sub_cap0 / 4 + 4096, q);
        if (rc) return rc;
        if (ws.rad.ensure((size_t)n * 16) || ws.counters.ensure(1)) return fail(VMX_ERR_NOMEM, "hipMalloc failed");
        HIP_TRY((hipError_t)query_blocks_per_cu(sc->block, (sc->block / 64) * sc->dev.stack_entries * 512, count, &pb, &bb));
    } else
#endif
    {
        rc = ensure_paths(sc, n, pa, qi);
        if (rc) return rc;
        HIP_TRY((hipError_t)query_trace_q_blocks_per_cu(kPathsBlock, (kPathsBlock / 64) * (tn.lds_bounce + 1) * 512, count, true,
                                                        false, false, &tb));
        HIP_TRY((hipError_t)query_paths_blocks_per_cu(kPathsBlock, lds_paths, count, &rb));
    }
    DevBuf<float> d_o, d_d;
    if (d_o.ensure((size_t)n * 3) || d_d.ensure((size_t)n * 3)) return fail(VMX_ERR_NOMEM, "hipMalloc failed");
    ws.events.reset();
    std::vector<TimedLaunch> timed;
    uint64_t launches = 0;
    hipEvent_t ev0 = ws.events.get(), ev1 = ws.events.get();
    auto body = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(d_o.p, origin, (size_t)n * 12, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_d.p, dir, (size_t)n * 12, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(ev0, s));
        HIP_TRY(hipMemsetAsync(ws.counters.p, 0, sizeof(DevCounters), s));
        int r;
#ifdef VMX_AB_KERNELS
        if (legacy) {
            HIP_TRY(hipMemsetAsync(q[0].counts, 0, kSubQueues * 32 * 4, s));
            LAUNCH_TRY(launch_radiance_init(d_o.p, d_d.p, n, opts->seed, q[0], s));
            launches += 2;
            r = run_queue(sc, fr, q, 0, ws.rad.p, ws.counters.p, count, tn.tail_threshold, s, timed, launches, bb);
        } else
#endif
        {
            HIP_TRY(hipMemsetAsync(qi[0].counts, 0, kSubQueues * 32 * 4, s));
            LAUNCH_TRY(launch_radiance_init_ids(d_o.p, d_d.p, n, opts->seed, pa, qi[0], s));
            launches += 2;
            r = run_ids(sc, fr, pa, qi, 0, ws.counters.p, count, tn, s, timed, launches, tb, rb);
        }
        if (r) return r;
        HIP_TRY(hipEventRecord(ev1, s));
        HIP_TRY(hipMemcpyAsync(out, ws.rad.p, (size_t)n * 16, hipMemcpyDeviceToHost, s));
        return finish_stats(sc, s, timed, ev0, ev1, stats, launches, 1, t0);
    };
    rc = body();
    d_o.release(), d_d.release();
    if (rc == VMX_OK && stats) stats->samples = n;
    return rc;
}

int vmx_trig(const float *x, uint32_t n, float *cos_out, float *sin_out, int device) {
    if (!x || !cos_out || !sin_out) return fail(VMX_ERR_INVALID, "NULL argument");
    if (n == 0) return VMX_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(VMX_ERR_NO_DEVICE, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    DevBuf<float> d_x, d_c, d_s;
    if (d_x.ensure(n) || d_c.ensure(n) || d_s.ensure(n)) return fail(VMX_ERR_NOMEM, "hipMalloc failed");
    hipError_t e = hipMemcpy(d_x.p, x, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = (hipError_t)launch_trig(d_x.p, n, d_c.p, d_s.p, nullptr);
    if (e == hipSuccess) e = hipMemcpy(cos_out, d_c.p, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(sin_out, d_s.p, (size_t)n * 4, hipMemcpyDeviceToHost);
    d_x.release(), d_c.release(), d_s.release();
    if (e != hipSuccess) return fail(VMX_ERR_HIP, std::string("vmx_trig: ") + hipGetErrorString(e));
    return VMX_OK;
}

int vmx_local_rows(uint32_t height, uint32_t stripe_rows, uint32_t rank, uint32_t world, uint32_t *rows) {
    if (!rows) return fail(VMX_ERR_INVALID, "rows is NULL");
    if (world > 1 && rank >= world) return fail(VMX_ERR_INVALID, "rank must be < world");
    *rows = local_rows_of(height, stripe_rows ? stripe_rows : 16u, rank, world);
    return VMX_OK;
}

int vmx_render_device(const vmx_scene *csc, const vmx_camera *cam, const vmx_opts *opts, void *d_out_rgbaz,
                      void *stream, vmx_stats *stats) {
    vmx_scene *sc = const_cast<vmx_scene *>(csc);
    if (!sc || !cam || !opts || !d_out_rgbaz) return fail(VMX_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(sc->mu);
    int rc = bind_device(sc);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : sc->stream;
    return render_impl(sc, cam, opts, (float *)d_out_rgbaz, s, stats);
}

int vmx_render(const vmx_scene *csc, const vmx_camera *cam, const vmx_opts *opts, float *out_rgbaz,
               vmx_stats *stats) {
    vmx_scene *sc = const_cast<vmx_scene *>(csc);
    if (!sc || !cam || !opts || !out_rgbaz) return fail(VMX_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(sc->mu);
    int rc = bind_device(sc);
    if (rc) return rc;
    FrameDev fr;
    rc = make_frame(*cam, *opts, fr);
    if (rc) return rc;
    const size_t nfloats = (size_t)fr.width * fr.local_rows * 5;
    if (nfloats == 0) return VMX_OK;
    if (sc->ws.out.ensure(nfloats)) return fail(VMX_ERR_NOMEM, "hipMalloc failed for the frame buffer");
    rc = render_impl(sc, cam, opts, sc->ws.out.p, sc->stream, stats);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out_rgbaz, sc->ws.out.p, nfloats * 4, hipMemcpyDeviceToHost));
    return VMX_OK;
}

int vmx_assemble_device(const void *d_gathered, uint64_t rank_stride_floats, uint32_t width, uint32_t height,
                        uint32_t stripe_rows, uint32_t world, void *d_frame, int device, void *stream) {
    if (!d_gathered || !d_frame || width == 0 || height == 0 || world == 0)
        return fail(VMX_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    LAUNCH_TRY(launch_assemble((const float *)d_gathered, rank_stride_floats, width, height,
                               stripe_rows ? stripe_rows : 16u, world, (float *)d_frame, stream));
    if (!stream) HIP_TRY(hipDeviceSynchronize());
    return VMX_OK;
}

int vmx_quantize_device(const void *d_frame_rgbaz, uint64_t npixels, void *d_rgba8, void *d_depth, int device,
                        void *stream) {
    if (!d_frame_rgbaz || !d_rgba8) return fail(VMX_ERR_INVALID, "NULL argument");
    if (npixels == 0) return VMX_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(VMX_ERR_NO_DEVICE, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    LAUNCH_TRY(launch_quantize((const float *)d_frame_rgbaz, npixels, d_rgba8, (float *)d_depth, stream));
    if (!stream) HIP_TRY(hipDeviceSynchronize());
    return VMX_OK;
}

} /* extern "C" */

namespace {

// BruteForceTracer::Render (integrators.cpp:9-186) on the device
int bruteforce_impl(vmx_scene *sc, const vmx_camera *cam, const vmx_opts *opts, uint32_t flags, float *d_out,
                    hipStream_t s, vmx_stats *stats) {
    const auto t0 = std::chrono::steady_clock::now();
    vmx_camera c = *cam;
    if (c.rays_per_pixel == 0 || c.rays_per_pixel > 65535)
        return fail(VMX_ERR_INVALID, "BruteForceTracer: rays_per_pixel must be 1..65535 (uint16_t sample counter, integrators.cpp:59)");
    if (flags & ~VMX_BF_ABS_INT) return fail(VMX_ERR_INVALID, "unknown BruteForceTracer flag");
    const uint32_t spp = c.rays_per_pixel;
    c.rays_per_pixel = std::max(spp, 4u);  // make_frame's PathTracer-only check (spp/4 strata) does not apply here
    vmx_opts o = *opts;
    o.sampling = VMX_SAMPLING_PARITY;
    FrameDev fr;
    int rc = make_frame(c, o, fr);
    if (rc) return rc;
    fr.spp = spp;
    Workspace &ws = sc->ws;
    const uint32_t W = fr.width, rows = fr.local_rows, npix = W * rows;
    if (npix == 0) {
        if (stats) std::memset(stats, 0, sizeof(*stats));
        return VMX_OK;
    }
    // rad doubles as the list of pixels that go on past the first samples (64 bytes each, at most every pixel)
    if (ws.active[0].ensure(npix) || ws.counters.ensure(1) || ws.rad.ensure((size_t)npix * 64) || ws.next_count.ensure(32))
        return fail(VMX_ERR_NOMEM, "hipMalloc failed for the render workspace");
    if (ws.order_w != W || ws.order_rows != rows) {
        tile_order(W, rows, ws.order);
        ws.order_w = W, ws.order_rows = rows;
    }
    ws.events.reset();
    std::vector<TimedLaunch> timed;
    hipEvent_t ev0 = ws.events.get(), ev1 = ws.events.get();
    if (!ev0 || !ev1) return fail(VMX_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipMemcpyAsync(ws.active[0].p, ws.order.data(), (size_t)npix * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(ws.counters.p, 0, sizeof(DevCounters), s));
    HIP_TRY(hipMemsetAsync(ws.next_count.p, 0, 4, s));
    HIP_TRY(hipEventRecord(ev0, s));
    // stack as in the traversal kernels: a few levels per lane in LDS, the rest in the global slab (the whole stack
    // in LDS leaves room for 3 waves per SIMD: 28 -> 20 ms for the 16-spp frame)
    const Tuning tn = make_tuning(sc, &o);
    LaunchCfg cfg = paths_cfg(sc, tn.lds_entries, (uint64_t)npix, 5);
    WorkDev stack;
    std::memset(&stack, 0, sizeof(stack));
    rc = bind_stack(sc, tn, tn.lds_entries, cfg.grid, (uint64_t)npix, stack);
    if (rc) return rc;
    // (one event pair per kernel: the host reads the long-pixel count between the two, and that round trip is not
    // device time)
    TimedLaunch tl{ws.events.get(), ws.events.get(), 0, VMX_K_BRUTEFORCE};
    if (!tl.a || !tl.b) return fail(VMX_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(tl.a, s));
    LAUNCH_TRY(launch_bruteforce(sc->dev, fr, ws.active[0].p, npix, flags, d_out, ws.counters.p, ws.rad.p, ws.next_count.p,
                                 stack, cfg, s));
    HIP_TRY(hipEventRecord(tl.b, s));
    timed.push_back(tl);
    uint64_t launches = 1;
    unsigned int n_long = 0;  // pixels the break has not stopped within the first samples: a wave each from here
    HIP_TRY(hipMemcpyAsync(&n_long, ws.next_count.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (n_long != 0) {
        TimedLaunch tl2{ws.events.get(), ws.events.get(), 0, VMX_K_BRUTEFORCE};
        if (!tl2.a || !tl2.b) return fail(VMX_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(tl2.a, s));
        LAUNCH_TRY(launch_bruteforce_long(sc->dev, fr, flags, d_out, ws.counters.p, ws.rad.p, ws.next_count.p, n_long, stack, cfg, s));
        HIP_TRY(hipEventRecord(tl2.b, s));
        timed.push_back(tl2);
        launches++;
    }
    HIP_TRY(hipEventRecord(ev1, s));
    return finish_stats(sc, s, timed, ev0, ev1, stats, 1, launches, t0);
}

}  // namespace

extern "C" {

int vmx_render_bruteforce_device(const vmx_scene *csc, const vmx_camera *cam, const vmx_opts *opts, uint32_t flags,
                                 void *d_out_rgbaz, void *stream, vmx_stats *stats) {
    vmx_scene *sc = const_cast<vmx_scene *>(csc);
    if (!sc || !cam || !opts || !d_out_rgbaz) return fail(VMX_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(sc->mu);
    int rc = bind_device(sc);
    if (rc) return rc;
    return bruteforce_impl(sc, cam, opts, flags, (float *)d_out_rgbaz, stream ? (hipStream_t)stream : sc->stream, stats);
}

int vmx_render_bruteforce(const vmx_scene *csc, const vmx_camera *cam, const vmx_opts *opts, uint32_t flags,
                          float *out_rgbaz, vmx_stats *stats) {
    vmx_scene *sc = const_cast<vmx_scene *>(csc);
    if (!sc || !cam || !opts || !out_rgbaz) return fail(VMX_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(sc->mu);
    int rc = bind_device(sc);
    if (rc) return rc;
    if (cam->image_res[0] == 0 || cam->image_res[1] == 0) return fail(VMX_ERR_INVALID, "image resolution must be non-zero");
    if (opts->world > 1 && opts->rank >= opts->world) return fail(VMX_ERR_INVALID, "rank must be < world");
    const uint32_t rows = local_rows_of(cam->image_res[1], opts->stripe_rows ? opts->stripe_rows : 16u,
                                        opts->world <= 1 ? 0u : opts->rank, opts->world <= 1 ? 1u : opts->world);
    const size_t nfloats = (size_t)cam->image_res[0] * rows * 5;
    if (nfloats == 0) return VMX_OK;
    if (sc->ws.out.ensure(nfloats)) return fail(VMX_ERR_NOMEM, "hipMalloc failed for the frame buffer");
    rc = bruteforce_impl(sc, cam, opts, flags, sc->ws.out.p, sc->stream, stats);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out_rgbaz, sc->ws.out.p, nfloats * 4, hipMemcpyDeviceToHost));
    return VMX_OK;
}

} /* extern "C" */

// ---------------------------------------------------------------------------
// multi-device rendering in ONE process (north_star: "tiles shard across the 8 GPUs of one node with a
// gather of per-tile framebuffers over xGMI"): a Vermilion main.cpp is a single process, so the
// sharding must be reachable from the C ABI, not only from one-process-per-GPU launchers.
// ---------------------------------------------------------------------------
// one persistent host thread per replica (a render hands each of them its job and waits for all of them)
struct ReplicaWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = false, quit = false;
    void start() {
        th = std::thread([this]() {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this]() { return has_job || quit; });
                if (quit) return;
                std::function<void()> j = std::move(job);
                has_job = false;
                lk.unlock();
                j();
                lk.lock();
                done = true;
                cv.notify_all();
            }
        });
    }
    void submit(std::function<void()> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j), has_job = true, done = false;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this]() { return done; });
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

struct vmx_multi {
    std::vector<vmx_scene *> replica;  // one scene replica per entry of the device list (entries may repeat)
    std::vector<std::unique_ptr<ReplicaWorker>> worker;
    // how replica r's stripes reach the root: 2 same device, 1 direct peer copy (xGMI), 0 staged through the host
    std::vector<int> route;
    DevBuf<float> gathered, frame;     // on the root = replica[0]'s device
    std::mutex mu;
    // the exchange step of the last render, timed apart from the rendering ("gather time separately"):
    // per replica the device time of its render and of its stripes' copy into the root's gather buffer (hipEvent pairs
    // on the replica's stream), the de-interleave kernel on the root, and the host's wall clock around all of it
    std::vector<double> render_ms, copy_ms;
    std::vector<hipEvent_t> copy_ev;   // two per replica, created on the replica's device by its worker
    hipEvent_t asm_ev[2] = {nullptr, nullptr};
    double assemble_ms = 0.0, wall_ms = 0.0;
};

namespace {

template <class RenderFn>
int multi_render(vmx_multi *m, const vmx_camera *cam, const vmx_opts *opts, float *out_host, void *d_out_root,
                 vmx_stats *stats, RenderFn render_one) {
    const uint32_t world = (uint32_t)m->replica.size();
    const uint32_t W = cam->image_res[0], H = cam->image_res[1];
    if (W == 0 || H == 0) return fail(VMX_ERR_INVALID, "image resolution must be non-zero");
    const uint32_t stripe = opts->stripe_rows ? opts->stripe_rows : 16u;
    uint32_t max_rows = 0;
    for (uint32_t r = 0; r < world; ++r) max_rows = std::max(max_rows, local_rows_of(H, stripe, r, world));
    const uint64_t stride = (uint64_t)max_rows * W * 5;  // floats per rank slot of the gather buffer
    vmx_scene *root = m->replica[0];
    HIP_TRY(hipSetDevice(root->device));
    if (m->gathered.ensure((size_t)stride * world) || m->frame.ensure((size_t)W * H * 5))
        return fail(VMX_ERR_NOMEM, "hipMalloc failed for the gather buffer");

    // one host thread per replica: render its interleaved stripes into its own device buffer, then push
    // them into the root's gather buffer — a device-to-device copy (peer copy over xGMI when the replica
    // sits on another GPU: every peer has its own link to the root; not a ring)
    std::vector<int> rc(world, VMX_OK);
    std::vector<std::string> msg(world);
    std::vector<vmx_stats> st(world);
    const auto wall0 = std::chrono::steady_clock::now();
    m->render_ms.assign(world, 0.0), m->copy_ms.assign(world, 0.0);
    m->copy_ev.resize((size_t)world * 2, nullptr);
    for (uint32_t r = 0; r < world; ++r) {
        m->worker[r]->submit([&, r]() {
            vmx_scene *sc = m->replica[r];
            vmx_opts o = *opts;
            o.rank = r, o.world = world, o.stripe_rows = stripe;
            std::lock_guard<std::mutex> lock(sc->mu);
            auto body = [&]() -> int {
                int e = bind_device(sc);
                if (e) return e;
                const size_t nfloats = (size_t)local_rows_of(H, stripe, r, world) * W * 5;
                if (nfloats == 0) return VMX_OK;
                if (sc->ws.out.ensure(nfloats)) return fail(VMX_ERR_NOMEM, "hipMalloc failed for the frame buffer");
                e = render_one(sc, &o, sc->ws.out.p, &st[r]);
                if (e) return e;
                hipEvent_t *ev = &m->copy_ev[(size_t)r * 2];
                for (int k = 0; k < 2; ++k)
                    if (!ev[k]) HIP_TRY(hipEventCreate(&ev[k]));
                HIP_TRY(hipEventRecord(ev[0], sc->stream));
                HIP_TRY(hipMemcpyPeerAsync(m->gathered.p + (size_t)stride * r, root->device, sc->ws.out.p, sc->device,
                                           nfloats * 4, sc->stream));
                HIP_TRY(hipEventRecord(ev[1], sc->stream));
                HIP_TRY(hipStreamSynchronize(sc->stream));
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
                m->copy_ms[r] = ms, m->render_ms[r] = st[r].ms_device;
                return VMX_OK;
            };
            std::memset(&st[r], 0, sizeof(vmx_stats));
            rc[r] = body();
            if (rc[r]) msg[r] = g_err;  // g_err is thread-local
        });
    }
    for (uint32_t r = 0; r < world; ++r) m->worker[r]->wait();
    for (uint32_t r = 0; r < world; ++r)
        if (rc[r]) return fail(rc[r], "device " + std::to_string(m->replica[r]->device) + ": " + msg[r]);

    HIP_TRY(hipSetDevice(root->device));
    float *d_frame = d_out_root ? (float *)d_out_root : m->frame.p;
    for (int k = 0; k < 2; ++k)
        if (!m->asm_ev[k]) HIP_TRY(hipEventCreate(&m->asm_ev[k]));
    HIP_TRY(hipEventRecord(m->asm_ev[0], root->stream));
    LAUNCH_TRY(launch_assemble(m->gathered.p, stride, W, H, stripe, world, d_frame, root->stream));
    HIP_TRY(hipEventRecord(m->asm_ev[1], root->stream));
    if (out_host) HIP_TRY(hipMemcpyAsync(out_host, d_frame, (size_t)W * H * 5 * 4, hipMemcpyDeviceToHost, root->stream));
    HIP_TRY(hipStreamSynchronize(root->stream));
    {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, m->asm_ev[0], m->asm_ev[1]));
        m->assemble_ms = ms;
        m->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (uint32_t r = 0; r < world; ++r) {
            const vmx_stats &a = st[r];
            stats->rays_primary += a.rays_primary, stats->rays_secondary += a.rays_secondary;
            stats->samples += a.samples, stats->samples_discarded += a.samples_discarded;
            stats->kernel_launches += a.kernel_launches;
            stats->passes = std::max(stats->passes, a.passes);
            stats->ms_total = std::max(stats->ms_total, a.ms_total);     // ranks run side by side: the slowest one
            stats->ms_device = std::max(stats->ms_device, a.ms_device);
            vmx_stage_stats *dst[3] = {&stats->primary, &stats->bounce, &stats->shade};
            const vmx_stage_stats *src[3] = {&a.primary, &a.bounce, &a.shade};
            for (int k = 0; k < 3; ++k) {
                dst[k]->rays += src[k]->rays, dst[k]->inner_visits += src[k]->inner_visits;
                dst[k]->tri_tests += src[k]->tri_tests, dst[k]->tri_hits += src[k]->tri_hits;
                dst[k]->continued += src[k]->continued, dst[k]->launches += src[k]->launches;
                dst[k]->ms = std::max(dst[k]->ms, src[k]->ms);
            }
        }
    }
    return VMX_OK;
}

}  // namespace

extern "C" {

int vmx_multi_create(const float *pos, const float *nrm, const float *uv, uint32_t ntris, const vmx_sphere *spheres,
                     uint32_t nspheres, uint32_t leaf_size, uint32_t builder, const int *devices, uint32_t ndevices,
                     vmx_multi **out) {
    if (!out) return fail(VMX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!devices || ndevices == 0 || ndevices > 64) return fail(VMX_ERR_INVALID, "device list must hold 1..64 entries");
    vmx_scene *first = nullptr;
    int rc = vmx_scene_create_ex(pos, nrm, uv, ntris, spheres, nspheres, leaf_size, builder, devices[0], &first);
    if (rc) return rc;
    vmx_multi *m = new vmx_multi();
    m->replica.push_back(first);
    m->route.push_back(2);
    int ndev = 0;
    (void)hipGetDeviceCount(&ndev);
    for (uint32_t i = 1; i < ndevices; ++i) {
        if (devices[i] < 0 || devices[i] >= ndev) {
            vmx_multi_destroy(m);
            return fail(VMX_ERR_NO_DEVICE, "device ordinal out of range");
        }
        vmx_scene *sc = nullptr;
        if (first->device_built) {
            // a device-built tree is built again on every device (deterministic: same sort, same boxes)
            rc = vmx_scene_create_ex(pos, nrm, uv, ntris, spheres, nspheres, leaf_size, builder, devices[i], &sc);
            if (rc) {
                const std::string keep = g_err;
                vmx_multi_destroy(m);
                return fail(rc, keep);
            }
        } else {
            sc = new vmx_scene();  // replica: shares the host-side build, uploads to its own device
            sc->device = devices[i];
            sc->ntris = first->ntris, sc->leaf_size = first->leaf_size;
            sc->bvh = first->bvh;
            sc->spheres = first->spheres;
            std::memcpy(sc->bounds_lo, first->bounds_lo, sizeof(sc->bounds_lo));
            std::memcpy(sc->bounds_hi, first->bounds_hi, sizeof(sc->bounds_hi));
            rc = scene_upload(sc);
            if (rc) {
                const std::string keep = g_err;
                vmx_scene_destroy(sc);
                vmx_multi_destroy(m);
                return fail(rc, keep);
            }
        }
        m->replica.push_back(sc);
        // direct peer copies into the root's gather buffer (xGMI); without peer access the runtime stages
        // the copy through the host, which is slower but still correct
        int route = 2;
        if (devices[i] != devices[0]) {
            route = 0;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can) {
                (void)hipSetDevice(devices[i]);
                const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
                // every non-success return (AlreadyEnabled included: a device listed twice, a second vmx_multi in the
                // process) stays behind as the thread's last error and would fail the next launch's status check
                if (e != hipSuccess) (void)hipGetLastError();
                if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) route = 1;
                else
                    std::fprintf(stderr, "vermilion_hip: peer access %d -> %d failed (%s): stripes of device %d are staged "
                                 "through the host\n", devices[i], devices[0], hipGetErrorString(e), devices[i]);
            } else {
                (void)hipGetLastError();
            }
        }
        m->route.push_back(route);
    }
    for (size_t i = 0; i < m->replica.size(); ++i) {
        m->worker.emplace_back(new ReplicaWorker());
        m->worker.back()->start();
    }
    *out = m;
    return VMX_OK;
}

int vmx_multi_destroy(vmx_multi *m) {
    if (!m) return VMX_OK;
    for (auto &w : m->worker) w->stop();
    for (size_t i = 0; i < m->copy_ev.size(); ++i)
        if (m->copy_ev[i]) {
            (void)hipSetDevice(m->replica[i / 2]->device);
            (void)hipEventDestroy(m->copy_ev[i]);
        }
    if (!m->replica.empty()) (void)hipSetDevice(m->replica[0]->device);
    for (int k = 0; k < 2; ++k)
        if (m->asm_ev[k]) (void)hipEventDestroy(m->asm_ev[k]);
    m->gathered.release(), m->frame.release();
    for (vmx_scene *sc : m->replica) vmx_scene_destroy(sc);
    delete m;
    return VMX_OK;
}

uint32_t vmx_multi_world(const vmx_multi *m) { return m ? (uint32_t)m->replica.size() : 0u; }

int vmx_multi_routes(const vmx_multi *m, int *devices, int *routes) {
    if (!m) return fail(VMX_ERR_INVALID, "NULL argument");
    for (size_t i = 0; i < m->replica.size(); ++i) {
        if (devices) devices[i] = m->replica[i]->device;
        if (routes) routes[i] = m->route[i];
    }
    return VMX_OK;
}

int vmx_multi_timings(const vmx_multi *cm, vmx_multi_times *out, double *render_ms, double *copy_ms) {
    vmx_multi *m = const_cast<vmx_multi *>(cm);
    if (!m || !out) return fail(VMX_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(m->mu);
    std::memset(out, 0, sizeof(*out));
    out->world = (uint32_t)m->replica.size();
    for (size_t r = 0; r < m->render_ms.size(); ++r) {
        out->slowest_render_ms = std::max(out->slowest_render_ms, m->render_ms[r]);
        out->gather_ms = std::max(out->gather_ms, m->copy_ms[r]);
        out->gather_sum_ms += m->copy_ms[r];
        if (render_ms) render_ms[r] = m->render_ms[r];
        if (copy_ms) copy_ms[r] = m->copy_ms[r];
    }
    out->assemble_ms = m->assemble_ms;
    out->wall_ms = m->wall_ms;
    return VMX_OK;
}

int vmx_multi_bind_texture(vmx_multi *m, const float *data, uint32_t width, uint32_t height, uint32_t channels) {
    if (!m) return fail(VMX_ERR_INVALID, "NULL argument");
    for (vmx_scene *sc : m->replica) {
        const int rc = vmx_scene_bind_texture(sc, data, width, height, channels);
        if (rc) return rc;
    }
    return VMX_OK;
}

int vmx_multi_render(vmx_multi *m, const vmx_camera *cam, const vmx_opts *opts, float *out_rgbaz, vmx_stats *stats) {
    if (!m || !cam || !opts || !out_rgbaz) return fail(VMX_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(m->mu);
    return multi_render(m, cam, opts, out_rgbaz, nullptr, stats,
                        [&](vmx_scene *sc, const vmx_opts *o, float *d_out, vmx_stats *st) {
                            return render_impl(sc, cam, o, d_out, sc->stream, st);
                        });
}

int vmx_multi_render_device(vmx_multi *m, const vmx_camera *cam, const vmx_opts *opts, void *d_out_rgbaz,
                            vmx_stats *stats) {
    if (!m || !cam || !opts || !d_out_rgbaz) return fail(VMX_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(m->mu);
    return multi_render(m, cam, opts, nullptr, d_out_rgbaz, stats,
                        [&](vmx_scene *sc, const vmx_opts *o, float *d_out, vmx_stats *st) {
                            return render_impl(sc, cam, o, d_out, sc->stream, st);
                        });
}

int vmx_multi_render_bruteforce(vmx_multi *m, const vmx_camera *cam, const vmx_opts *opts, uint32_t flags,
                                float *out_rgbaz, vmx_stats *stats) {
    if (!m || !cam || !opts || !out_rgbaz) return fail(VMX_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(m->mu);
    return multi_render(m, cam, opts, out_rgbaz, nullptr, stats,
                        [&](vmx_scene *sc, const vmx_opts *o, float *d_out, vmx_stats *st) {
                            return bruteforce_impl(sc, cam, o, flags, d_out, sc->stream, st);
                        });
}

} /* extern "C" */
