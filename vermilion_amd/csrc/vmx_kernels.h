// vmx_kernels.h — host-callable launchers of the gfx950 kernels (vmx_kernels.hip).
// All launchers enqueue on `stream` and return the hipError_t of the launch as int.
#pragma once
#include <stdint.h>
#include "vmx_device.h"

namespace vmx {

constexpr uint32_t kSubQueues = 16;  // path sub-queues (one tail counter each, own cache line)

struct QueueDev {  // first-generation kernels only (vmx_kernels_ab.inc): queue of 96-byte path records
    void *planes;             // float4[kPathPlanes][capacity]
    unsigned int *counts;     // [kSubQueues * 32] — counter q at counts[q*32] (128-byte spacing)
    uint32_t capacity;        // total slots = kSubQueues * sub_capacity
    uint32_t sub_capacity;
};

struct PixelStateDev {
    void *accum;              // float4[npix_local]  (rgb sum, w unused)
    unsigned int *count;      // samples taken
    unsigned int *cursor;     // next linear sample index k
};

// Per-pass path state, one slot per path id `pid` = sample_plane * n_pad + slot (fixed for the
// pass: compaction between bounces moves 4-byte ids, never the state).
struct PathArrays {
    void *rayA;   // camera rays: float4 (d.x, d.y, d.z, bits(depth flag)), read densely by path id
    // paths past their first hit: one 64-byte record per path id — (o.xyz, d.x) (d.yz, bits(depth), 0)
    // (xoshiro s0, s1) (s2, s3).  Few paths live on (16 % after the first hit with the reference's
    // sampling), so their readers touch scattered ids: one line per path instead of four.
    void *state;
    void *hit;    // float2 (t of the BVH query, bits(leaf slot or -1))
    void *rad;    // float4 accumColour, updated in place; final when the path ends
    void *thr;    // float4 accumRadiance (rgb); only with a bound texture (else it stays 1,1,1)
    // one bit per path id: rad[pid] holds something k_resolve has to read — the path went on past its first hit, or
    // its first hit already gave it a non-zero colour.  ~85 % of the bench frame's paths end black at their first hit:
    // k_shade<0> does not store their radiance and k_resolve adds an exact +0 instead of loading it.  NULL: every
    // path's radiance is in `rad` (fused kernel, vmx_radiance)
    unsigned long long *rad_mask;
};

// compacted list of live path ids: kSubQueues sub-lists, each `sub_capacity` ids long
struct IdQueue {
    unsigned int *ids;
    unsigned int *counts;   // counter q at counts[q*32]
    uint32_t sub_capacity;
    uint32_t pad;
};

// work source of k_paths
struct WorkDev {
    unsigned int *heads;        // [nsrc * 32] reservation heads, zeroed before the launch
    uint32_t nsrc;              // 8 image bands (primary) or kSubQueues (queue)
    uint32_t refill_min;        // refill when this many lanes of a wave are idle
    uint32_t reserve;           // items a wave reserves per atomic on a head (multiple of 64)
    uint32_t shade_min;         // shade when this many lanes have finished traversal
    uint32_t leaf_min;          // k_paths: run the triangle step when this many lanes sit at a leaf
    uint32_t lds_entries;       // stack levels kept in LDS
    uint32_t overflow_entries;  // deeper levels, in the global slab below (64 lanes x 8 B each)
    void *overflow_stack;       // uint2[waves][overflow_entries][64]
    // primary source
    const unsigned int *active;
    uint32_t n_active, n_pad, samples;
    FastDiv div_samples;        // path id / samples (pixel-major ids); make_fastdiv(samples), set with samples
    uint32_t band_slots;        // slots per band (multiple of 64)
    uint32_t band_items;        // band_slots * samples
    uint32_t pixel_major;       // 1: pid = slot * samples + j (samples of a pixel contiguous); 0: j * n_pad + slot
    // queue source
    IdQueue qids;     // k_trace_w / k_trace_q / k_shade / tail (path ids)
    // VMX_SAMPLING_ELIDE_DEAD: camera paths whose radiance is provably zero are not traced.  k_raygen<1> leaves one word
    // of live bits + its popcount per 64 path ids; launch_live_compact turns them into the ordered list of live path ids;
    // k_raygen_live writes their rays to rayA[list position], k_trace_w<0> the hit records to hit[list position], and
    // k_shade<0> walks the list (NULL: every path of the pass, rayA / hit indexed by path id)
    unsigned long long *live_mask;
    unsigned int *live_cnt;
    const unsigned int *live_ids;
    const unsigned int *live_count;  // device scalar: entries of live_ids
    // classified output of k_trace_w<.., SORT> (DESIGN_HISTORY.md 5.1): 32-byte records of the rays that still need shading,
    // the number of list entries reserved so far (device scalar, multiple of 256), the frame's counters
    void *out_rec;
    unsigned int *out_count;
    DevCounters *out_ctr;
    uint32_t keep_all;      // k_trace_w<.., SORT>: hand EVERY finished ray on as a record, settle none (the one-phase form
                            // of a bounce generation: k_shade reads dense (t, leaf slot, path id) records instead of hit[pid])
    uint32_t pool_slots;    // A/B library: ray slots per block of k_trace_pool (vmx_trace_pool.inc)
    uint32_t out_capacity;  // entries the list holds (paths of the pass + 256 per wave of the launch): an append never
                            // writes at or past it, and a count beyond it is reported (DevCounters::overflow)
    // two-phase shading: the ordered list of positions k_shade_ends left for k_shade (NULL: one phase)
    const unsigned int *flat_ids;
    const unsigned int *flat_count;
    // camera rays: per-frame origin-relative node / triangle tables (k_camera_tables)
    const void *cam_inner;  // float4[8 * n_inner * 4]: one copy per direction octant, octant 0 = plain (lo, hi)
    uint32_t cam_n_inner;   // records per copy
    const void *cam_tris;   // float4[ntris * 4]
};

struct LaunchCfg {
    uint32_t grid;            // persistent blocks
    uint32_t block;           // threads per block (multiple of 64)
    uint32_t lds_bytes;       // dynamic LDS = waves_per_block * stack_entries * 512
};

// ---- parity hooks -----------------------------------------------------------
int launch_trace(const SceneDev &sc, const float *o, const float *d, uint32_t n, int32_t *tri_id,
                 float *t, DevCounters *counters, bool count, LaunchCfg cfg, void *stream);
int launch_raycast(const SceneDev &sc, const float *o, const float *d, uint32_t n, void *out_rayhit,
                   LaunchCfg cfg, void *stream);
int launch_primary_ids(const SceneDev &sc, const FrameDev &fr, uint32_t k, int32_t *tri_id, float *t,
                       LaunchCfg cfg, void *stream);

int launch_trig(const float *x, uint32_t n, float *cs, float *sn, void *stream);

// ---- render pipeline -----------------------------------------------------------
int launch_init_pixels(PixelStateDev px, uint32_t npix, void *stream);
int launch_zero_u32(unsigned int *p, uint32_t n, void *stream);
#ifdef VMX_AB_KERNELS  // first-generation kernels (vmx_kernels_ab.inc), A/B library only
// raygen + trace + shade for `samples` samples of each of n_active pixels.
// loop_to_end: every lane follows its path to termination (megakernel form).
int launch_primary(const SceneDev &sc, const FrameDev &fr, const unsigned int *active,
                   uint32_t n_active, uint32_t samples, PixelStateDev px, QueueDev qout, void *rad,
                   DevCounters *counters, bool count, bool loop_to_end, LaunchCfg cfg, void *stream);
// explicit-ray path starts for vmx_radiance: writes initial path states
int launch_radiance_init(const float *o, const float *d, uint32_t n, uint64_t seed, QueueDev qout,
                         void *stream);
// one bounce (or, loop_to_end, all remaining bounces) of every queued path
int launch_bounce(const SceneDev &sc, float r2scale, uint32_t libm_double, QueueDev qin, uint32_t max_chunks, QueueDev qout,
                  void *rad, DevCounters *counters, bool count, bool loop_to_end, bool first_step,
                  LaunchCfg cfg, void *stream);
int query_blocks_per_cu(uint32_t block, uint32_t lds_bytes, bool count, int *primary_blocks, int *bounce_blocks);
// phase-pure bounce traversal with the ray state in LDS (vmx_trace_pool.inc; profiles/r04_state_pool.txt)
int launch_trace_pool(const SceneDev &sc, const WorkDev &wk, PathArrays pa, LaunchCfg cfg, void *stream);
int query_trace_pool(uint32_t block, uint32_t pool_slots, uint32_t lds_levels, uint32_t *lds_bytes, int *blocks);
#endif
// per-pixel accumulation in sample order, early-stop rule, pixel write, next active list
// persistent fused kernel with per-lane refill (ray generation source): every lane keeps its path to the end
int launch_paths(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PixelStateDev px, void *rad,
                 DevCounters *counters, bool count, LaunchCfg cfg, void *stream);
int query_paths_blocks_per_cu(uint32_t block, uint32_t lds_bytes, bool count, int *blocks);
// split wavefront: persistent trace kernel (per-lane refill) ...
int launch_camera_tables(const SceneDev &sc, uint32_t n_inner, float ox, float oy, float oz, void *cam_inner,
                         void *cam_tris, void *stream);
int launch_raygen(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PixelStateDev px, PathArrays pa, void *stream);
int launch_raygen_live(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PixelStateDev px, PathArrays pa, void *stream);
size_t live_compact_tmp_bytes(uint32_t nwords);
int launch_live_compact(const unsigned long long *live_mask, const unsigned int *live_cnt, uint32_t nwords,
                        unsigned int *offs, unsigned int *ids, unsigned int *count, void *tmp, size_t tmp_bytes, void *stream);
int launch_trace_q(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PixelStateDev px, PathArrays pa,
                   DevCounters *counters, bool count, bool from_queue, LaunchCfg cfg, void *stream);
int query_trace_q_blocks_per_cu(uint32_t block, uint32_t lds_bytes, bool count, bool from_queue, bool sorted, bool live, int *blocks);
// ... and the wide shading kernel: RayCast tail + Radiance step + id compaction
int launch_shade(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PixelStateDev px, PathArrays pa,
                 IdQueue qout, uint32_t max_chunks, DevCounters *counters, bool from_queue, void *stream);
int launch_shade_ends(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PathArrays pa, unsigned long long *full_mask,
                      unsigned int *full_cnt, uint32_t max_chunks, DevCounters *counters, bool from_queue, void *stream);
// follows every queued path (ids) to its end in one launch
int launch_tail(const SceneDev &sc, const FrameDev &fr, const WorkDev &wk, PathArrays pa, DevCounters *counters,
                bool count, LaunchCfg cfg, void *stream);
int launch_radiance_init_ids(const float *o, const float *d, uint32_t n, uint64_t seed, PathArrays pa,
                             IdQueue qout, void *stream);
int launch_resolve(const FrameDev &fr, const unsigned int *active, uint32_t n_active, uint32_t samples,
                   bool pixel_major, const void *rad, const unsigned long long *rad_mask, PixelStateDev px, unsigned int *next_active,
                   unsigned int *next_count, float *out_rgbaz, DevCounters *counters, void *stream);
// BruteForceTracer::Render (integrators.cpp:9-186): one lane per pixel of `order` (tile-ordered local pixels)
int launch_bruteforce(const SceneDev &sc, const FrameDev &fr, const unsigned int *order, uint32_t npix, uint32_t flags,
                      float *out, DevCounters *counters, void *longs /* 64 bytes per pixel */, unsigned int *long_count,
                      const WorkDev &stack /* lds_entries, overflow_entries, overflow_stack */, LaunchCfg cfg, void *stream);
int launch_bruteforce_long(const SceneDev &sc, const FrameDev &fr, uint32_t flags, float *out, DevCounters *counters,
                           const void *longs, const unsigned int *long_count, uint32_t count, const WorkDev &stack,
                           LaunchCfg cfg, void *stream);
int launch_quantize(const float *frame, uint64_t npix, void *rgba8, float *depth, void *stream);
int launch_assemble(const float *gathered, uint64_t rank_stride_floats, uint32_t width, uint32_t height,
                    uint32_t stripe_rows, uint32_t world, float *frame, void *stream);

// ---- live-path reordering between bounce generations (path_sort.hip) -----------------------------
struct SortKeyCfg {
    float lo[3], inv[3];  // scene bounds: cell = (o - lo) * inv in [0,1)
    uint32_t obits;       // bits per axis of the origin cell (Morton-interleaved), 0 = no origin part
    uint32_t dbits;       // bits per axis of the octahedral direction cell, 0 = no direction part
    uint32_t dir_major;   // 1: direction cell in the high bits
    uint32_t chunk_log2;  // > 0: key = (queue position >> chunk_log2, direction cell): direction order inside chunks
};
size_t path_sort_tmp_bytes(uint32_t max_n);
// sorts every sub-queue of `q` (h_counts[kSubQueues] entries each) by the key of its paths' next rays into ids_out
// (same sub-queue layout); keys_a / keys_b: scratch of the queue's size
int path_sort_ids(const IdQueue &q, const uint32_t *h_counts, const void *state, const SortKeyCfg &cfg,
                  unsigned int *keys_a, unsigned int *keys_b, unsigned int *ids_out, void *tmp, size_t tmp_bytes,
                  void *stream);

}  // namespace vmx
