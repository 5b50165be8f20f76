// vmx_kernels.h — host-callable launchers of the gfx950 kernels (vmx_kernels.hip).
// All launchers enqueue on `stream` and return the hipError_t of the launch as int.
#pragma once
#include <stdint.h>
#include "vmx_device.h"

namespace vmx {

constexpr uint32_t kSubQueues = 16;  // path sub-queues (one tail counter each, own cache line)

struct QueueDev {
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

// ---- render pipeline -----------------------------------------------------------
int launch_init_pixels(PixelStateDev px, uint32_t npix, void *stream);
int launch_zero_u32(unsigned int *p, uint32_t n, void *stream);
// raygen + trace + shade for `samples` samples of each of n_active pixels.
// loop_to_end: every lane follows its path to termination (megakernel form).
int launch_primary(const SceneDev &sc, const FrameDev &fr, const unsigned int *active,
                   uint32_t n_active, uint32_t samples, PixelStateDev px, QueueDev qout, void *rad,
                   DevCounters *counters, bool count, bool loop_to_end, LaunchCfg cfg, void *stream);
// explicit-ray path starts for vmx_radiance: writes initial path states
int launch_radiance_init(const float *o, const float *d, uint32_t n, uint64_t seed, QueueDev qout,
                         void *stream);
// one bounce (or, loop_to_end, all remaining bounces) of every queued path
int launch_bounce(const SceneDev &sc, float r2scale, QueueDev qin, uint32_t max_chunks, QueueDev qout,
                  void *rad, DevCounters *counters, bool count, bool loop_to_end, bool first_step,
                  LaunchCfg cfg, void *stream);
// per-pixel accumulation in sample order, early-stop rule, pixel write, next active list
int launch_resolve(const FrameDev &fr, const unsigned int *active, uint32_t n_active, uint32_t samples,
                   const void *rad, PixelStateDev px, unsigned int *next_active,
                   unsigned int *next_count, float *out_rgbaz, DevCounters *counters, void *stream);
int launch_assemble(const float *gathered, uint64_t rank_stride_floats, uint32_t width, uint32_t height,
                    uint32_t stripe_rows, uint32_t world, float *frame, void *stream);

// occupancy helpers (host): blocks per CU for the trace-heavy kernels at this LDS size
int query_blocks_per_cu(uint32_t block, uint32_t lds_bytes, bool count, int *primary_blocks,
                        int *bounce_blocks);

}  // namespace vmx
