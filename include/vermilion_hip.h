/*
 * vermilion_hip.h — C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * This is the drop-in boundary for Vermilion's `Integrator::Render` seam
 * (reference: core/integrators/integrators.h:11-16, installed through
 * RenderEngine::assignIntegrator, core/engines/renderEngine.cpp:70-78, and
 * called from RenderEngine::draw, core/engines/renderEngine.cpp:163-164).
 * A `HipPathTracer : Vermilion::Integrator` adapter flattens
 * MeshEngine::sceneMeshes in createBVH order (core/engines/meshEngine.cpp:660-718)
 * into plain float arrays and calls the functions below; everything else
 * (Assimp import, OIIO texture read / image write, logging) stays on the host.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types cross this boundary
 *   - every function returns an int status (VMX_OK == 0); the message for the
 *     last failure on the calling thread is vmx_last_error()
 *   - nothing throws across the ABI; the library never falls back to a CPU
 *     path: without a usable HIP device every compute entry point fails with
 *     VMX_ERR_NO_DEVICE
 *   - synchronous and blocking unless a stream is passed explicitly
 *     (reference Render is one synchronous call, renderEngine.cpp:163-164)
 */
#ifndef VERMILION_HIP_H
#define VERMILION_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMX_ABI_VERSION 2 /* 2: vmx_camera.rotation_units / rotation_rad, vmx_multi_timings */

/* status codes */
#define VMX_OK 0
#define VMX_ERR_INVALID 1     /* bad argument (NULL, zero size, NaN geometry ...)  */
#define VMX_ERR_NO_DEVICE 2   /* no HIP device / device ordinal out of range        */
#define VMX_ERR_HIP 3         /* a hip* call failed; text in vmx_last_error()       */
#define VMX_ERR_DEPTH 4       /* BVH deeper than the reference's 64-entry traversal
                                 stack (core/accelerators/bvh.cpp:54)               */
#define VMX_ERR_NOMEM 5

/* sphere flags */
#define VMX_SPHERE_EMIT 1u /* hit writes `colour` into hitColour (meshEngine.cpp:382-383,415-416) */

/* sampling modes (vmx_opts.sampling) */
#define VMX_SAMPLING_PARITY 0u    /* r2 = 10*U, reference-faithful (pathtracer.cpp:156,170) */
#define VMX_SAMPLING_CORRECTED 1u /* r2 = U, an actual cosine-weighted lobe; not a parity mode */
#define VMX_SAMPLING_MODE_MASK 0xFFu
/* Flag, OR-ed into vmx_opts.sampling.  pathtracer.cpp:155,162 call the unqualified cos(r1) / sin(r1) with a
 * `float r1`.  Default reading: <cmath>'s float overloads are visible in the global namespace (the premise
 * under which integrators.cpp:170's abs(float) is std::abs(float), see VMX_BF_ABS_INT), so these are
 * cosf / sinf — evaluated on the device by glibc's algorithm (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c),
 * the libm the reference links on Linux.  With this flag they are C's `double cos(double)` / `sin` of the
 * widened argument, narrowed to float.  The two readings differ in the last bit for ~1.3 % of the arguments
 * (uniform in [0, 2 pi)); of 10^6 paths' radiance none differed (tests/test_oracle.py). */
#define VMX_SAMPLING_LIBM_DOUBLE 0x100u
/* Flag, OR-ed into vmx_opts.sampling; OFF by default.  Under the reference's sampling (r2 = 10 U, pathtracer.cpp:156,170)
 * nine in ten diffuse bounces produce a NaN direction and end the path.  A Radiance step adds
 * accumRadiance * hitColour (pathtracer.cpp:43), and hitColour is non-zero only where the hit — or a nearer-so-far test on
 * the way (meshEngine.cpp:377-420) — is a light sphere.  Whether a step is the path's last one whatever it hits is
 * decided by the path's own random draws alone (Russian roulette :56, the specular test :98 and r2 :156 / :170, for both
 * values of the material flag), and whether a light sphere can colour it by the ray alone.  With this flag a ray whose
 * step is provably the last one and cannot meet a light sphere is not traced — camera rays and bounce rays alike, 78 % of
 * all rays of a parity frame: the frame is bit-identical (Camera::mImage holds r,g,b and the sample count,
 * camera.cpp:106-113 — the primary hit distance Radiance also returns, :44-47, is not part of it), but
 * vmx_stats.rays_primary / rays_secondary count only the rays that were traced, so throughput figures are not comparable
 * with the default.  Applies to vmx_render* (split-wavefront and fused passes; not to vmx_radiance, whose output includes
 * that distance, nor to a call with collect_counters, whose totals are the reference's).  With VMX_SAMPLING_CORRECTED
 * (r2 = U) only Russian roulette past depth 5 ever ends a path by its draws alone and the flag changes little. */
#define VMX_SAMPLING_ELIDE_DEAD 0x200u

/*
 * One analytic sphere of MeshEngine::RayCast's hard-coded table
 * (core/engines/meshEngine.cpp:377-500).  Spheres are tested after the BVH,
 * in table order, each with `testHit > 0 && testHit < nearestHit`
 * (meshEngine.cpp:378); a nearer sphere overwrites the hit normal with
 *   normal_sign * normalize(hit - normal_centre)
 * and, if VMX_SPHERE_EMIT is set, overwrites hitColour with `colour`.
 * hitColour is never cleared by a later, nearer sphere (quirk kept on purpose).
 */
typedef struct vmx_sphere {
    float centre[3];
    float radius;
    float colour[3];
    uint32_t flags;
    float normal_centre[3];
    float normal_sign; /* +1 or -1 */
} vmx_sphere;

/*
 * Camera parameters = Vermilion::cameraSettings (core/camera/camera.h:32-47)
 * restricted to the fields PathTracer::Render reads.  The library applies the
 * Camera constructor's conversion (core/camera/camera.cpp:43-47):
 *   mRotation = (-rx, -ry, +rz) * 3.1415926535 / 180.
 */
#define VMX_ROTATION_DEGREES 0u /* rotation_deg = cameraSettings.rotation; the library applies camera.cpp:43-47 */
#define VMX_ROTATION_RADIANS 1u /* rotation_rad = Camera::mRotation as it stands (already negated in x, y and in
                                   radians): what an Integrator holds — PathTracer::Render reads mRotation, not the
                                   settings (pathtracer.cpp:219-221) — passed through without a degree round trip */
typedef struct vmx_camera {
    float position[3];     /* cameraSettings.position                        */
    float rotation_deg[3]; /* cameraSettings.rotation (degrees); read when rotation_units == VMX_ROTATION_DEGREES */
    float back_distance;   /* cameraSettings.fBackDistance -> mDistToFilm    */
    float back_size[2];    /* cameraSettings.fBackSizeX/Y  -> sensorSizeX/Y  */
    uint32_t image_res[2]; /* imageResX, imageResY                           */
    uint32_t rays_per_pixel; /* raysPerPixel -> uSamplesPerPixel             */
    uint32_t rotation_units; /* VMX_ROTATION_DEGREES (0, the default of a zeroed struct) or VMX_ROTATION_RADIANS */
    float rotation_rad[3]; /* Camera::mRotation (camera.h, set at camera.cpp:43-47); read when rotation_units ==
                              VMX_ROTATION_RADIANS: the three angles glm::rotate gets at pathtracer.cpp:219-221 */
} vmx_camera;

typedef struct vmx_opts {
    uint64_t seed;         /* the reference seeds from std::random_device (pathtracer.cpp:231);
                              here the stream of sample k of pixel p is keyed by (seed, p, k) */
    uint32_t early_stop;   /* 1: reference early-stop rule (pathtracer.cpp:290-311); 0: fixed spp */
    uint32_t sampling;     /* VMX_SAMPLING_PARITY / _CORRECTED, | VMX_SAMPLING_LIBM_DOUBLE | VMX_SAMPLING_ELIDE_DEAD */
    uint32_t rank;         /* image-stripe sharding: this call renders the stripes s   */
    uint32_t world;        /*   with s % world == rank; world 0 or 1 = whole image     */
    uint32_t stripe_rows;  /* rows per stripe; 0 -> 16                         */
    uint32_t samples_per_batch; /* fixed-spp mode: samples per pixel in flight per pass; 0 -> auto */
    uint32_t collect_counters;  /* 1: also count inner-node visits / triangle tests
                                   (instrumented kernels, slower; for roofline accounting) */
    uint32_t reserved[7];  /* 0 unless tuning: [0] pipeline form (bits 0-7: 0 default routing, 1 fused kernel for every
                              pass, 4 split wavefront for every pass; bit 8: plain one-phase shading; bit 9: two-phase
                              shading through k_shade_ends instead of rays sorted by the traversal kernel),
                              [1] max paths per pass, [2] tail threshold, [3] refill_min, [4] shade_min,
                              [5] bounce reordering key (A/B library only), [6] LDS stack levels — all forms and
                              settings produce the same frame (see vmx_api.cpp: render_impl, make_tuning) */
} vmx_opts;

/* per-stage figures: `primary` = Radiance steps taken at depth 0 (the fused
 * raygen + trace + shade kernel), `bounce` = every later step */
typedef struct vmx_stage_stats {
    uint64_t rays;          /* RayCast-equivalents with a finite direction            */
    uint64_t inner_visits;  /* inner-node visits (both child boxes tested); 0 unless collect_counters */
    uint64_t tri_tests;     /* Moller-Trumbore tests; 0 unless collect_counters       */
    uint64_t tri_hits;      /* rays whose BVH query hit a triangle                    */
    uint64_t continued;     /* path states written across a bounce boundary           */
    uint64_t launches;      /* kernel launches of this stage                          */
    double ms;              /* sum of hipEvent durations of those launches            */
} vmx_stage_stats;

typedef struct vmx_stats {
    uint64_t rays_primary;      /* = primary.rays                                      */
    uint64_t rays_secondary;    /* = bounce.rays (NaN directions are not rays)         */
    uint64_t samples;           /* pixel samples accumulated into the image            */
    uint64_t samples_discarded; /* speculative samples traced but dropped by early stop */
    uint64_t passes;            /* sample batches processed                            */
    uint64_t kernel_launches;
    double ms_total;            /* wall time of the call, host clock                   */
    double ms_device;           /* hipEvent time of the device work on the render stream */
    vmx_stage_stats primary; /* ms/launches: the depth-0 traversal kernel                  */
    vmx_stage_stats bounce;  /* ms/launches: bounce traversal kernels + the tail kernel    */
    vmx_stage_stats shade;   /* ms/launches only: the shading kernels of the split wavefront */
} vmx_stats;

/* per-kernel device time (hipEvent pairs on the render stream) of the LAST render / radiance call on
 * a scene: what bench.py's roofline object is computed from (the dominant kernel of a step) */
#define VMX_K_RAYGEN 0        /* k_raygen (+ live-path list under VMX_SAMPLING_ELIDE_DEAD)  */
#define VMX_K_TRACE_CAMERA 1  /* k_trace_w<0>: BVH traversal of the camera rays            */
#define VMX_K_SHADE_CAMERA 2  /* k_shade<0> (with k_shade_ends<0> + list compaction where used) */
#define VMX_K_TRACE_BOUNCE 3  /* k_trace_w<1>: BVH traversal of the bounce generations     */
#define VMX_K_SHADE_BOUNCE 4  /* k_shade<1> (with k_shade_ends<1> + list compaction where used) */
#define VMX_K_TAIL 5          /* k_paths<2>: fused kernel that finishes the last generations */
#define VMX_K_FUSED 6         /* k_paths<0>: whole small passes in one fused kernel        */
#define VMX_K_RESOLVE 7       /* k_resolve                                                 */
#define VMX_K_BRUTEFORCE 8    /* k_bruteforce                                              */
#define VMX_K_OTHER 9         /* first-generation kernels (pipeline forms 2, 3)            */
#define VMX_K_COUNT 10
typedef struct vmx_timings {
    double ms[VMX_K_COUNT];          /* summed over the launches of the call */
    double longest_ms[VMX_K_COUNT];  /* the longest single launch            */
    uint64_t launches[VMX_K_COUNT];
} vmx_timings;

typedef struct vmx_scene_desc {
    uint32_t ntris;
    uint32_t nspheres;
    uint32_t leaf_size;
    uint32_t n_nodes;      /* reference flat-tree node count (bvh.cpp:203)   */
    uint32_t n_leaves;     /* bvh.cpp:221                                    */
    uint32_t n_inner;      /* 2-wide records on the device                   */
    uint32_t max_depth;    /* root = 0                                       */
    uint32_t stack_entries;/* per-lane LDS stack entries the kernels use     */
    uint64_t device_bytes; /* HBM held by the scene                          */
    int32_t device;
    uint32_t pad;
} vmx_scene_desc;

/* full output tuple of MeshEngine::RayCast (meshEngine.cpp:239-509) for one ray */
typedef struct vmx_rayhit {
    float location[3]; /* pHitLocation                                         */
    float distance;    /* pHitDistance (INFINITY on a miss)                    */
    float normal[3];   /* pHitNormal                                           */
    int32_t tri_id;    /* createBVH push-order index of the BVH hit, -1 if none */
    float uv[2];       /* pHitTexCoord                                         */
    float tri_t;       /* BVH t (999999999.f if none), bvh.cpp:48              */
    uint32_t flags;    /* bit0: return value (nearest < INF); bit1: material non-null */
    float colour[3];   /* pHitColour                                           */
    uint32_t pad;
} vmx_rayhit;

typedef struct vmx_scene vmx_scene;

/* ---- library ---------------------------------------------------------- */
int vmx_abi_version(void);
const char *vmx_last_error(void);
/* number of HIP devices visible to the library (0 if none / runtime missing) */
int vmx_device_count(void);

/* The reference's eight spheres (meshEngine.cpp:377-500): 2 lights + 6 walls. */
const vmx_sphere *vmx_default_spheres(uint32_t *count);

/* ---- scene ------------------------------------------------------------ */
/*
 * Replaces MeshEngine::createBVH + BVH::BVH/build (meshEngine.cpp:649-724,
 * bvh.cpp:155-279) for the device: `pos`/`nrm` are [ntris*9] floats
 * (v0,v1,v2 / n0,n1,n2 per triangle), `uv` is [ntris*6] or NULL (zeros),
 * in createBVH push order — that order defines triangle IDs.
 * spheres == NULL && nspheres == 0 selects vmx_default_spheres(); a non-NULL
 * pointer with nspheres == 0 means no spheres at all.
 * leaf_size 0 -> 4 (bvh.h:29).  The BVH is built on the host with the
 * reference's topology, flattened to 2-wide records and uploaded to `device`.
 */
int vmx_scene_create(const float *pos, const float *nrm, const float *uv, uint32_t ntris,
                     const vmx_sphere *spheres, uint32_t nspheres, uint32_t leaf_size,
                     int device, vmx_scene **out);
/* BVH builders for vmx_scene_create_ex */
#define VMX_BVH_REFERENCE 0u /* BVH::build's topology (bvh.cpp:179-279): triangle-ID / tie parity        */
#define VMX_BVH_SAH 1u       /* binned-SAH quality tree: same triangle tests and nearest
                                distance, but exact-distance ties and `near > t` pruning follow ITS order */
#define VMX_BVH_LBVH 2u      /* linear BVH built on the GPU (Morton sort, Karras hierarchy, bottom-up fit):
                                for scenes that change per frame; same caveat as SAH.  Many coincident
                                centroids can make the tree deeper than the 64-entry traversal stack the
                                reference allows (bvh.cpp:54): VMX_ERR_DEPTH, use another builder */
#define VMX_BVH_PLOC 3u      /* quality tree built on the GPU by parallel locally-ordered clustering (Morton
                                sort, rounds of nearest-neighbour merging within 16 positions): close to the SAH
                                tree's visit counts at a few ms per build; same caveat as SAH */
int vmx_scene_create_ex(const float *pos, const float *nrm, const float *uv, uint32_t ntris,
                        const vmx_sphere *spheres, uint32_t nspheres, uint32_t leaf_size, uint32_t builder,
                        int device, vmx_scene **out);
int vmx_scene_destroy(vmx_scene *scene);
/*
 * Replaces MeshEngine::bindTexture (meshEngine.cpp:74-93) minus the OpenImageIO read: `data` is the
 * float image bindTexture would hand to VermiTexture (height rows of width texels of `channels`
 * floats, 1..4 channels, width/height <= 65535 as VermiTexture stores uint16_t).  Only the FIRST
 * bound texture is ever sampled by the path tracer (pathtracer.cpp:63-66: boundTextures[0], wrap
 * + nearest, meshEngine.cpp:21-46); the SECOND one is kept for BruteForceTracer, which reads
 * boundTextures[1] as its albedo (integrators.cpp:141-147); later calls are counted and otherwise
 * ignored, as there.
 */
int vmx_scene_bind_texture(vmx_scene *scene, const float *data, uint32_t width, uint32_t height,
                           uint32_t channels);
int vmx_scene_describe(const vmx_scene *scene, vmx_scene_desc *out);
int vmx_scene_timings(const vmx_scene *scene, vmx_timings *out);
/*
 * Host-side BVH topology in the reference's flat layout (bvh.h:11-14): per
 * node start, nPrims, rightOffset ([n_nodes] each, any may be NULL), bbox
 * [n_nodes*6] (min,max) and the final build_prims permutation [ntris].
 */
int vmx_scene_bvh(const vmx_scene *scene, uint32_t *start, uint32_t *nprims,
                  uint32_t *right_offset, float *bbox, uint32_t *prim_order);

/* ---- parity hooks (explicit ray batches, host buffers) ----------------- */
/* BVH::getIntersection (bvh.cpp:47-145): tri_id[n] (-1 = miss), t[n] */
int vmx_trace(const vmx_scene *scene, const float *origin, const float *dir, uint32_t n,
              int32_t *tri_id, float *t);
/* MeshEngine::RayCast (meshEngine.cpp:239-509) */
int vmx_raycast(const vmx_scene *scene, const float *origin, const float *dir, uint32_t n,
                vmx_rayhit *out);
/*
 * Primary-hit AOV: generates sample k's camera ray of every pixel on the
 * device (pathtracer.cpp:251-280) and returns BVH::getIntersection's result,
 * tri_id[W*H] / t[W*H] in pixel order (the "primary-hit triangle ID" map).
 */
int vmx_primary_ids(const vmx_scene *scene, const vmx_camera *cam, const vmx_opts *opts, uint32_t k,
                    int32_t *tri_id, float *t);
/*
 * Radiance (pathtracer.cpp:21-198) for n explicit camera rays; ray i draws
 * from the stream keyed (opts->seed, i, 0) with the two pixel-jitter draws
 * skipped.  out[n*4] = accumColour (rgb, w = primary hit distance or -100).
 */
int vmx_radiance(const vmx_scene *scene, const float *origin, const float *dir, uint32_t n,
                 const vmx_opts *opts, float *out, vmx_stats *stats);

/* cosf(x[i]), sinf(x[i]) for x in [0, 2 pi] exactly as the shading kernels evaluate the cos(r1) / sin(r1) of
 * pathtracer.cpp:162 under the default reading (see VMX_SAMPLING_LIBM_DOUBLE); host buffers */
int vmx_trig(const float *x, uint32_t n, float *cos_out, float *sin_out, int device);

/* ---- render (replaces PathTracer::Render, pathtracer.cpp:200-328) ------ */
/* number of image rows / pixels this (rank, world) owns */
int vmx_local_rows(uint32_t height, uint32_t stripe_rows, uint32_t rank, uint32_t world,
                   uint32_t *rows);
/*
 * Render into a caller-owned HOST buffer.  world <= 1: out is W*H*5 floats in
 * Camera::mImage RGBAZ layout (camera.cpp:106-113): r,g,b in [0,1], alpha 1,
 * depth = samples taken (pathtracer.cpp:318-323).  world > 1: out holds only
 * this rank's rows, packed in ascending row order (local_rows*W*5 floats).
 */
int vmx_render(const vmx_scene *scene, const vmx_camera *cam, const vmx_opts *opts,
               float *out_rgbaz, vmx_stats *stats);
/* Same, but `d_out` is DEVICE memory on the scene's device and the work is
 * enqueued on `stream` (a hipStream_t; NULL = the library's own stream).
 * Blocks until the frame is complete (the early-stop loop needs the host). */
int vmx_render_device(const vmx_scene *scene, const vmx_camera *cam, const vmx_opts *opts,
                      void *d_out_rgbaz, void *stream, vmx_stats *stats);
/*
 * Multi-GPU assembly on the root: `d_gathered` = world packed per-rank buffers
 * back to back, each padded to `rank_stride_floats`; writes the W*H*5 frame.
 */
int vmx_assemble_device(const void *d_gathered, uint64_t rank_stride_floats, uint32_t width,
                        uint32_t height, uint32_t stripe_rows, uint32_t world, void *d_frame,
                        int device, void *stream);

/* ---- multi-device render in one process ---------------------------------------------------
 * Vermilion's main.cpp (main.cpp:58-104) is ONE process; these entry points let its Integrator use
 * every GPU of the node: the scene is replicated on each device of the list (one host-side BVH build,
 * one upload per device), every replica renders its interleaved stripes of `stripe_rows` rows
 * (vmx_opts.rank/world are set by the library; the RNG is keyed by the global pixel, so the frame does
 * not depend on the device count), the packed stripes are pushed device-to-device into a gather buffer
 * on devices[0] (peer copies over xGMI — each peer has its own link to the root, no ring) and
 * de-interleaved there.  A device may appear more than once in the list (rehearsal of the N-rank path
 * on fewer GPUs).  The result is bit-identical to vmx_render on one device.
 */
typedef struct vmx_multi vmx_multi;
int vmx_multi_create(const float *pos, const float *nrm, const float *uv, uint32_t ntris, const vmx_sphere *spheres,
                     uint32_t nspheres, uint32_t leaf_size, uint32_t builder, const int *devices, uint32_t ndevices,
                     vmx_multi **out);
int vmx_multi_destroy(vmx_multi *m);
uint32_t vmx_multi_world(const vmx_multi *m);
/* per entry of the device list: its device, and how its stripes reach devices[0] — 2 same device, 1 direct peer
 * copy (xGMI), 0 staged through the host (peer access unavailable; a warning went to stderr at creation).
 * Either array may be NULL; each holds vmx_multi_world() ints. */
int vmx_multi_routes(const vmx_multi *m, int *devices, int *routes);
/* The exchange step of the LAST vmx_multi_render* call, timed apart from the rendering ("gather time
 * separately"): hipEvent pairs on each replica's stream around its stripes' copy into the root's gather buffer and on the
 * root's stream around the de-interleave kernel.  render_ms / copy_ms: per entry of the device list
 * (vmx_multi_world() doubles each), either may be NULL. */
typedef struct vmx_multi_times {
    double slowest_render_ms; /* max over the replicas of vmx_stats.ms_device of its stripes' render  */
    double gather_ms;         /* max over the replicas of its device-to-device (or staged) copy        */
    double gather_sum_ms;     /* sum of those copies                                                    */
    double assemble_ms;       /* k_assemble on devices[0]                                               */
    double wall_ms;           /* host clock from handing the jobs out to the assembled frame            */
    uint32_t world, pad;
} vmx_multi_times;
int vmx_multi_timings(const vmx_multi *m, vmx_multi_times *out, double *render_ms, double *copy_ms);
int vmx_multi_bind_texture(vmx_multi *m, const float *data, uint32_t width, uint32_t height, uint32_t channels);
/* whole frame (W*H*5 floats) into a caller-owned HOST buffer / into DEVICE memory on devices[0];
 * stats: rays and samples summed over the devices, times = the slowest device's */
int vmx_multi_render(vmx_multi *m, const vmx_camera *cam, const vmx_opts *opts, float *out_rgbaz, vmx_stats *stats);
int vmx_multi_render_device(vmx_multi *m, const vmx_camera *cam, const vmx_opts *opts, void *d_out_rgbaz,
                            vmx_stats *stats);
int vmx_multi_render_bruteforce(vmx_multi *m, const vmx_camera *cam, const vmx_opts *opts, uint32_t flags,
                                float *out_rgbaz, vmx_stats *stats);

/* ---- BruteForceTracer: the engine's DEFAULT integrator --------------------
 * Replaces Vermilion::BruteForceTracer::Render (core/integrators/integrators.cpp:9-186; installed by
 * RenderEngine::Initialise when no integrator is assigned, core/engines/renderEngine.cpp:49-53):
 * per pixel up to raysPerPixel jittered camera rays (:59-81), N.L against a point light at
 * (500,1100,2000) (:16,83-88), the normal perturbed by boundTextures[0] (:98-106), one mirror
 * probe whose MISS lifts the term to 0.9 N.L + 0.1 (:119-137), albedo from boundTextures[1] or the
 * constant (0.890196078, 0.258823529, 0.203921569) (:141-156), and a convergence break once more
 * than two samples are in (:166-172).  Output layout as vmx_render (RGBAZ, camera.cpp:106-113) with
 * alpha = accum.w / samples (the hit fraction, :180) and depth = the LAST sample's hitDistance
 * (:181; INFINITY after a miss, meshEngine.cpp:507).  vmx_opts: seed, rank/world/stripe_rows are
 * used (the jitter of sample s of pixel p comes from the stream keyed (seed, p, s): the reference
 * shares one time(0)-seeded std::mt19937 between its OpenMP threads, :30, and is not reproducible);
 * early_stop / sampling / samples_per_batch do not apply.  rays_per_pixel must be 1..65535 (the
 * reference's loop counter is a uint16_t, :59).
 */
#define VMX_BF_ABS_INT 1u /* read the unqualified `abs(float)` of integrators.cpp:170 as C's abs(int) (the sum is
                             truncated to int first) instead of std::abs(float), the default */
int vmx_render_bruteforce(const vmx_scene *scene, const vmx_camera *cam, const vmx_opts *opts, uint32_t flags,
                          float *out_rgbaz, vmx_stats *stats);
int vmx_render_bruteforce_device(const vmx_scene *scene, const vmx_camera *cam, const vmx_opts *opts, uint32_t flags,
                                 void *d_out_rgbaz, void *stream, vmx_stats *stats);

/* ---- frame output (the step after the path; replaces the conversion loop of
 * Camera::saveFrame, core/camera/camera.cpp:140-175, for the default RGBAZ mode) ------------ */
/*
 * rgba8[p*4 + c] = (unsigned char)floor(frame[p*5 + c] * 255)   c = 0..3   (camera.cpp:159-162)
 * depth[p]       = frame[p*5 + 4]                                          (camera.cpp:163)
 * All three pointers are DEVICE memory on `device`; depth may be NULL.  The PNG/EXR encoding
 * (OpenImageIO, camera.cpp:178-188) stays with the host application.
 */
int vmx_quantize_device(const void *d_frame_rgbaz, uint64_t npixels, void *d_rgba8, void *d_depth, int device,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VERMILION_HIP_H */
