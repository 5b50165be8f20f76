/*
 * vmx_oracle.h — C API of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (vermilion_amd/, libvermilion_hip.so) never
 * links, imports or calls anything declared here.
 *
 * Struct types (vmx_sphere, vmx_camera, vmx_opts, vmx_stats, vmx_rayhit) are
 * the plain-C ones of include/vermilion_hip.h so that the same test inputs
 * feed both sides.
 */
#ifndef VMX_ORACLE_H
#define VMX_ORACLE_H

#include <stdint.h>
#include "../include/vermilion_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

#define ORC_RNG_XOSHIRO_KEYED 0 /* stream keyed by (seed, pixel, sample): GPU parity */
#define ORC_RNG_MT19937_64 1    /* thread-local std::mt19937_64 as pathtracer.cpp:231, seeded seed+thread */

typedef struct orc_trace_counters {
    uint64_t inner_visits;
    uint64_t tri_tests;
    uint64_t pops;
    uint64_t max_stack;
} orc_trace_counters;

const char *orc_build_flags(void);
const vmx_sphere *orc_default_spheres(uint32_t *count);

orc_scene *orc_scene_create(const float *pos, const float *nrm, const float *uv, uint32_t ntris,
                            const vmx_sphere *spheres, uint32_t nspheres, uint32_t leaf_size);
orc_scene *orc_scene_create_from_tree(const float *pos, const float *nrm, const float *uv, uint32_t ntris,
                                      const vmx_sphere *spheres, uint32_t nspheres, uint32_t n_nodes,
                                      const uint32_t *start, const uint32_t *nprims, const uint32_t *right_offset,
                                      const float *bbox, const uint32_t *prim_order);
void orc_scene_destroy(orc_scene *);
int orc_scene_bind_texture(orc_scene *, const float *data, uint32_t w, uint32_t h, uint32_t channels);
void orc_scene_describe(const orc_scene *, uint32_t *n_nodes, uint32_t *n_leaves,
                        uint32_t *max_depth);
void orc_scene_bvh(const orc_scene *, uint32_t *start, uint32_t *nprims, uint32_t *right_offset,
                   float *bbox, uint32_t *prim_order);

/* BVH::getIntersection */
void orc_trace(const orc_scene *, const float *o, const float *d, uint32_t n, int32_t *tri_id,
               float *t, orc_trace_counters *counters);
/* MeshEngine::RayCast */
void orc_raycast(const orc_scene *, const float *o, const float *d, uint32_t n, vmx_rayhit *out);
/* Radiance with the keyed xoshiro stream (seed, i, 0), jitter draws skipped */
void orc_radiance(const orc_scene *, const float *o, const float *d, uint32_t n,
                  const vmx_opts *opts, float *out4, vmx_stats *stats);
/* Radiance with std::mt19937_64(seeds[i]) + uniform_real_distribution<double>, as the reference */
/* audit of the elision / two-phase shading argument on the oracle's own Radiance (DESIGN_HISTORY.md 5.1): out6 = steps,
 * predicted_last, predicted_dead, and the three violation counts not_last, dead_changed, colour_mismatch (all must be 0) */
void orc_audit_elision(const orc_scene *, const float *o, const float *d, uint32_t n, const vmx_opts *opts, float *out4,
                       uint64_t *out6);
void orc_radiance_mt(const orc_scene *, const float *o, const float *d, uint32_t n,
                     const uint64_t *seeds, uint32_t sampling, float *out4);

/* camera matrix (3x3 upper-left of the reference's mat4, column-major m[col*3+row]) */
void orc_camera_matrix(const vmx_camera *cam, float m9[9]);
/* primary ray of sample k of every pixel in [0, W*H): o[n*3], d[n*3] */
void orc_primary_rays(const vmx_camera *cam, const vmx_opts *opts, uint32_t k, float *o,
                      float *d);
/* raw keyed stream: n successive 64-bit outputs of the (seed, pixel, k) stream */
void orc_stream(uint64_t seed, uint32_t pixel, uint32_t k, uint32_t n, uint64_t *out);
uint64_t orc_splitmix64(uint64_t *state);

/* PathTracer::Render — whole image, W*H*5 floats RGBAZ.  threads 0 -> OpenMP default */
void orc_render(const orc_scene *, const vmx_camera *cam, const vmx_opts *opts, int rng_mode,
                int threads, float *out_rgbaz, vmx_stats *stats);
/* BruteForceTracer::Render (integrators.cpp:9-186); flags: VMX_BF_* of vermilion_hip.h */
void orc_render_bruteforce(const orc_scene *, const vmx_camera *cam, const vmx_opts *opts, uint32_t flags,
                           int threads, float *out_rgbaz, vmx_stats *stats);
int orc_max_threads(void);
/* cosf / sinf as pathtracer.cpp:162 gets them under the default reading (glibc's algorithm restated), and the
 * count of floats in a bit-pattern range on which that restatement differs from the host's libm */
void orc_trig(const float *x, uint32_t n, float *cos_out, float *sin_out);
void orc_trig_compare_libm(uint32_t lo_bits, uint32_t hi_bits, uint64_t *cos_diff, uint64_t *sin_diff);
/* check of the kernels' three-operation division by the image size against the IEEE division (see vmx_oracle.cpp) */
void orc_check_div_by_count(uint32_t n, uint32_t lo_bits, uint32_t hi_bits, uint64_t *float_diff, uint64_t *double_diff);
/* Camera::saveFrame conversion (camera.cpp:159-163) */
void orc_quantize(const float *frame, uint64_t npix, unsigned char *rgba8, float *depth);

#ifdef __cplusplus
}
#endif
#endif
