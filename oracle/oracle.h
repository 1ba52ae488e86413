/* oracle.h -- C entry points of the parity oracle (TEST INFRASTRUCTURE; see oracle.cpp header).
 * Loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg. */
#ifndef SOLSTRALE_ORACLE_H
#define SOLSTRALE_ORACLE_H
#include <stdint.h>

#include "../include/solstrale_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_F64 0 /* the reference's arithmetic */
#define ORC_F32 1 /* fp32 contract: the GPU parity target */

typedef struct OrcStats {
  uint64_t samples, rays, node_visits, sphere_tests, quad_tests, triangle_tests, shades, texel_fetches;
  uint32_t threads, _pad;
  /* rays of `rays` that the DEVICE traces as well: it ends a path at a ScatterPdf level whose factor colour * probability has no positive
   * component (that level returns exactly 0 whatever lies beyond it, shader.rs:95-125; csrc/sol_path.h shade_vertex), the reference
   * traces on. rays - live_rays = searches whose result the reference multiplies by zero. */
  uint64_t live_rays;
} OrcStats;

/* Renders samples [first, first+n) of the pixels in [x0,x1) x [y0,y1) (output coordinates, row 0 = top) and ADDS
 * the per-pixel sums to out (W*H*3 doubles, row-major, row 0 = top). threads <= 0: all hardware threads.
 * Returns 0, -1 bad input, -2 no light. */
int orc_render(const SolSceneDesc* d, int real_kind, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t first,
               uint32_t n, uint64_t seed, int threads, double* out, OrcStats* stats);

void orc_vec3_ops(const double a[3], const double b[3], double out[16]);
void orc_vec3_reflect(const double v[3], const double n[3], double out[3]);
void orc_vec3_refract(const double v[3], const double n[3], double ior, double out[3]);
void orc_vec3_unit(const double v[3], double out[3]);
void orc_ray_at(const double o[3], const double d[3], double t, double out[3]);
int orc_aabb_hit(const double box[6], const double o[3], const double d[3]);
void orc_onb_local(const double t[3], const double b[3], const double n[3], const double a[3], double out[3]);
void orc_transform_normal_by_map(const double rgb[3], const double t[3], const double b[3], const double n[3], double out[3]);
void orc_rgb_to_vec3(const uint8_t p[3], double out[3]);
void orc_to_rgb_color(const double col[3], uint32_t spp, uint8_t out[3]);
uint32_t orc_rng_bits(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t counter);
void orc_f32_funcs(float r, float x, float y, float out[5]);
int orc_debug_path(const SolSceneDesc* d, int real_kind, uint32_t x, uint32_t y, uint32_t sample, uint64_t seed, float* rows, uint32_t max_rows);
int orc_eval_f32(uint32_t fn, const float* in, uint32_t n, uint32_t in_stride, float* out, uint32_t out_stride);
int orc_closest_hit(const SolSceneDesc* d, int real_kind, const double o[3], const double dir[3], double* t_out, uint32_t* mat_out);

#ifdef __cplusplus
}
#endif
#endif
