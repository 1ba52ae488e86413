/*
 * solstrale_host.h -- C entry points of `libsolstrale_host.so`, the C++ host that stands in for the
 * reference's Rust host above the device C ABI (solstrale_hip.h). No Rust toolchain exists in this
 * environment; the C++ mirror (solstrale-rust_amd/host/solstrale.hpp) keeps the reference's names. These C
 * wrappers exist so the Python test/bench harness can build scenes with the same calls the reference's
 * tests/scenes.rs makes: every function cites the reference constructor it wraps.
 *
 * Objects are addressed by small integer ids inside a builder; -1 is "none". Functions return an id >= 0
 * (or 0 for void-like calls) on success and a negative value on error; solh_last_error() has the message.
 */
#ifndef SOLSTRALE_HOST_H
#define SOLSTRALE_HOST_H

#include <stdint.h>

#include "solstrale_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SolhBuilder SolhBuilder;

SolhBuilder* solh_builder_new(void);
void solh_builder_free(SolhBuilder* b);
const char* solh_last_error(void);

/* Transformations (src/geo/transformation.rs). A transform id names a list applied in order
 * (`Transformations::new(vec![..])`); id -1 = NopTransformer. kinds: 0 Translation(x,y,z), 1 RotationX(deg),
 * 2 RotationY(deg), 3 RotationZ(deg), 4 Scale(s). `params` holds 3 doubles per op. */
int solh_transform(SolhBuilder* b, int n_ops, const int* kinds, const double* params);

/* Textures (src/material/texture.rs): SolidColor::new, ImageMap::new, load_normal_texture (on decoded RGB8). */
int solh_solid_color(SolhBuilder* b, double r, double g, double bl);
int solh_image_map(SolhBuilder* b, uint32_t width, uint32_t height, const uint8_t* rgb8);
int solh_normal_texture(SolhBuilder* b, uint32_t width, uint32_t height, const uint8_t* rgb8);

/* Materials (src/material/mod.rs): Lambertian::new, Metal::new, Dielectric::new, DiffuseLight::new, Blend::new.
 * normal_tex -1 = None; attenuation_half_length NaN = None. */
int solh_lambertian(SolhBuilder* b, int albedo_tex, int normal_tex);
int solh_metal(SolhBuilder* b, int albedo_tex, int normal_tex, double fuzz);
int solh_dielectric(SolhBuilder* b, int albedo_tex, int normal_tex, double index_of_refraction);
int solh_diffuse_light(SolhBuilder* b, double r, double g, double bl, double attenuation_half_length);
int solh_blend(SolhBuilder* b, int material_1, int material_2, double blend_factor);

/* Hittables (src/hittable/ *.rs): Sphere::new, Quad::new, Quad::new_box (returns the first of 6 consecutive
 * ids), Triangle::new_with_tex_coords, ConstantMedium::new, Bvh::new. */
int solh_sphere(SolhBuilder* b, const double center[3], double radius, int material);
int solh_quad(SolhBuilder* b, const double q[3], const double u[3], const double v[3], int material, int transform);
int solh_box(SolhBuilder* b, const double a[3], const double bb[3], int material, int transform);
int solh_triangle(SolhBuilder* b, const double v0[3], const double v1[3], const double v2[3], const float uv[6],
                  int material, int transform);
/* Bulk form of solh_triangle for large meshes: vertices 9 doubles per triangle, uvs 6 floats per triangle
 * (NULL = all zero), materials one id per triangle. Returns the first of n consecutive ids. */
int solh_triangles(SolhBuilder* b, uint32_t n, const double* vertices, const float* uvs, const int* materials,
                   int transform);
int solh_spheres(SolhBuilder* b, uint32_t n, const double* centers, const double* radii, const int* materials);
int solh_constant_medium(SolhBuilder* b, int boundary, double density, const double color[3]);
int solh_bvh(SolhBuilder* b, int n, const int* hittables);
/* Bvh::new over the id range [first, first+n) */
int solh_bvh_range(SolhBuilder* b, int first, int n);

/* Scene + RenderConfig + CameraConfig (src/renderer/mod.rs:26-72, src/camera.rs:8-31); flattens the tree
 * (Camera::new included) and returns the description handed to sol_scene_create. The pointer stays valid
 * until the next solh_finish on this builder or solh_builder_free. NULL on error. */
const SolSceneDesc* solh_finish(SolhBuilder* b, int world, uint32_t width, uint32_t height, uint32_t shader_kind,
                                uint32_t max_depth, const double background[3], double vertical_fov_degrees,
                                double aperture_size, const double look_from[3], const double look_at[3],
                                const double up[3]);
/* deepest Bvh nesting of the flattened tree (information for the device stack) */
/* EXTENSION, not in the reference: a latitude-longitude environment map (width * height * 3 floats, linear radiance, row 0 =
 * up) that rays which hit nothing return, scaled, instead of Scene.background_color (SolSceneDesc.env_*). Before solh_finish. */
int solh_environment(SolhBuilder* b, uint32_t width, uint32_t height, const float* rgb, double scale);
uint32_t solh_tree_depth(const SolhBuilder* b);

/* ray_trace (src/lib.rs:93-99) on the scene of the last solh_finish: samples_per_pixel passes, progress
 * callback per sample index (image pointer non-NULL when the strategy produced one; RGB8, row 0 top),
 * abort callback polled between batches (may be NULL). strategy: 0 EverySample, 1 Interval(seconds),
 * 2 OnlyFinal. Returns 0, or negative with solh_last_error() = the reference's error string. */
/* Obj::new(path, filename).load(transformation, default_material) (src/loader/obj.rs:29-136): returns the hittable id of the
 * Bvh of the model's triangles. default_material < 0 = None (white Lambertian). Image files named by the MTL are decoded
 * by `decoder` (the reference uses the `image` crate): it returns 0 and a W*H*3 RGB8 buffer that stays valid until its next
 * call, 1 when the file cannot be opened, 2 when it cannot be decoded. Errors carry the reference's strings:
 * "failed to load obj model from ..", "failed to load MTL file for ..", "Failed to open image texture ..: ..". */
typedef int (*solh_image_decoder)(void* user, const char* path, uint32_t* width, uint32_t* height, const uint8_t** rgb8);
int solh_load_obj(SolhBuilder* b, const char* path, const char* filename, int transform, int default_material,
                  solh_image_decoder decoder, void* user);

/* RenderConfig::post_processors (src/renderer/mod.rs:35) for the following solh_ray_trace calls: kinds[i] 0 = NopPostProcessor,
 * 1 = BloomPostProcessor with params[3i..3i+2] = kernel_size_fraction, threshold, max_intensity (NaN = None); n = 0: no image
 * is produced. The default is one NopPostProcessor. Errors carry the reference's strings
 * ("kernel_size_fraction must be between 0 and 0.5"). */
int solh_set_post_processors(SolhBuilder* b, int n, const int* kinds, const double* params);

typedef void (*solh_progress_fn)(void* user, double progress, double fps, double eta_seconds,
                                 const uint8_t* image_rgb8, uint32_t width, uint32_t height);
typedef int (*solh_abort_fn)(void* user);
int solh_ray_trace(SolhBuilder* b, uint32_t samples_per_pixel, uint64_t seed, int strategy, double interval_seconds,
                   int device, solh_progress_fn progress, solh_abort_fn abort_cb, void* user);
/* The same from one process on n_devices GPUs of the node (extension: the reference has no devices): the frame's 8x8 blocks are dealt out over
 * `devices` (by their cost in the creation probe); every device holds the scene and renders its blocks of each batch, images are gathered into devices[0]
 * by peer copies (sol_gather_local) and post-processed there. The picture does not depend on n_devices. A device may be named more than once. */
int solh_ray_trace_devices(SolhBuilder* b, uint32_t samples_per_pixel, uint64_t seed, int strategy, double interval_seconds,
                           int n_devices, const int* devices, solh_progress_fn progress, solh_abort_fn abort_cb, void* user);

/* sizeof of every POD struct of solstrale_hip.h in declaration order (SolAabb, SolBvhNode, SolSphere, SolQuad,
 * SolTriangle, SolMedium, SolMaterial, SolTexture, SolCamera, SolSceneDesc, SolStats): lets a foreign-language
 * binding assert that its mirror of the structs has the same layout. */
void solh_abi_sizes(uint32_t out[11]);

/* src/util/rgb_color.rs:14-35 host arithmetic (KATs) */
void solh_to_rgb_color(const double col[3], uint32_t samples_per_pixel, uint8_t out[3]);

#ifdef __cplusplus
}
#endif
#endif
