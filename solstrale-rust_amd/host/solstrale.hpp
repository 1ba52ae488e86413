// solstrale.hpp -- C++ host mirror of the reference's public surface for the accelerated path.
//
// The reference host is Rust (crate `solstrale`); no Rust toolchain exists in this environment, so the host
// side above the C ABI (include/solstrale_hip.h) is written in C++ with the SAME names, argument meaning and
// error behaviour as the reference's operator interface:
//   Scene / RenderConfig / RenderProgress / RenderImageStrategy   (src/renderer/mod.rs:26-118)
//   CameraConfig / Camera::new                                    (src/camera.rs:8-74)
//   Sphere::new, Quad::new, Quad::new_box, Triangle::new[_with_tex_coords], ConstantMedium::new, Bvh::new
//                                                                 (src/hittable/*.rs)
//   Lambertian / Metal / Dielectric / DiffuseLight / Blend        (src/material/mod.rs)
//   SolidColor / ImageMap                                         (src/material/texture.rs)
//   Transformer impls                                             (src/geo/transformation.rs)
//   ray_trace(scene, output, abort)                               (src/lib.rs:93-99)
// What stays on the host exactly as in the reference: scene construction, the BVH builder
// (src/hittable/bvh.rs:61-162), Camera::new, the pass loop with progress/abort, the Nop post-processor.
// What is handed to the GPU: the flattened tree (flatten()) through sol_scene_create / sol_render / sol_read.
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../../include/solstrale_hip.h"

namespace solstrale {

// ---- geo (src/geo/vec3.rs, src/geo/mod.rs, src/util/interval.rs) -------------------------------------
struct Vec3 {
  double x = 0, y = 0, z = 0;
  Vec3() = default;
  Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
  Vec3 operator+(const Vec3& v) const { return {x + v.x, y + v.y, z + v.z}; }
  Vec3 operator-(const Vec3& v) const { return {x - v.x, y - v.y, z - v.z}; }
  Vec3 operator*(double t) const { return {x * t, y * t, z * t}; }
  Vec3 operator/(double t) const { return {x / t, y / t, z / t}; }
  Vec3 neg() const { return {-x, -y, -z}; }
  double dot(const Vec3& v) const { return x * v.x + y * v.y + z * v.z; }
  Vec3 cross(const Vec3& v) const { return {y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x}; }
  double length_squared() const { return x * x + y * y + z * z; }
  double length() const { return std::sqrt(length_squared()); }
  Vec3 unit() const { return *this / length(); }
  double axis(int a) const { return a == 0 ? x : (a == 1 ? y : z); }
};

struct Uv {
  float u = 0, v = 0;
  Uv() = default;
  Uv(float u_, float v_) : u(u_), v(v_) {}
  Uv operator-(const Uv& r) const { return {u - r.u, v - r.v}; }
};

struct Interval {
  double min = std::numeric_limits<double>::infinity();  // EMPTY_INTERVAL (interval.rs:15-18)
  double max = -std::numeric_limits<double>::infinity();
  double size() const { return max - min; }
  Interval expand(double delta) const { return {min - delta / 2., max + delta / 2.}; }
};
inline Interval combine_intervals(const Interval& a, const Interval& b) {
  return {std::fmin(a.min, b.min), std::fmax(a.max, b.max)};
}

struct Aabb {
  Interval x, y, z;
  static Aabb new_from_2_points(const Vec3& a, const Vec3& b);
  static Aabb new_from_3_points(const Vec3& a, const Vec3& b, const Vec3& c);
  Aabb combine(const Aabb& a) const { return {combine_intervals(x, a.x), combine_intervals(y, a.y), combine_intervals(z, a.z)}; }
  Aabb pad_if_needed() const;
  Vec3 center() const { return {(x.min + x.max) * 0.5, (y.min + y.max) * 0.5, (z.min + z.max) * 0.5}; }
};

// ---- transformations (src/geo/transformation.rs) ------------------------------------------------------
struct Transformer {
  virtual ~Transformer() = default;
  virtual Vec3 transform(const Vec3& v, bool skip_translation) const = 0;
};
struct NopTransformer : Transformer {
  Vec3 transform(const Vec3& v, bool) const override { return v; }
};
struct Translation : Transformer {
  Vec3 translation;
  explicit Translation(const Vec3& t) : translation(t) {}
  Vec3 transform(const Vec3& v, bool skip) const override { return skip ? v : v + translation; }
};
struct RotationX : Transformer {
  double sin_theta, cos_theta;
  explicit RotationX(double angle_degrees);
  Vec3 transform(const Vec3& v, bool) const override {
    return {v.x, cos_theta * v.y + sin_theta * v.z, -sin_theta * v.y + cos_theta * v.z};
  }
};
struct RotationY : Transformer {
  double sin_theta, cos_theta;
  explicit RotationY(double angle_degrees);
  Vec3 transform(const Vec3& v, bool) const override {
    return {cos_theta * v.x + sin_theta * v.z, v.y, -sin_theta * v.x + cos_theta * v.z};
  }
};
struct RotationZ : Transformer {
  double sin_theta, cos_theta;
  explicit RotationZ(double angle_degrees);
  Vec3 transform(const Vec3& v, bool) const override {
    return {cos_theta * v.x + sin_theta * v.y, -sin_theta * v.x + cos_theta * v.y, v.z};
  }
};
struct Scale : Transformer {
  double scale;
  explicit Scale(double s) : scale(s) {}
  Vec3 transform(const Vec3& v, bool) const override { return v * scale; }
};
struct Transformations : Transformer {
  std::vector<std::shared_ptr<Transformer>> transformations;
  explicit Transformations(std::vector<std::shared_ptr<Transformer>> t) : transformations(std::move(t)) {}
  Vec3 transform(const Vec3& v, bool skip) const override {
    Vec3 r = v;
    for (auto& t : transformations) r = t->transform(r, skip);
    return r;
  }
};

// ---- textures (src/material/texture.rs) ---------------------------------------------------------------
struct RgbImage {
  uint32_t width = 0, height = 0;
  std::vector<uint8_t> data;  // RGB8, row-major, row 0 = top (image crate convention)
};
struct Texture;
using Textures = std::shared_ptr<const Texture>;
struct Texture {
  int kind = SOL_TEX_SOLID;
  Vec3 color;                             // SolidColor(Vec3)
  std::shared_ptr<const RgbImage> image;  // ImageMap{image,..}
};
struct SolidColor {
  static Textures create(double r, double g, double b);
  static Textures new_from_vec3(const Vec3& c) { return create(c.x, c.y, c.z); }
};
struct ImageMap {
  static Textures create(std::shared_ptr<const RgbImage> image);
};
// load_normal_texture's decision + conversion on an already decoded bump map (texture.rs:53-97,
// util/height_map.rs:68-86). File decoding stays with the caller.
Textures normal_texture_from_bump_map(std::shared_ptr<const RgbImage> image);

// ---- materials (src/material/mod.rs) ------------------------------------------------------------------
struct Material;
using Materials = std::shared_ptr<const Material>;
struct Material {
  int kind = SOL_MAT_LAMBERTIAN;
  Textures albedo;
  Textures normal;  // Option<Textures>
  double param = 0;
  Materials m1, m2;
  bool is_light() const { return kind == SOL_MAT_DIFFUSE_LIGHT; }  // mod.rs:100-102,353-355
};
struct Lambertian { static Materials create(Textures albedo, Textures normal = nullptr); };
struct Metal { static Materials create(Textures albedo, Textures normal, double fuzz); };
struct Dielectric { static Materials create(Textures albedo, Textures normal, double index_of_refraction); };
struct DiffuseLight {
  // attenuation_half_length < 0 or NaN = None (mod.rs:335-340)
  static Materials create(double r, double g, double b, double attenuation_half_length = std::numeric_limits<double>::quiet_NaN());
};
struct Blend { static Materials create(Materials m1, Materials m2, double blend_factor); };

// ---- hittables (src/hittable/*.rs) --------------------------------------------------------------------
struct Hittable;
using Hittables = std::shared_ptr<const Hittable>;
struct BvhItem {
  int kind = 0;  // 0 None, 1 Node, 2 Leaf
  Hittables item;
};
struct Hittable {
  uint32_t kind = SOL_REF_NONE;  // SOL_REF_SPHERE/QUAD/TRIANGLE/MEDIUM, SOL_REF_NODE = Bvh
  Aabb b_box;
  Materials mat;
  // sphere
  Vec3 center; double radius = 0;
  // quad
  Vec3 q, u, v, normal, w; double d = 0, area = 0;
  // triangle
  Vec3 v0, v0v1, v0v2, tangent, bi_tangent; Uv uv0, uv1, uv2;
  // constant medium
  Hittables boundary; double negative_inverse_density = 0;
  // bvh
  BvhItem left, right;
};
struct Sphere { static Hittables create(const Vec3& center, double radius, Materials mat); };
struct Quad {
  static Hittables create(const Vec3& q, const Vec3& u, const Vec3& v, Materials mat, const Transformer& t);
  static std::vector<Hittables> new_box(const Vec3& a, const Vec3& b, Materials mat, const Transformer& t);
};
struct Triangle {
  static Hittables create(const Vec3& v0, const Vec3& v1, const Vec3& v2, Materials mat, const Transformer& t);
  static Hittables new_with_tex_coords(const Vec3& v0, const Vec3& v1, const Vec3& v2, Uv uv0, Uv uv1, Uv uv2,
                                       Materials mat, const Transformer& t);
};
struct ConstantMedium { static Hittables create(Hittables boundary, double density, const Vec3& color); };
struct Bvh {
  // Bvh::new (bvh.rs:61-73): midpoint split on the axis of largest centroid spread, median fallback,
  // one primitive per leaf; children built in parallel.
  static Hittables create(std::vector<Hittables> list);
};
std::vector<Hittables> get_lights(const Hittables& h);  // Hittable::get_lights, depth-first

// ---- loader (src/loader/obj.rs) -------------------------------------------------------------------------
// File decoding (the `image` crate in the reference) is supplied by the caller: decode(path, what) returns the RGB8 image
// or throws std::runtime_error carrying the reference's message, "Failed to open|load|decode {what} texture {path}: .."
// with what = "image" (ImageMap::load, texture.rs:137-153) or "bump" (load_bump_map, texture.rs:53-65).
using ImageDecoder = std::function<std::shared_ptr<const RgbImage>(const std::string& path, const char* what)>;
struct Obj {
  std::string path, filename;
  Obj(std::string p, std::string f) : path(std::move(p)), filename(std::move(f)) {}  // Obj::new (obj.rs:29-34)
  // Loader::load (obj.rs:38-136): every material Lambertian (Kd / map_Kd, map_bump with normal-vs-height detection),
  // `default_material` (nullptr = white Lambertian) for faces without one, all triangles in one Bvh. Throws
  // std::runtime_error("failed to load obj model from {path}{filename}") / ("failed to load MTL file for ..").
  Hittables load(const Transformer& transformation, Materials default_material, const ImageDecoder& decode) const;
};

// ---- camera / render config (src/camera.rs, src/renderer/mod.rs) -------------------------------------
struct CameraConfig {
  double vertical_fov_degrees = 50.0;
  double aperture_size = 0.0;
  Vec3 look_from, look_at, up{0., 1., 0.};
};
SolCamera camera_new(size_t image_width, size_t image_height, const CameraConfig& c);  // Camera::new

struct Shaders {
  uint32_t kind = SOL_SHADER_PATH_TRACING;
  uint32_t max_depth = 50;
};
struct PathTracingShader { static Shaders create(uint32_t max_depth) { return {SOL_SHADER_PATH_TRACING, max_depth}; } };
struct AlbedoShader { static Shaders create() { return {SOL_SHADER_ALBEDO, 0}; } };
struct NormalShader { static Shaders create() { return {SOL_SHADER_NORMAL, 0}; } };
struct SimpleShader { static Shaders create() { return {SOL_SHADER_SIMPLE, 0}; } };

struct RenderImageStrategy {
  enum Kind { EverySample, Interval, OnlyFinal } kind = OnlyFinal;
  double interval_seconds = 0;
  bool should_generate_image(uint32_t sample, uint32_t total, double now_s, double last_s) const;
};

// src/post/mod.rs:45-55. Post-processors run on the device (sol_tonemap_rgb8, sol_bloom*); OidnPostProcessor is not built
// (a third-party denoiser binary, out of scope).
struct PostProcessors {
  enum Kind { Nop, Bloom } kind = Nop;
  double kernel_size_fraction = 0, threshold = 0, max_intensity = 0;  // Bloom
};
struct NopPostProcessor { static PostProcessors create() { return {}; } };  // src/post/nop.rs:11-17
// src/post/oidn.rs:85-128: without the crate's optional `oidn-postprocessor` feature (its default build) the OIDN
// post-processor IS the Nop post-processor and asks for no albedo/normal buffers; that is the behaviour mirrored here.
struct OidnPostProcessor { static PostProcessors create() { return {}; } };
struct BloomPostProcessor {
  // src/post/bloom.rs:27-47: throws std::invalid_argument("kernel_size_fraction must be between 0 and 0.5");
  // a NaN threshold / max_intensity means None (defaults |(1,1,1)| and f64::MAX)
  static PostProcessors create(double kernel_size_fraction, double threshold = std::nan(""), double max_intensity = std::nan(""));
};

struct RenderConfig {
  size_t width = 300, height = 200;
  uint32_t samples_per_pixel = 50;
  Shaders shader = PathTracingShader::create(50);
  std::vector<PostProcessors> post_processors{NopPostProcessor::create()};  // src/renderer/mod.rs:35,49
  RenderImageStrategy render_image_strategy;
  uint64_t seed = 0x5017A1Eull;  // the reference has no seed (entropy-seeded fastrand); the build adds one
};

struct Scene {
  Hittables world;
  CameraConfig camera;
  Vec3 background_color;
  RenderConfig render_config;
  // EXTENSION (not in the reference): latitude-longitude environment of linear radiance, row 0 = up; empty = none
  std::vector<float> environment;
  uint32_t env_width = 0, env_height = 0;
  double env_scale = 1.0;
};

struct RenderProgress {
  double progress = 0;
  double fps = 0;
  double estimated_time_left_s = 0;
  bool has_image = false;
  uint32_t width = 0, height = 0;
  std::vector<uint8_t> render_image;  // RGB8
};

// ---- flattening for the C ABI --------------------------------------------------------------------------
struct FlatScene {
  std::vector<SolBvhNode> nodes;
  std::vector<SolSphere> spheres;
  std::vector<SolQuad> quads;
  std::vector<SolTriangle> triangles;
  std::vector<SolMedium> mediums;
  std::vector<SolMaterial> materials;
  std::vector<SolTexture> textures;
  std::vector<uint8_t> texels;
  std::vector<uint32_t> lights;
  uint32_t max_depth_nodes = 0;  // deepest node nesting, for information
  SolSceneDesc desc{};           // views into the vectors above
};
// Walks `scene.world` depth-first: inlines nested Bvh leaves, dedups materials/textures by identity,
// assigns dfs_index, collects lights in get_lights order. Throws std::runtime_error on malformed input.
std::unique_ptr<FlatScene> flatten(const Scene& scene);

// ray_trace (src/lib.rs:93-99): Renderer::new (light check) then the pass loop of Renderer::render, with the
// row tasks replaced by sol_render. `output` receives one RenderProgress per sample index; `abort` is polled
// between GPU batches. Returns "" on success (also when aborted), else the error string
// (e.g. "Scene should have at least one light").
std::string ray_trace(const Scene& scene, const std::function<void(RenderProgress&&)>& output,
                      const std::function<bool()>& abort, int device = 0);
// The same from ONE process on several GPUs of the node (extension; the picture does not depend on how many): the frame's 8x8 blocks are dealt out
// over `devices`, each holds the scene and renders its blocks of every batch; images are gathered into devices[0] by peer copies (sol_gather_local).
std::string ray_trace(const Scene& scene, const std::function<void(RenderProgress&&)>& output,
                      const std::function<bool()>& abort, const std::vector<int>& devices);

// src/util/rgb_color.rs:14-35 (host-side reference arithmetic for the Nop post-processor)
void to_rgb_color(const double col[3], uint32_t samples_per_pixel, uint8_t out[3]);

}  // namespace solstrale
