// oracle.cpp -- CPU restatement of the reference's per-sample path (TEST INFRASTRUCTURE, not product).
//
// PURPOSE. This file is the parity oracle for the HIP path: a plain C++ restatement of what
// DanielPettersson/Solstrale-Rust computes under `Renderer::ray_color` (src/renderer/mod.rs:164-206), in the
// reference's own structure: recursive ray_color <-> PathTracingShader::shade, depth-first BVH search in the
// fixed left->right order WITHOUT t culling at boxes (src/hittable/bvh.rs:165-180), every formula in the order
// the reference writes it. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
// the product (solstrale-rust_amd/) never includes, links or calls anything in oracle/.
//
// It consumes the same flattened description the device library gets (include/solstrale_hip.h) and is
// instantiated twice:
//   Real = double : the reference's arithmetic (f64, with the reference's own f32 spots: Uv, barycentrics);
//                   used for the golden-image checks and as the timed CPU baseline.
//   Real = float  : the same algorithm in fp32 with the spec'd fp32 elementary functions of DESIGN.md
//                   ("fp32 arithmetic contract"); the fixed-seed parity target of the GPU kernels.
//
// PINNING. The reference cannot be built here (Rust, no toolchain). The restatement is pinned by
// (i) the reference's exact known-answer tests (tests/test_oracle_kat.py: vec3 doc-tests, Aabb, Interval,
// transform_normal_by_map, rgb_to_vec3, to_rgb_color, transformations), and (ii) the reference's golden images
// under its own similarity criterion (tests/test_oracle_golden.py, fixtures in tests/golden/).
// Deviations from the reference, all forced by the north star and listed in DESIGN.md:
//   * src/random.rs (fastrand, entropy seeded) is replaced by the counter-based generator below;
//   * Blend's normal-choice draw (src/material/mod.rs:438-444) and the normal-map fetch are applied to the
//     closest hit only instead of every candidate (independent draws: same distribution);
//   * ConstantMedium's draws come from a sub-stream keyed by (path, depth, medium) so that they do not depend on
//     the order in which the tree is searched;
//   * the FLOAT instantiation alone carries the eight fp32-only rules of DESIGN.md 4 (`sizeof(R) == 4` branches: box pad, sphere
//     roots in the sphere's box, needle triangles, rotated triangle records, sphere / quad hit points back on their surface, the
//     cancellation-free sphere test, no hit on the flat primitive a ray leaves); the double instantiation is the reference's lines.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off: no FMA contraction, so float results are the plain
// IEEE sequence the device code reproduces).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/solstrale_hip.h"
#include "oracle.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// Counter-based RNG (replaces src/random.rs). Spec in DESIGN.md "RNG".
// ---------------------------------------------------------------------------------------------------------
inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
  return x;
}
struct Rng {
  uint32_t k0, k1, ctr;
  void init(uint64_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    k0 = mix32(mix32(pixel ^ lo) + sample);
    k1 = mix32(mix32(sample ^ hi ^ 0x9E3779B9u) + pixel);
    ctr = 0;
  }
  static uint32_t bits(uint32_t k0, uint32_t k1, uint32_t c) { return mix32(mix32(k0 + c * 0x9E3779B9u) ^ k1); }
  uint32_t next_u32() { return bits(k0, k1, ctr++); }
  uint32_t at(uint32_t c) const { return bits(k0, k1, c); }
};
template <typename R> inline R u32_to_unit(uint32_t x) { return (R)(x >> 8) * (R)(1.0 / 16777216.0); }

// ---------------------------------------------------------------------------------------------------------
// Elementary functions. double: libm (the reference's f64 arithmetic). float: the fp32 contract of DESIGN.md,
// fixed polynomials evaluated with separate mul/add in the written order (so CPU and GPU agree bit for bit).
// ---------------------------------------------------------------------------------------------------------
const double PI_D = 3.14159265358979323846;

inline void sincos2pi(double r, double& c, double& s) { double phi = 2. * PI_D * r; c = std::cos(phi); s = std::sin(phi); }
inline void sincos2pi(float r, float& c, float& s) {
  float t = r * 4.0f;
  float j = std::floor(t + 0.5f);
  float f = t - j;
  float x = f * 1.57079632679489661923f;
  float z = x * x;
  float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
  float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  int q = ((int)j) & 3;
  switch (q) {
    case 0: c = cp; s = sp; break;
    case 1: c = -sp; s = cp; break;
    case 2: c = -cp; s = -sp; break;
    default: c = sp; s = -cp; break;
  }
}
inline double acos_r(double x) { return std::acos(x); }
inline float acos_r(float x) {
  float a = std::fabs(x);
  float p = -0.0012624911f;
  p = p * a + 0.0066700901f;
  p = p * a - 0.0170881256f;
  p = p * a + 0.0308918810f;
  p = p * a - 0.0501743046f;
  p = p * a + 0.0889789874f;
  p = p * a - 0.2145988016f;
  p = p * a + 1.5707963050f;
  float r = std::sqrt(1.0f - a) * p;
  return x < 0.0f ? 3.14159265358979323846f - r : r;
}
inline double atan2_r(double y, double x) { return std::atan2(y, x); }
inline float atan2_r(float y, float x) {
  float ax = std::fabs(x), ay = std::fabs(y);
  float mx = std::fmax(ax, ay), mn = std::fmin(ax, ay);
  if (mx == 0.0f) return 0.0f;
  float a = mn / mx;
  float off = 0.0f;
  if (a > 0.4142135623730950f) { off = 0.78539816339744831f; a = (a - 1.0f) / (a + 1.0f); }
  float z = a * a;
  float r = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * a + a;
  r = off + r;
  if (ay > ax) r = 1.57079632679489661923f - r;
  if (x < 0.0f) r = 3.14159265358979323846f - r;
  return y < 0.0f ? -r : r;
}
inline double log_r(double x) { return std::log(x); }
inline float log_r(float x) {
  if (x <= 0.0f) return -std::numeric_limits<float>::infinity();
  uint32_t b; std::memcpy(&b, &x, 4);
  int e = (int)(b >> 23) - 126;            // x = m * 2^e, m in [0.5,1)   (denormals do not occur: x >= 2^-24)
  b = (b & 0x007FFFFFu) | 0x3F000000u;
  float m; std::memcpy(&m, &b, 4);
  if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float y = 7.0376836292e-2f;
  y = y * m - 1.1514610310e-1f;
  y = y * m + 1.1676998740e-1f;
  y = y * m - 1.2420140846e-1f;
  y = y * m + 1.4249322787e-1f;
  y = y * m - 1.6668057665e-1f;
  y = y * m + 2.0000714765e-1f;
  y = y * m - 2.4999993993e-1f;
  y = y * m + 3.3333331174e-1f;
  y = y * m * z;
  float fe = (float)e;
  y = y + fe * -2.12194440e-4f;
  y = y - 0.5f * z;
  float r = m + y;
  r = r + fe * 0.693359375f;
  return r;
}

template <typename R> struct Consts;
template <> struct Consts<double> { static constexpr double pi = 3.14159265358979323846; };
template <> struct Consts<float> { static constexpr float pi = 3.14159265358979323846f; };

// ---------------------------------------------------------------------------------------------------------
// geo (src/geo/vec3.rs, src/geo/mod.rs, src/util/interval.rs)
// ---------------------------------------------------------------------------------------------------------
template <typename R> struct V3 {
  R x, y, z;
  V3 operator+(const V3& v) const { return {x + v.x, y + v.y, z + v.z}; }
  V3 operator-(const V3& v) const { return {x - v.x, y - v.y, z - v.z}; }
  V3 operator*(const V3& v) const { return {x * v.x, y * v.y, z * v.z}; }
  V3 operator*(R t) const { return {x * t, y * t, z * t}; }
  V3 operator/(R t) const { return {x / t, y / t, z / t}; }
  V3 neg() const { return {-x, -y, -z}; }
  R dot(const V3& v) const { return x * v.x + y * v.y + z * v.z; }                                      // vec3.rs:227
  V3 cross(const V3& v) const { return {y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x}; }     // vec3.rs:238
  R length_squared() const { return x * x + y * y + z * z; }
  R length() const { return std::sqrt(length_squared()); }
  V3 unit() const { return *this / length(); }                                                           // vec3.rs:287
  V3 reflect(const V3& n) const { return *this - n * (dot(n) * (R)2); }                                  // vec3.rs:329
  V3 refract(const V3& n, R ior) const {                                                                 // vec3.rs:341-346
    R cos_theta = std::fmin(neg().dot(n), (R)1);
    V3 perp = (n * cos_theta + *this) * ior;
    V3 par = n * (-std::sqrt(std::fabs((R)1 - perp.length_squared())));
    return perp + par;
  }
};

const double ALMOST_ZERO = 1e-8;  // vec3.rs:21
const double RAY_MIN = 0.001;     // RAY_INTERVAL (interval.rs:25-28)

template <typename R> struct Ray {
  V3<R> origin, direction, inv;
  uint32_t from = 0;  // (fp32 rule 8, float instantiation only) the quad / triangle this ray leaves, 0 = none
  static Ray make(const V3<R>& o, const V3<R>& d) { return {o, d, {(R)1 / d.x, (R)1 / d.y, (R)1 / d.z}, 0u}; }  // geo/mod.rs:277-285
  V3<R> at(R t) const { return origin + direction * t; }
};

template <typename R> struct Onb {
  V3<R> tangent, bi_tangent, normal;
  static Onb make(const V3<R>& w) {  // geo/mod.rs:245-257
    V3<R> uw = w.unit();
    V3<R> a = std::fabs(uw.x) > (R)0.9 ? V3<R>{0, 1, 0} : V3<R>{1, 0, 0};
    V3<R> v = uw.cross(a).unit();
    V3<R> u = uw.cross(v);
    return {u, v, uw};
  }
  V3<R> local(const V3<R>& a) const { return tangent * a.x + bi_tangent * a.y + normal * a.z; }  // geo/mod.rs:260-262
};

struct UvF { float u, v; };

// Aabb::hit (geo/mod.rs:159-188): slab test over [0, inf); Rust's f64::max/min return the non-NaN operand.
template <typename R> inline bool aabb_hit(const R b[6], const Ray<R>& r) {
  R t_min = 0, t_max = std::numeric_limits<R>::infinity();
  const R o[3] = {r.origin.x, r.origin.y, r.origin.z};
  const R inv[3] = {r.inv.x, r.inv.y, r.inv.z};
  for (int a = 0; a < 3; ++a) {
    R lo = b[2 * a], hi = b[2 * a + 1];
    if (std::signbit(inv[a])) {
      t_min = std::fmax((hi - o[a]) * inv[a], t_min);
      t_max = std::fmin((lo - o[a]) * inv[a], t_max);
    } else {
      t_min = std::fmax((lo - o[a]) * inv[a], t_min);
      t_max = std::fmin((hi - o[a]) * inv[a], t_max);
    }
  }
  return t_min < t_max;
}

// ---------------------------------------------------------------------------------------------------------
// Scene converted to Real (plain casts of the f64 description; the device upload does the same)
// ---------------------------------------------------------------------------------------------------------
template <typename R> struct Scene {
  struct Node { R box[6]; uint32_t left, right; };
  struct Sph { V3<R> center; R radius; int mat; uint32_t dfs; };
  struct Qd { V3<R> q, u, v, normal, w; R d, area; int mat; uint32_t dfs; };
  struct Tri { V3<R> v0, e1, e2, normal, tangent, bi_tangent; R area; UvF uv0, uv1, uv2; int mat; uint32_t dfs; };
  struct Med { uint32_t boundary; int mat; R nid; uint32_t dfs; };
  struct Mat { int kind, albedo, normal, m1, m2; R param; bool param_none; };
  struct Tex { int kind; uint32_t w, h; uint64_t off; V3<R> rgb; };
  std::vector<Node> nodes; std::vector<Sph> spheres; std::vector<Qd> quads; std::vector<Tri> tris;
  std::vector<Med> meds; std::vector<Mat> mats; std::vector<Tex> texs; const uint8_t* texels = nullptr;
  std::vector<uint32_t> lights;
  struct Frame { V3<R> v0, e1, e2; };
  std::vector<Frame> light_frames;  // by light index: a triangle light's (v0, v0v1, v0v2) in the REFERENCE's vertex order (what random_direction samples)
  uint32_t root = 0, width = 0, height = 0, shader = 0, max_depth = 0;
  V3<R> background;
  const float* env = nullptr; uint32_t env_w = 0, env_h = 0; R env_scale = 1;  // EXTENSION (SolSceneDesc, abi_version >= 2)
  V3<R> cam_origin, cam_llc, cam_h, cam_v, cam_u, cam_vv; R lens_radius = 0;
  R box_pad = 0;  // fp32 box pad (0 in f64)
  R tri_delta = 0; // tolerance of the triangle consistency rule (strict_tri): 0.8 box pads
  bool strict_tri = false;  // fp32 only: the scene has needle triangles - fatter pad, consistency rule of hit_triangle (solstrale_hip.h, sol_scene_has_needles)

  static V3<R> cv(const double* p) { return {(R)p[0], (R)p[1], (R)p[2]}; }
  explicit Scene(const SolSceneDesc& d) {
    root = d.root; width = d.width; height = d.height; shader = d.shader_kind; max_depth = d.max_depth;
    background = cv(d.background);
    if (d.abi_version >= 2 && d.env_texels && d.env_width && d.env_height) { env = d.env_texels; env_w = d.env_width; env_h = d.env_height; env_scale = (R)d.env_scale; }
    cam_origin = cv(d.camera.origin); cam_llc = cv(d.camera.lower_left_corner); cam_h = cv(d.camera.horizontal);
    cam_v = cv(d.camera.vertical); cam_u = cv(d.camera.u); cam_vv = cv(d.camera.v); lens_radius = (R)d.camera.lens_radius;
    // fp32 box contract: cast, then pad outward by S * 2^-20, S = largest finite |coordinate| of the world's
    // box and the camera origin. The f64 instantiation keeps the reference's boxes untouched.
    R pad = 0;
    if (sizeof(R) == 4) {
      float S = 0.0f;
      auto take = [&](double v) { float a = std::fabs((float)v); if (std::isfinite(a) && a > S) S = a; };
      const uint32_t k = SOL_REF_KIND(d.root), i = SOL_REF_INDEX(d.root);
      const SolAabb* b = nullptr;
      if (k == SOL_REF_NODE && i < d.n_nodes) b = &d.nodes[i].bbox;
      else if (k == SOL_REF_SPHERE && i < d.n_spheres) b = &d.spheres[i].bbox;
      else if (k == SOL_REF_QUAD && i < d.n_quads) b = &d.quads[i].bbox;
      else if (k == SOL_REF_TRIANGLE && i < d.n_triangles) b = &d.triangles[i].bbox;
      else if (k == SOL_REF_MEDIUM && i < d.n_mediums) b = &d.mediums[i].bbox;
      if (b) for (int j = 0; j < 6; ++j) take(b->v[j]);
      for (int j = 0; j < 3; ++j) take(d.camera.origin[j]);
      pad = (R)(S * (1.0f / 1048576.0f));
      // (ORC_NO_NEEDLE_RULE: a diagnostic switch of this checker only - "plain fp32" for tests/tools/needle_bias.py, which measures what the
      // rule does to the image; the device has no such switch and the parity tests never set it)
      strict_tri = sol_scene_has_needles(&d) != 0 && !std::getenv("ORC_NO_NEEDLE_RULE");
      if (strict_tri) pad = (R)(S * (SOL_NEEDLE_PAD / 1048576.0f));  // (the consistency tolerance of hit_triangle is 0.8 of it)
    }
    box_pad = pad;
    tri_delta = pad * (R)0.8f;
    nodes.resize(d.n_nodes);
    for (uint32_t i = 0; i < d.n_nodes; ++i) {
      for (int k = 0; k < 6; k += 2) {
        nodes[i].box[k] = (R)d.nodes[i].bbox.v[k] - pad;
        nodes[i].box[k + 1] = (R)d.nodes[i].bbox.v[k + 1] + pad;
      }
      nodes[i].left = d.nodes[i].left; nodes[i].right = d.nodes[i].right;
    }
    spheres.resize(d.n_spheres);
    for (uint32_t i = 0; i < d.n_spheres; ++i) {
      const SolSphere& s = d.spheres[i];
      spheres[i] = {cv(s.center), (R)s.radius, s.material, s.dfs_index};
    }
    quads.resize(d.n_quads);
    for (uint32_t i = 0; i < d.n_quads; ++i) {
      const SolQuad& s = d.quads[i];
      quads[i] = {cv(s.q), cv(s.u), cv(s.v), cv(s.normal), cv(s.w), (R)s.d, (R)s.area, s.material, s.dfs_index};
    }
    tris.resize(d.n_triangles);
    for (uint32_t i = 0; i < d.n_triangles; ++i) {
      const SolTriangle& s = d.triangles[i];
      // fp32 contract (solstrale_hip.h, sol_triangle_rotation): the float record starts at the vertex opposite the longest edge; f64
      // keeps the reference's order (a triangle LIGHT is sampled in the reference's frame either way: light_frames below)
      const int k = sizeof(R) == 4 ? sol_triangle_rotation(&s) : 0;
      double v0[3], e1[3], e2[3];
      int uo[3];
      sol_triangle_rotated(&s, k, v0, e1, e2, uo);
      const float* uvs[3] = {s.uv0, s.uv1, s.uv2};
      tris[i] = {cv(v0), cv(e1), cv(e2), cv(s.normal), cv(s.tangent), cv(s.bi_tangent), (R)s.area,
                 {uvs[uo[0]][0], uvs[uo[0]][1]}, {uvs[uo[1]][0], uvs[uo[1]][1]}, {uvs[uo[2]][0], uvs[uo[2]][1]}, s.material, s.dfs_index};
    }
    meds.resize(d.n_mediums);
    for (uint32_t i = 0; i < d.n_mediums; ++i) meds[i] = {d.mediums[i].boundary, d.mediums[i].material, (R)d.mediums[i].negative_inverse_density, d.mediums[i].dfs_index};
    mats.resize(d.n_materials);
    for (uint32_t i = 0; i < d.n_materials; ++i) {
      const SolMaterial& m = d.materials[i];
      mats[i] = {m.kind, m.albedo_tex, m.normal_tex, m.m1, m.m2, (R)m.param, std::isnan(m.param)};
    }
    texs.resize(d.n_textures);
    for (uint32_t i = 0; i < d.n_textures; ++i) {
      const SolTexture& t = d.textures[i];
      texs[i] = {t.kind, t.width, t.height, t.texel_offset, cv(t.rgb)};
    }
    texels = d.texels;
    lights.assign(d.lights, d.lights + d.n_lights);
    light_frames.assign(d.n_lights, Frame{});
    for (uint32_t i = 0; i < d.n_lights; ++i)
      if (SOL_REF_KIND(d.lights[i]) == SOL_REF_TRIANGLE && SOL_REF_INDEX(d.lights[i]) < d.n_triangles) {
        const SolTriangle& s = d.triangles[SOL_REF_INDEX(d.lights[i])];
        light_frames[i] = {cv(s.v0), cv(s.v0v1), cv(s.v0v2)};
      }
  }
};

// Geometric part of a RayHit candidate (material/mod.rs:23-57 before get_transformed_normal)
template <typename R> struct Cand {
  R t; V3<R> p; Onb<R> onb; UvF uv; bool front; int mat;
  uint32_t ref = 0;  // the primitive that was hit (diagnostics)
};

struct Counters { uint64_t rays = 0, node_visits = 0, sphere_tests = 0, quad_tests = 0, tri_tests = 0, shades = 0, texels = 0, samples = 0, live_rays = 0; };

template <typename R> struct Tracer {
  const Scene<R>& sc;
  Counters cnt;
  Rng rng;
  uint32_t cur_depth = 0;  // depth of the ray being searched (medium sub-stream key)
  bool live = true;        // no level of the current path has multiplied by zero yet (OrcStats::live_rays)
  std::vector<float>* trace = nullptr;  // diagnostics: 12 floats per ray (orc_debug_path)
  R sphere_slack = 0;                   // half the fp32 box pad (float instantiation only)
  explicit Tracer(const Scene<R>& s) : sc(s) { sphere_slack = s.box_pad * (R)0.5; }

  R rnd() { return u32_to_unit<R>(rng.next_u32()); }                 // random_normal_float (random.rs:4-6)
  R rnd_range(R mn, R mx) { return rnd() * (mx - mn) + mn; }         // random_float (random.rs:9-11)
  uint32_t rnd_index(uint32_t n) { return (uint32_t)(((uint64_t)rng.next_u32() * n) >> 32); }  // random_element_index

  V3<R> random_in_unit_sphere() {  // vec3.rs:380-392
    for (;;) {
      V3<R> p; p.x = rnd_range(-1, 1); p.y = rnd_range(-1, 1); p.z = rnd_range(-1, 1);
      if (p.length_squared() < (R)1) return p;
    }
  }
  V3<R> random_in_unit_disc() {  // vec3.rs:400-412
    for (;;) {
      V3<R> p; p.x = rnd_range(-1, 1); p.y = rnd_range(-1, 1); p.z = 0;
      if (p.length_squared() < (R)1) return p;
    }
  }
  V3<R> random_cosine_direction() {  // vec3.rs:417-428
    R r1 = rnd(), r2 = rnd();
    R r2_sqrt = std::sqrt(r2);
    R c, s; sincos2pi(r1, c, s);
    return {c * r2_sqrt, s * r2_sqrt, std::sqrt((R)1 - r2)};
  }

  static bool contains(R mn, R mx, R x) { return mn <= x && x <= mx; }  // Interval::contains (interval.rs:67-69)

  // ---- primitive hits --------------------------------------------------------------------------------------
  // `in_a_search`: false for the one-primitive test of a light's pdf_value - no tree is involved there, so the needle rule (which exists
  // to make SEARCH results independent of the tree) does not apply; and it must not: a refused hit would make pdf_value 0 for a direction
  // that random_direction generates with a high density - the mixture estimator then weighs whatever lies behind the light with
  // 1 / (cosine pdf / 2) instead of next to nothing (a needle light seen edge-on: +0.3 % on the frame mean, found by comparing with f64).
  bool hit_triangle(const typename Scene<R>::Tri& T, const Ray<R>& r, R tmin, R tmax, Cand<R>& out, bool in_a_search = true) {  // triangle.rs:119-173
    cnt.tri_tests++;
    V3<R> p_vec = r.direction.cross(T.e2);
    R det = T.e1.dot(p_vec);
    if (std::fabs(det) < (R)ALMOST_ZERO) return false;
    R inv_det = (R)1 / det;
    V3<R> t_vec = r.origin - T.v0;
    V3<R> q_vec = t_vec.cross(T.e1);
    float u = (float)(t_vec.dot(p_vec) * inv_det);
    if (!(u >= 0.f && u <= 1.f)) return false;
    float v = (float)(r.direction.dot(q_vec) * inv_det);
    if (v < 0.f || u + v > 1.f) return false;
    R tt = T.e2.dot(q_vec) * inv_det;
    V3<R> intersection = r.at(tt);
    if (!contains(tmin, tmax, tt)) return false;
    if (sc.strict_tri && in_a_search) {
      // fp32 contract, scenes with needle triangles (solstrale_hip.h): the ray's point and the triangle's point of this hit must agree
      // within 0.8 box pads - or (t, u, v) are rounding noise, and whether the "hit" is seen would depend on the boxes around it.
      // (The device checks the CLOSEST hit of a search and searches again behind a failure: the closest of the valid candidates.)
      const V3<R> q = T.v0 + T.e1 * (R)u + T.e2 * (R)v;
      const V3<R> dl = intersection - q;
      if (!(std::fabs(dl.x) <= sc.tri_delta && std::fabs(dl.y) <= sc.tri_delta && std::fabs(dl.z) <= sc.tri_delta)) return false;
    }
    float uv0 = 1.f - u - v;
    UvF uv = {uv0 * T.uv0.u + u * T.uv1.u + v * T.uv2.u, uv0 * T.uv0.v + u * T.uv1.v + v * T.uv2.v};
    V3<R> normal = T.normal;
    bool front = r.direction.dot(normal) < (R)0;
    if (!front) normal = normal.neg();
    out = {tt, intersection, {T.tangent, T.bi_tangent, normal}, uv, front, T.mat};
    return true;
  }
  bool hit_quad(const typename Scene<R>::Qd& Q, const Ray<R>& r, R tmin, R tmax, Cand<R>& out) {  // quad.rs:150-194
    cnt.quad_tests++;
    R denom = Q.normal.dot(r.direction);
    if (std::fabs(denom) < (R)ALMOST_ZERO) return false;
    R t = (Q.d - Q.normal.dot(r.origin)) / denom;
    if (!contains(tmin, tmax, t)) return false;
    V3<R> hp = r.at(t);
    V3<R> planar = hp - Q.q;
    float u = (float)Q.w.dot(planar.cross(Q.v));
    float v = (float)Q.w.dot(Q.u.cross(planar));
    if (!(u >= 0.f && u <= 1.f) || !(v >= 0.f && v <= 1.f)) return false;
    bool front = r.direction.dot(Q.normal) < (R)0;
    V3<R> normal = front ? Q.normal : Q.normal.neg();
    // fp32 contract (seventh rule; the fifth rule's twin for a plane): the hit POINT of a quad is put back on the quad's plane. In single
    // precision `origin + t * direction` from a distant origin - the Cornell box's camera, 800 units away with |d| = 800: a sum of magnitude 1000 that ends near
    // 250 - lands up to ~1e-4 beside the plane, inside the box as often as outside; a scattered ray that leaves at a grazing angle from a point INSIDE re-hits the
    // same quad from behind at t > 0.001 and the path goes dark (C1: 4 samples in 10^5, a frame 6e-5 darker than f64, every difference of one sign; found by the
    // device-against-f64 gate of round 5). n . x = d is the quad's own plane (quad.rs:40-45), n a unit vector: one dot product and three FMAs put the point on it.
    // u, v, t and the normal stay as computed. f64: nothing changes.
    if (sizeof(R) == 4) hp = hp + Q.normal * (Q.d - Q.normal.dot(hp));
    out = {t, hp, {Q.u.unit(), Q.v.unit(), normal}, {u, v}, front, Q.mat};
    return true;
  }
  bool hit_sphere(const typename Scene<R>::Sph& S, const Ray<R>& r, R tmin, R tmax, Cand<R>& out) {  // sphere.rs:64-108
    cnt.sphere_tests++;
    V3<R> oc = r.origin - S.center;
    R a = r.direction.length_squared();
    R half_b = oc.dot(r.direction);
    R c = oc.length_squared() - S.radius * S.radius;
    // fp32 contract (sixth rule): in single precision the reference's discriminant half_b^2 - a c is the difference of two
    // numbers of the size a |oc|^2 - for an origin 800 units away it errs by 0.08, as much as r^2 - l^2 itself over the rim of a small sphere
    // (hits missed, misses hit), and the roots lose the same digits. The float instantiation takes the discriminant from the distance l of
    // the centre to the ray, a (r^2 - l^2) = half_b^2 - a c exactly, and the roots as q / a and c / q with q = -half_b -+ sqrt(..) of the sign
    // that does not cancel (Haines, Guenther, Akenine-Moeller, "Precision improvements for ray / sphere intersection", Ray Tracing Gems 2019,
    // ch. 7). Same roots, same order, to the last digits fp32 has. f64: the reference's lines, below.
    R root_first, root_second;
    if (sizeof(R) == 4) {
      const R k = half_b / a;
      const V3<R> l = oc - r.direction * k;
      const R disc1 = S.radius * S.radius - l.length_squared();
      if (disc1 < (R)0) return false;
      const R sq = std::sqrt(a * disc1);
      const R q = half_b >= (R)0 ? -half_b - sq : -half_b + sq;
      const R rq = q / a, rc = c / q;
      root_first = half_b >= (R)0 ? rq : rc;
      root_second = half_b >= (R)0 ? rc : rq;
    } else {
      R disc = half_b * half_b - a * c;
      if (disc < (R)0) return false;
      R sqrt_d = std::sqrt(disc);
      root_first = (-half_b - sqrt_d) / a;
      root_second = (-half_b + sqrt_d) / a;
    }
    // fp32 contract: in the float instantiation a root also has to put its hit point inside the sphere's own
    // box (centre +- radius, widened by half the box pad); the fp32 quadratic loses 7 digits for distant origins and would
    // otherwise report points that lie outside every box bounding the sphere. In f64 (the reference) nothing is added.
    auto root_ok = [&](R root) {
      if (!contains(tmin, tmax, root)) return false;
      if (sizeof(R) == 4) {
        const V3<R> p = r.at(root);
        const R lim = std::fabs(S.radius) + sphere_slack;  // (|r|: the reference knows a radius only through r^2 and its min/max box - a negative one is the hollow-glass idiom)
        return std::fabs(p.x - S.center.x) <= lim && std::fabs(p.y - S.center.y) <= lim && std::fabs(p.z - S.center.z) <= lim;
      }
      return true;
    };
    R root = root_first;
    if (!root_ok(root)) {
      root = root_second;
      if (!root_ok(root)) return false;
    }
    V3<R> hp = r.at(root);
    V3<R> n = hp - S.center;
    V3<R> normal = n.unit();
    // calculate_sphere_uv (sphere.rs:134-140)
    R theta = acos_r(-normal.y);
    R phi = -atan2_r(normal.z, normal.x) + Consts<R>::pi;
    UvF uv = {(float)(phi / ((R)2 * Consts<R>::pi)), (float)(theta / Consts<R>::pi)};
    V3<R> tangent = V3<R>{0, 1, 0}.cross(n).unit();
    V3<R> bi_tangent = n.cross(tangent);
    bool front = r.direction.dot(normal) < (R)0;
    if (!front) normal = normal.neg();
    // fp32 contract (fifth rule): the hit POINT of a sphere is put back on the sphere. The reference's quadratic in single
    // precision loses the digits of a distant origin - a camera ray from 800 units away reports t a few thousandths off, its point that
    // deep inside (or outside) a sphere of radius 5 -, and a scattered ray that starts inside re-hits the same sphere from within at
    // t > 0.001: one more bounce, darker (Cornell box + 10 000 spheres: 7 % more rays, the frame 8 % darker than f64). The direction
    // centre -> point is good to an ulp whatever t is; the point at distance r along it is what f64 computes to 13 digits. Normal, uv and
    // tangents are taken from the point as computed, as before. f64: nothing changes.
    if (sizeof(R) == 4) hp = S.center + n * (std::fabs(S.radius) / n.length());  // (|r|: a negative radius must not send the point to the antipode)
    out = {root, hp, {tangent, bi_tangent, normal}, uv, front, S.mat};
    return true;
  }
  bool hit_medium(const typename Scene<R>::Med& M, uint32_t midx, const Ray<R>& r, R tmin, R tmax, Cand<R>& out) {  // constant_medium.rs:35-79
    const R inf = std::numeric_limits<R>::infinity();
    Cand<R> rec1, rec2;
    if (!hit_ref(M.boundary, r, -inf, inf, rec1)) return false;
    if (!hit_ref(M.boundary, r, rec1.t + (R)0.0001, inf, rec2)) return false;
    R t1 = std::fmax(rec1.t, tmin);
    R t2 = std::fmin(rec2.t, tmax);
    if (t1 >= t2) return false;
    t1 = std::fmax(t1, (R)0);
    R r_length = r.direction.length();
    R distance_inside = (t2 - t1) * r_length;
    // sub-stream: counters 0x40000000 + (depth<<20 | medium<<8) + i   (DESIGN.md "RNG")
    uint32_t c = 0x40000000u + (((cur_depth & 0x3FFu) << 20) | ((midx & 0xFFFu) << 8));
    R hit_distance = M.nid * log_r(u32_to_unit<R>(rng.at(c++)));
    if (hit_distance > distance_inside) return false;
    R t = t1 + hit_distance / r_length;
    V3<R> p;
    for (;;) {  // random_unit_vector = random_in_unit_sphere().unit() (vec3.rs:395-397)
      p.x = u32_to_unit<R>(rng.at(c++)) * (R)2 + (R)-1;
      p.y = u32_to_unit<R>(rng.at(c++)) * (R)2 + (R)-1;
      p.z = u32_to_unit<R>(rng.at(c++)) * (R)2 + (R)-1;
      if (p.length_squared() < (R)1) break;
    }
    out = {t, r.at(t), {{1, 1, 1}, {1, 1, 1}, p.unit()}, {0.f, 0.f}, false, M.mat};
    return true;
  }

  // BvhItem::hit / Bvh::hit (bvh.rs:38-44,165-180): fixed left->right, right searched in [min, t_left].
  bool hit_ref(uint32_t ref, const Ray<R>& r, R tmin, R tmax, Cand<R>& out) {
    uint32_t idx = SOL_REF_INDEX(ref);
    switch (SOL_REF_KIND(ref)) {
      case SOL_REF_NONE: return false;
      case SOL_REF_NODE: {
        const auto& n = sc.nodes[idx];
        cnt.node_visits++;
        if (!aabb_hit(n.box, r)) return false;
        Cand<R> l;
        if (!hit_ref(n.left, r, tmin, tmax, l)) return hit_ref(n.right, r, tmin, tmax, out);
        Cand<R> rr;
        if (hit_ref(n.right, r, tmin, l.t, rr)) out = rr; else out = l;
        return true;
      }
      case SOL_REF_SPHERE: if (!hit_sphere(sc.spheres[idx], r, tmin, tmax, out)) return false; out.ref = ref; return true;
      case SOL_REF_QUAD: if (!hit_quad(sc.quads[idx], r, tmin, tmax, out)) return false; out.ref = ref; return true;
      case SOL_REF_TRIANGLE: if (!hit_triangle(sc.tris[idx], r, tmin, tmax, out)) return false; out.ref = ref; return true;
      case SOL_REF_MEDIUM: if (!hit_medium(sc.meds[idx], idx, r, tmin, tmax, out)) return false; out.ref = ref; return true;
    }
    return false;
  }

  // ---- textures (texture.rs:121-123,170-179; util/rgb_color.rs:37-43) -----------------------------------------
  V3<R> tex_color(int id, UvF uv) {
    const auto& t = sc.texs[id];
    if (t.kind == SOL_TEX_SOLID) return t.rgb;
    cnt.texels++;
    float au = std::fabs(uv.u), av = std::fabs(uv.v);
    float u = au - std::floor(au);                 // `% 1.` on a non-negative f32
    float v = 1.f - (av - std::floor(av));
    float x = u * ((float)t.w - 1.f), y = v * ((float)t.h - 1.f);
    uint32_t xi = x >= 0.f ? (x < 4294967296.f ? (uint32_t)x : 0xFFFFFFFFu) : 0u;  // Rust `as u32` (NaN -> 0)
    uint32_t yi = y >= 0.f ? (y < 4294967296.f ? (uint32_t)y : 0xFFFFFFFFu) : 0u;
    if (xi >= t.w) xi = t.w - 1;  // get_pixel would panic; cannot happen for u,v in [0,1]
    if (yi >= t.h) yi = t.h - 1;
    const uint8_t* p = sc.texels + t.off + ((size_t)yi * t.w + xi) * 3;
    const R s = (R)(1.0 / 255.);
    return {(R)p[0] * s, (R)p[1] * s, (R)p[2] * s};
  }

  // Material::get_transformed_normal (mod.rs:108-110,209-213,251-255,304-308,438-444) + transform_normal_by_map (:386-389)
  V3<R> transformed_normal(int mid, const Onb<R>& onb, UvF uv) {
    const auto& m = sc.mats[mid];
    if (m.kind == SOL_MAT_BLEND) return transformed_normal(rnd() > m.param ? m.m1 : m.m2, onb, uv);
    if ((m.kind == SOL_MAT_LAMBERTIAN || m.kind == SOL_MAT_METAL || m.kind == SOL_MAT_DIELECTRIC) && m.normal >= 0) {
      V3<R> n = tex_color(m.normal, uv) * (R)2 - V3<R>{1, 1, 1};
      return onb.local(n);
    }
    return onb.normal;
  }

  // EXTENSION, not in the reference (include/solstrale_hip.h, SolSceneDesc::env_*): radiance of a ray that hits nothing, from a
  // latitude-longitude map. Direction -> (u, v) as a point of the reference's unit sphere (calculate_sphere_uv, sphere.rs:134-140),
  // nearest texel, row 0 = up.
  V3<R> env_color(const V3<R>& dir) {
    const V3<R> n = dir.unit();
    const R theta = acos_r(-n.y);
    const R phi = -atan2_r(n.z, n.x) + Consts<R>::pi;
    const float u = (float)(phi / ((R)2 * Consts<R>::pi)), v = (float)(theta / Consts<R>::pi);
    const float x = u * ((float)sc.env_w - 1.f), y = (1.f - v) * ((float)sc.env_h - 1.f);
    uint32_t xi = x >= 0.f ? (x < 4294967296.f ? (uint32_t)x : 0xFFFFFFFFu) : 0u;
    uint32_t yi = y >= 0.f ? (y < 4294967296.f ? (uint32_t)y : 0xFFFFFFFFu) : 0u;
    if (xi >= sc.env_w) xi = sc.env_w - 1;
    if (yi >= sc.env_h) yi = sc.env_h - 1;
    const float* p = sc.env + ((size_t)yi * sc.env_w + xi) * 3;
    return V3<R>{(R)p[0], (R)p[1], (R)p[2]} * sc.env_scale;
  }

  struct RayHit { V3<R> p, normal; int mat; R t; UvF uv; bool front; };

  // ---- light pdf (pdf.rs:75-102; quad.rs:132-148; triangle.rs:100-117; sphere.rs:40-62,142-153) ---------------
  R light_pdf_value(uint32_t ref, const V3<R>& origin, const V3<R>& direction) {
    Ray<R> ray = Ray<R>::make(origin, direction);
    const R inf = std::numeric_limits<R>::infinity();
    Cand<R> c;
    uint32_t idx = SOL_REF_INDEX(ref);
    switch (SOL_REF_KIND(ref)) {
      case SOL_REF_QUAD: {
        if (!hit_quad(sc.quads[idx], ray, (R)RAY_MIN, inf, c)) return 0;
        R ds = c.t * c.t * direction.length_squared();
        R cosine = std::fabs(direction.dot(c.onb.normal) / direction.length());
        return ds / (cosine * sc.quads[idx].area);
      }
      case SOL_REF_TRIANGLE: {
        if (!hit_triangle(sc.tris[idx], ray, (R)RAY_MIN, inf, c, false)) return 0;
        R ds = c.t * c.t * direction.length_squared();
        R cosine = std::fabs(direction.dot(c.onb.normal) / direction.length());
        return ds / (cosine * sc.tris[idx].area);
      }
      case SOL_REF_SPHERE: {
        const auto& S = sc.spheres[idx];
        if (!hit_sphere(S, ray, (R)RAY_MIN, inf, c)) return 0;
        R cos_theta_max = std::sqrt((R)1 - S.radius * S.radius / (S.center - origin).length_squared());
        R solid_angle = (R)2 * Consts<R>::pi * ((R)1 - cos_theta_max);
        return (R)1 / solid_angle;
      }
    }
    return 0;  // the reference panics for non-light shapes (hittable/mod.rs:28-35); unreachable by construction
  }
  V3<R> light_random_direction(uint32_t ref, uint32_t light_index, const V3<R>& origin) {
    uint32_t idx = SOL_REF_INDEX(ref);
    switch (SOL_REF_KIND(ref)) {
      case SOL_REF_QUAD: {
        const auto& Q = sc.quads[idx];
        R r1 = rnd(); R r2 = rnd();
        return Q.q + Q.u * r1 + Q.v * r2 - origin;
      }
      case SOL_REF_TRIANGLE: {  // triangle.rs:114-117: the parallelogram at the reference's first vertex
        const auto& F = sc.light_frames[light_index];
        R r1 = rnd(); R r2 = rnd();
        return F.v0 + F.e1 * r1 + F.e2 * r2 - origin;
      }
      case SOL_REF_SPHERE: {
        const auto& S = sc.spheres[idx];
        V3<R> direction = S.center - origin;
        Onb<R> uvw = Onb<R>::make(direction);
        R ds = direction.length_squared();
        R r1 = rnd(), r2 = rnd();
        R z = (R)1 + r2 * (std::sqrt((R)1 - S.radius * S.radius / ds) - (R)1);
        R c, s; sincos2pi(r1, c, s);
        R zz = std::sqrt((R)1 - z * z);
        return uvw.local({c * zz, s * zz, z});
      }
    }
    return {0, 0, 0};
  }
  R container_pdf_value(const V3<R>& origin, const V3<R>& direction) {  // pdf.rs:89-96
    R sum = 0;
    for (uint32_t l : sc.lights) sum += light_pdf_value(l, origin, direction);
    return sum / (R)sc.lights.size();
  }
  V3<R> container_pdf_generate(const V3<R>& origin) {  // pdf.rs:98-101
    uint32_t idx = rnd_index((uint32_t)sc.lights.size());
    return light_random_direction(sc.lights[idx], idx, origin);
  }

  // ---- scatter (material/mod.rs) -----------------------------------------------------------------------------
  struct Scatter { int type; V3<R> color; Ray<R> ray; R probability; bool has_af; R af; };  // 0 Pdf, 1 Basic, 2 Emission

  Scatter scatter(int mid, const Ray<R>& ray, const RayHit& rec) {
    const auto& m = sc.mats[mid];
    cnt.shades++;
    Scatter s{};
    switch (m.kind) {
      case SOL_MAT_LAMBERTIAN: {  // mod.rs:191-207
        V3<R> color = tex_color(m.albedo, rec.uv);
        Onb<R> uvw = Onb<R>::make(rec.normal);
        V3<R> dir = rnd() < (R)0.5 ? container_pdf_generate(rec.p) : uvw.local(random_cosine_direction());  // mix_generate (pdf.rs:42-48)
        Ray<R> scattered = Ray<R>::make(rec.p, dir);
        R cosine_theta = dir.unit().dot(uvw.normal);
        R cos_pdf = std::fmax(cosine_theta / Consts<R>::pi, (R)0);                       // CosinePdf::value (pdf.rs:64-67)
        R mix = (R)0.5 * container_pdf_value(rec.p, dir) + (R)0.5 * cos_pdf;             // mix_value (pdf.rs:36-38)
        R cos_theta = rec.normal.dot(dir.unit());                                        // scattering_pdf_value (mod.rs:179-186)
        R scattering = cos_theta < (R)0 ? (R)0 : cos_theta / Consts<R>::pi;
        s.type = 0; s.color = color; s.ray = scattered; s.probability = scattering / mix;
        return s;
      }
      case SOL_MAT_METAL: {  // mod.rs:239-249
        V3<R> reflected = ray.direction.unit().reflect(rec.normal);
        s.type = 1; s.color = tex_color(m.albedo, rec.uv);
        s.ray = Ray<R>::make(rec.p, reflected + random_in_unit_sphere() * m.param);
        return s;
      }
      case SOL_MAT_DIELECTRIC: {  // mod.rs:279-302,312-316
        R ratio = rec.front ? (R)1 / m.param : m.param;
        V3<R> ud = ray.direction.unit();
        R cos_theta = std::fmin(ud.neg().dot(rec.normal), (R)1);
        R sin_theta = std::sqrt((R)1 - cos_theta * cos_theta);
        bool cannot_refract = ratio * sin_theta > (R)1;
        bool refl = cannot_refract;
        if (!refl) {
          R r0 = ((R)1 - ratio) / ((R)1 + ratio);
          r0 = r0 * r0;
          R x = (R)1 - cos_theta, x2 = x * x, x4 = x2 * x2;
          refl = r0 + ((R)1 - r0) * (x4 * x) > rnd();
        }
        V3<R> dir = refl ? ud.reflect(rec.normal) : ud.refract(rec.normal, ratio);
        s.type = 1; s.color = tex_color(m.albedo, rec.uv); s.ray = Ray<R>::make(rec.p, dir);
        return s;
      }
      case SOL_MAT_DIFFUSE_LIGHT: {  // mod.rs:359-368
        s.type = 2;
        s.color = rec.front ? tex_color(m.albedo, rec.uv) : V3<R>{0, 0, 0};
        s.has_af = !m.param_none; s.af = m.param;
        return s;
      }
      case SOL_MAT_ISOTROPIC: {  // mod.rs:396-410
        V3<R> color = tex_color(m.albedo, rec.uv);
        V3<R> dir;
        if (rnd() < (R)0.5) dir = container_pdf_generate(rec.p);
        else dir = random_in_unit_sphere().unit();                                        // SpherePdf::generate
        Ray<R> scattered = Ray<R>::make(rec.p, dir);
        const R sphere_pdf = (R)(1. / (4. * PI_D));
        R mix = (R)0.5 * container_pdf_value(rec.p, dir) + (R)0.5 * sphere_pdf;
        s.type = 0; s.color = color; s.ray = scattered; s.probability = sphere_pdf / mix;
        return s;
      }
      case SOL_MAT_BLEND:  // mod.rs:430-436
        cnt.shades--;
        return scatter(rnd() > m.param ? m.m1 : m.m2, ray, rec);
    }
    return s;
  }

  // ---- ray_color <-> shade (renderer/mod.rs:164-206, shader.rs:62-125) ---------------------------------------
  struct AttCol { V3<R> color; bool has_af; R af; R len; };

  static R filter_color_value(R v) { return std::isnan(v) ? (R)0 : std::fmin(v, (R)3); }  // shader.rs:117-125

  AttCol ray_color(const Ray<R>& ray, uint32_t depth, R acc_len) {
    cnt.rays++;
    if (live) cnt.live_rays++;  // (OrcStats::live_rays: the rays the device traces too)
    cur_depth = depth;
    Cand<R> c;
    bool any_hit = hit_ref(sc.root, ray, (R)RAY_MIN, std::numeric_limits<R>::infinity(), c);
    // fp32 rule 8 (float instantiation only): a ray does not hit the FLAT primitive it leaves. A line meets a plane once, and the ray starts on
    // it: in f64 the second "hit" lies at t ~ 1e-11, far below RAY_MIN, and never counts (the reference's behaviour); in fp32 the start point is a few
    // 1e-5 off its plane at coordinates of hundreds, and a grazing ray (|n.d| of a few per cent) finds the plane again at t just above RAY_MIN = 1e-3 - from
    // C1's camera 7 samples in a million went into the box they had just left and came back black. When the closest hit is that primitive, the search is
    // repeated behind it.
    if (sizeof(R) == 4 && any_hit && ray.from != 0u && c.ref == ray.from) {
      Cand<R> behind;
      any_hit = hit_ref(sc.root, ray, std::nextafter(c.t, std::numeric_limits<R>::infinity()), std::numeric_limits<R>::infinity(), behind);
      if (any_hit) c = behind;
    }
    if (trace) {
      uint32_t rb = any_hit ? c.ref : 0u;
      float rf; std::memcpy(&rf, &rb, 4);
      const float row[12] = {(float)ray.origin.x, (float)ray.origin.y, (float)ray.origin.z, (float)ray.direction.x, (float)ray.direction.y,
                             (float)ray.direction.z, any_hit ? (float)c.t : std::numeric_limits<float>::infinity(), rf, 0.f, (float)depth, 0.f, 0.f};
      trace->insert(trace->end(), row, row + 12);
    }
    if (!any_hit) return {sc.env ? env_color(ray.direction) : sc.background, false, 0, 0};  // renderer/mod.rs:197-204
    RayHit rec{c.p, {}, c.mat, c.t, c.uv, c.front};
    rec.normal = transformed_normal(c.mat, c.onb, c.uv);  // RayHit::new (material/mod.rs:50), closest hit only
    switch (sc.shader) {
      case SOL_SHADER_ALBEDO: { Scatter s = scatter(rec.mat, ray, rec); return {s.color, false, 0, 0}; }       // shader.rs:141-151
      case SOL_SHADER_NORMAL: return {rec.normal, false, 0, 0};                                               // shader.rs:165-172
      case SOL_SHADER_SIMPLE: {                                                                               // shader.rs:192-214
        Scatter s = scatter(rec.mat, ray, rec);
        if (s.type == 2) return {s.color, false, 0, 0};
        R f = rec.normal.dot({1, 1, -1}) * (R)0.5 + (R)0.75;
        return {s.color * f, false, 0, 0};
      }
    }
    // PathTracingShader::shade
    if (depth >= sc.max_depth) return {{0, 0, 0}, false, 0, 0};
    R total = rec.t + acc_len;
    Scatter s = scatter(rec.mat, ray, rec);
    if (s.type == 2) return {s.color, s.has_af, s.af, total};
    if (s.type == 0) {  // a ScatterPdf level whose factor has no positive component returns 0 whatever its child returns: the device stops here
      const V3<R> a = s.color * s.probability;
      if (!(a.x > (R)0 || a.y > (R)0 || a.z > (R)0)) live = false;
    }
    s.ray.from = (SOL_REF_KIND(c.ref) == SOL_REF_QUAD || SOL_REF_KIND(c.ref) == SOL_REF_TRIANGLE) ? c.ref : 0u;  // (rule 8: the scattered ray starts on this primitive)
    AttCol child = ray_color(s.ray, depth + 1, total);
    if (s.type == 1) return {s.color * child.color, child.has_af, child.af, child.len};
    V3<R> sc_col = s.color * s.probability * child.color;
    return {{filter_color_value(sc_col.x), filter_color_value(sc_col.y), filter_color_value(sc_col.z)}, child.has_af, child.af, child.len};
  }

  // One sample of pixel (x, y_ref) where y_ref counts from the image bottom (renderer/mod.rs:261-268).
  V3<R> sample_pixel(uint32_t x, uint32_t y_ref, uint32_t sample, uint64_t seed) {
    uint32_t row = (sc.height - 1) - y_ref;
    rng.init(seed, row * sc.width + x, sample);
    cnt.samples++;
    live = true;
    R u = ((R)x + rnd()) / (R)(sc.width - 1);
    R v = ((R)y_ref + rnd()) / (R)(sc.height - 1);
    float uf = (float)u, vf = (float)v;
    // Camera::get_ray (camera.rs:77-89)
    V3<R> offset{0, 0, 0};
    if (sc.lens_radius > (R)0) {
      V3<R> rd = random_in_unit_disc() * sc.lens_radius;
      offset = sc.cam_u * rd.x + sc.cam_vv * rd.y;
    }
    V3<R> dir = sc.cam_llc + (sc.cam_h * (R)uf) + (sc.cam_v * (R)vf) - sc.cam_origin - offset;
    Ray<R> ray = Ray<R>::make(sc.cam_origin + offset, dir);
    AttCol a = ray_color(ray, 0, 0);
    if (a.has_af) return a.color * (R)1 / ((R)1 + a.af * a.len);  // get_attenuated_color (material/mod.rs:127-131)
    return a.color;
  }
};

template <typename R>
int render_impl(const SolSceneDesc* d, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t first, uint32_t n,
                uint64_t seed, int threads, double* out, OrcStats* stats) {
  if (!d || !out || d->width < 2 || d->height < 2) return -1;
  if (d->n_lights == 0) return -2;
  x1 = std::min(x1, d->width); y1 = std::min(y1, d->height);
  Scene<R> sc(*d);
  if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
  if (threads <= 0) threads = 1;
  std::atomic<uint32_t> next_row{y0};
  std::vector<Counters> counters(threads);
  auto work = [&](int tid) {
    Tracer<R> tr(sc);
    for (;;) {
      uint32_t row = next_row.fetch_add(1);  // output row (0 = top), like the reference's row tasks (renderer/mod.rs:242)
      if (row >= y1) break;
      uint32_t y_ref = (d->height - 1) - row;
      for (uint32_t x = x0; x < x1; ++x) {
        V3<R> sum{0, 0, 0};
        for (uint32_t s = first; s < first + n; ++s) sum = sum + tr.sample_pixel(x, y_ref, s, seed);  // add_row_data order
        double* o = out + ((size_t)row * d->width + x) * 3;
        o[0] += (double)sum.x; o[1] += (double)sum.y; o[2] += (double)sum.z;
      }
    }
    counters[tid] = tr.cnt;
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& t : pool) t.join();
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    for (auto& c : counters) {
      stats->samples += c.samples; stats->rays += c.rays; stats->node_visits += c.node_visits;
      stats->sphere_tests += c.sphere_tests; stats->quad_tests += c.quad_tests; stats->triangle_tests += c.tri_tests;
      stats->shades += c.shades; stats->texel_fetches += c.texels; stats->live_rays += c.live_rays;
    }
    stats->threads = (uint32_t)threads;
  }
  return 0;
}

}  // namespace

extern "C" {

int orc_render(const SolSceneDesc* d, int real_kind, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t first,
               uint32_t n, uint64_t seed, int threads, double* out, OrcStats* stats) {
  if (real_kind == ORC_F32) return render_impl<float>(d, x0, y0, x1, y1, first, n, seed, threads, out, stats);
  return render_impl<double>(d, x0, y0, x1, y1, first, n, seed, threads, out, stats);
}

// ---- known-answer-test hooks (tests/test_oracle_kat.py) ------------------------------------------------------
void orc_vec3_ops(const double a[3], const double b[3], double out[16]) {
  V3<double> A{a[0], a[1], a[2]}, B{b[0], b[1], b[2]};
  V3<double> s = A + B, d = A - B, m = A * B, c = A.cross(B);
  out[0] = s.x; out[1] = s.y; out[2] = s.z; out[3] = d.x; out[4] = d.y; out[5] = d.z;
  out[6] = m.x; out[7] = m.y; out[8] = m.z; out[9] = A.dot(B); out[10] = c.x; out[11] = c.y; out[12] = c.z;
  out[13] = A.length_squared(); out[14] = A.length(); out[15] = 0;
}
void orc_vec3_reflect(const double v[3], const double n[3], double out[3]) {
  V3<double> r = V3<double>{v[0], v[1], v[2]}.reflect({n[0], n[1], n[2]});
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_vec3_refract(const double v[3], const double n[3], double ior, double out[3]) {
  V3<double> r = V3<double>{v[0], v[1], v[2]}.refract({n[0], n[1], n[2]}, ior);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_vec3_unit(const double v[3], double out[3]) {
  V3<double> r = V3<double>{v[0], v[1], v[2]}.unit();
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_ray_at(const double o[3], const double d[3], double t, double out[3]) {
  V3<double> r = Ray<double>::make({o[0], o[1], o[2]}, {d[0], d[1], d[2]}).at(t);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int orc_aabb_hit(const double box[6], const double o[3], const double d[3]) {
  return aabb_hit<double>(box, Ray<double>::make({o[0], o[1], o[2]}, {d[0], d[1], d[2]})) ? 1 : 0;
}
void orc_onb_local(const double t[3], const double b[3], const double n[3], const double a[3], double out[3]) {
  Onb<double> o{{t[0], t[1], t[2]}, {b[0], b[1], b[2]}, {n[0], n[1], n[2]}};
  V3<double> r = o.local({a[0], a[1], a[2]});
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
// transform_normal_by_map with a solid-colour map (material/mod.rs:386-389, test :456-469)
void orc_transform_normal_by_map(const double rgb[3], const double t[3], const double b[3], const double n[3], double out[3]) {
  V3<double> c{rgb[0], rgb[1], rgb[2]};
  V3<double> m = c * 2. - V3<double>{1, 1, 1};
  Onb<double> o{{t[0], t[1], t[2]}, {b[0], b[1], b[2]}, {n[0], n[1], n[2]}};
  V3<double> r = o.local(m);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_rgb_to_vec3(const uint8_t p[3], double out[3]) {  // util/rgb_color.rs:37-43
  const double s = 1.0 / 255.;
  out[0] = p[0] * s; out[1] = p[1] * s; out[2] = p[2] * s;
}
void orc_to_rgb_color(const double col[3], uint32_t spp, uint8_t out[3]) {  // util/rgb_color.rs:14-35
  double scale = 1.0 / (double)spp;
  for (int c = 0; c < 3; ++c) {
    double v = std::sqrt(scale * col[c]);
    if (v < -0.999) v = -0.999;
    if (v > 0.999) v = 0.999;
    double s = 256. * v;
    out[c] = std::isnan(s) ? 0 : (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
  }
}
uint32_t orc_rng_bits(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t counter) {
  Rng r; r.init(seed, pixel, sample);
  return r.at(counter);
}
void orc_f32_funcs(float r, float x, float y, float out[5]) {
  float c, s; sincos2pi(r, c, s);
  out[0] = c; out[1] = s; out[2] = acos_r(x); out[3] = atan2_r(y, x); out[4] = log_r(r);
}
// Diagnostic: the rays of one path (same row layout as the device's sol_debug_path; dfs column left 0).
int orc_debug_path(const SolSceneDesc* d, int real_kind, uint32_t x, uint32_t y, uint32_t sample, uint64_t seed, float* rows, uint32_t max_rows) {
  std::vector<float> tr;
  float col[3];
  if (real_kind == ORC_F32) {
    Scene<float> sc(*d); Tracer<float> t(sc); t.trace = &tr;
    V3<float> c = t.sample_pixel(x, (d->height - 1) - y, sample, seed);
    col[0] = c.x; col[1] = c.y; col[2] = c.z;
  } else {
    Scene<double> sc(*d); Tracer<double> t(sc); t.trace = &tr;
    V3<double> c = t.sample_pixel(x, (d->height - 1) - y, sample, seed);
    col[0] = (float)c.x; col[1] = (float)c.y; col[2] = (float)c.z;
  }
  uint32_t n = (uint32_t)(tr.size() / 12);
  if (n + 1 > max_rows) n = max_rows - 1;
  std::memcpy(rows, tr.data(), (size_t)n * 12 * sizeof(float));
  float* last = rows + (size_t)n * 12;
  std::memset(last, 0, 12 * sizeof(float));
  last[0] = col[0]; last[1] = col[1]; last[2] = col[2]; last[3] = -1.0f;
  return (int)n;
}

// fp32 function table with the row layouts of the device's sol_eval (tests/test_gpu_functions.py compares bit for bit).
int orc_eval_f32(uint32_t fn, const float* in, uint32_t n, uint32_t is, float* out, uint32_t os) {
  typedef V3<float> F3;
  auto u2f = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
  auto f2u = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
  SolSceneDesc empty{};
  Scene<float> sc(empty);
  for (uint32_t i = 0; i < n; ++i) {
    const float* x = in + (size_t)i * is;
    float* y = out + (size_t)i * os;
    Tracer<float> tr(sc);
    const float inf = std::numeric_limits<float>::infinity();
    (void)inf;
    switch (fn) {
      case 0: {
        float a = x[0], b = x[1], c = x[2];
        y[0] = a / b; y[1] = std::sqrt(std::fabs(a)); y[2] = 1.0f / a; y[3] = a * b + c; y[4] = std::fmax(a, b);
        y[5] = std::fmin(a, b); y[6] = std::floor(a);
        break;
      }
      case 1: {
        float c, s; sincos2pi(x[0], c, s);
        y[0] = c; y[1] = s; y[2] = acos_r(x[1]); y[3] = atan2_r(x[2], x[1]); y[4] = log_r(x[0]);
        break;
      }
      case 2: {
        Rng r; r.init(((uint64_t)f2u(x[1]) << 32) | f2u(x[0]), f2u(x[2]), f2u(x[3]));
        y[0] = u2f(r.at(f2u(x[4]))); y[1] = u32_to_unit<float>(r.at(f2u(x[4])));
        break;
      }
      case 3: {
        F3 v{x[0], x[1], x[2]}, nn{x[3], x[4], x[5]};
        F3 u = v.unit(), rf = v.reflect(nn), rr = v.refract(nn, x[6]);
        Onb<float> o = Onb<float>::make(v);
        const F3 r[6] = {u, rf, rr, o.tangent, o.bi_tangent, o.normal};
        for (int k = 0; k < 6; ++k) { y[3 * k] = r[k].x; y[3 * k + 1] = r[k].y; y[3 * k + 2] = r[k].z; }
        break;
      }
      case 4: {
        Scene<float>::Sph S{{x[0], x[1], x[2]}, x[3], 0, 0};
        Cand<float> c;
        tr.sphere_slack = x[12];
        bool h = tr.hit_sphere(S, Ray<float>::make({x[4], x[5], x[6]}, {x[7], x[8], x[9]}), x[10], x[11], c);
        y[0] = h ? 1.f : 0.f; y[1] = h ? c.t : 0.f;
        break;
      }
      case 5: {
        Scene<float>::Qd Q{{x[4], x[5], x[6]}, {x[10], x[11], x[12]}, {x[13], x[14], x[15]}, {x[0], x[1], x[2]}, {x[7], x[8], x[9]}, x[3], 0, 0, 0};
        Cand<float> c;
        bool h = tr.hit_quad(Q, Ray<float>::make({x[16], x[17], x[18]}, {x[19], x[20], x[21]}), x[22], x[23], c);
        y[0] = h ? 1.f : 0.f; y[1] = h ? c.t : 0.f; y[2] = h ? c.uv.u : 0.f; y[3] = h ? c.uv.v : 0.f;
        break;
      }
      case 6: {
        // uv0=(0,0), uv1=(1,0), uv2=(0,1) make the interpolated Uv equal the barycentrics (u, v)
        Scene<float>::Tri T{{x[0], x[1], x[2]}, {x[3], x[4], x[5]}, {x[6], x[7], x[8]}, {0, 0, 1}, {1, 0, 0}, {0, 1, 0}, 0,
                            {0.f, 0.f}, {1.f, 0.f}, {0.f, 1.f}, 0, 0};
        Cand<float> c;
        bool h = tr.hit_triangle(T, Ray<float>::make({x[9], x[10], x[11]}, {x[12], x[13], x[14]}), x[15], x[16], c);
        y[0] = h ? 1.f : 0.f; y[1] = h ? c.t : 0.f; y[2] = h ? c.uv.u : 0.f; y[3] = h ? c.uv.v : 0.f;
        break;
      }
      case 7: {
        Ray<float> r = Ray<float>::make({x[6], x[7], x[8]}, {x[9], x[10], x[11]});
        y[0] = aabb_hit<float>(x, r) ? 1.f : 0.f;
        y[1] = 0.f;  // the reference's Aabb::hit returns only the predicate
        break;
      }
      case 8: {
        tr.rng.init(((uint64_t)f2u(x[1]) << 32) | f2u(x[0]), f2u(x[2]), f2u(x[3]));
        F3 c = tr.random_cosine_direction();
        F3 s = tr.random_in_unit_sphere();
        y[0] = c.x; y[1] = c.y; y[2] = c.z; y[3] = s.x; y[4] = s.y; y[5] = s.z; y[6] = u2f(tr.rng.ctr);
        break;
      }
      default: return -1;
    }
  }
  return 0;
}

// Closest hit of one ray in reference order (for traversal parity tests): returns 1 and fills t / prim ref.
int orc_closest_hit(const SolSceneDesc* d, int real_kind, const double o[3], const double dir[3], double* t_out, uint32_t* mat_out) {
  if (real_kind == ORC_F32) {
    Scene<float> sc(*d); Tracer<float> tr(sc); tr.rng.init(0, 0, 0);
    Cand<float> c;
    if (!tr.hit_ref(sc.root, Ray<float>::make({(float)o[0], (float)o[1], (float)o[2]}, {(float)dir[0], (float)dir[1], (float)dir[2]}), (float)RAY_MIN, std::numeric_limits<float>::infinity(), c)) return 0;
    *t_out = c.t; *mat_out = (uint32_t)c.mat; return 1;
  }
  Scene<double> sc(*d); Tracer<double> tr(sc); tr.rng.init(0, 0, 0);
  Cand<double> c;
  if (!tr.hit_ref(sc.root, Ray<double>::make({o[0], o[1], o[2]}, {dir[0], dir[1], dir[2]}), RAY_MIN, std::numeric_limits<double>::infinity(), c)) return 0;
  *t_out = c.t; *mat_out = (uint32_t)c.mat; return 1;
}

}  // extern "C"
