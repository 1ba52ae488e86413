// solstrale_host.cpp -- host side above the C ABI: scene construction, BVH builder, flattener, ray_trace().
// See solstrale.hpp for the mapping to the reference's files. Nothing in this file is on the per-sample
// path; it runs once per scene (construction) or once per pass batch (ray_trace loop).
#include "solstrale.hpp"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <future>
#include <stdexcept>
#include <unordered_map>

namespace solstrale {

static const double PI = 3.14159265358979323846;
static const double PAD_DELTA = 0.0001;  // src/geo/mod.rs:11

static inline double degrees_to_radians(double d) { return d * (PI / 180.); }  // src/util/mod.rs:11-13

// ---- Aabb (src/geo/mod.rs:87-158) -----------------------------------------------------------------------
Aabb Aabb::new_from_2_points(const Vec3& a, const Vec3& b) {
  Aabb r;
  r.x = {std::fmin(a.x, b.x), std::fmax(a.x, b.x)};
  r.y = {std::fmin(a.y, b.y), std::fmax(a.y, b.y)};
  r.z = {std::fmin(a.z, b.z), std::fmax(a.z, b.z)};
  return r;
}
Aabb Aabb::new_from_3_points(const Vec3& a, const Vec3& b, const Vec3& c) {
  Aabb r;
  r.x = {std::fmin(std::fmin(a.x, b.x), c.x), std::fmax(std::fmax(a.x, b.x), c.x)};
  r.y = {std::fmin(std::fmin(a.y, b.y), c.y), std::fmax(std::fmax(a.y, b.y), c.y)};
  r.z = {std::fmin(std::fmin(a.z, b.z), c.z), std::fmax(std::fmax(a.z, b.z), c.z)};
  return r;
}
Aabb Aabb::pad_if_needed() const {
  Aabb r;
  r.x = x.size() >= PAD_DELTA ? x : x.expand(PAD_DELTA);
  r.y = y.size() >= PAD_DELTA ? y : y.expand(PAD_DELTA);
  r.z = z.size() >= PAD_DELTA ? z : z.expand(PAD_DELTA);
  return r;
}

RotationX::RotationX(double a) : sin_theta(std::sin(degrees_to_radians(a))), cos_theta(std::cos(degrees_to_radians(a))) {}
RotationY::RotationY(double a) : sin_theta(std::sin(degrees_to_radians(a))), cos_theta(std::cos(degrees_to_radians(a))) {}
RotationZ::RotationZ(double a) : sin_theta(std::sin(degrees_to_radians(a))), cos_theta(std::cos(degrees_to_radians(a))) {}

// ---- textures -------------------------------------------------------------------------------------------
Textures SolidColor::create(double r, double g, double b) {
  auto t = std::make_shared<Texture>();
  t->kind = SOL_TEX_SOLID;
  t->color = {r, g, b};
  return t;
}
Textures ImageMap::create(std::shared_ptr<const RgbImage> image) {
  if (!image || image->width == 0 || image->height == 0 ||
      image->data.size() != (size_t)image->width * image->height * 3)
    throw std::runtime_error("ImageMap: malformed image");
  auto t = std::make_shared<Texture>();
  t->kind = SOL_TEX_IMAGE;
  t->image = std::move(image);
  return t;
}

// src/util/height_map.rs:68-95 (Sobel, STRENGTH 6, f32 arithmetic, edge pixels duplicated)
static std::shared_ptr<const RgbImage> height_to_normal_map(const RgbImage& img) {
  auto out = std::make_shared<RgbImage>();
  out->width = img.width;
  out->height = img.height;
  out->data.resize(img.data.size());
  auto px = [&](uint32_t y, uint32_t x) -> float { return (float)img.data[((size_t)y * img.width + x) * 3] / 255.0f; };
  for (uint32_t y = 0; y < img.height; ++y)
    for (uint32_t x = 0; x < img.width; ++x) {
      uint32_t n = y == 0 ? 0 : y - 1, s = y >= img.height - 1 ? img.height - 1 : y + 1;
      uint32_t w = x == 0 ? 0 : x - 1, e = x >= img.width - 1 ? img.width - 1 : x + 1;
      float nw = px(n, w), nn = px(n, x), ne = px(n, e), ww = px(y, w), ee = px(y, e), sw = px(s, w), ss = px(s, x),
            se = px(s, e);
      float p0 = -(se - sw + 2.0f * (ee - ww) + ne - nw);
      float p1 = -(nw - sw + 2.0f * (nn - ss) + ne - se);
      float p2 = 1.0f / 6.0f;
      float mag = std::sqrt(p0 * p0 + p1 * p1 + p2 * p2);
      float v[3] = {p0 / mag * 0.5f + 0.5f, p1 / mag * 0.5f + 0.5f, p2 / mag * 0.5f + 0.5f};
      for (int c = 0; c < 3; ++c) out->data[((size_t)y * img.width + x) * 3 + c] = (uint8_t)(v[c] * 255.0f);
    }
  return out;
}

Textures normal_texture_from_bump_map(std::shared_ptr<const RgbImage> image) {
  // load_bump_map's normal-vs-height vote (src/material/texture.rs:68-86)
  size_t num_normal = 0, num_height = 0;
  const double s = 1.0 / 255.;
  for (size_t i = 0; i + 2 < image->data.size(); i += 3) {
    Vec3 p(image->data[i] * s, image->data[i + 1] * s, image->data[i + 2] * s);
    if (std::fabs(p.length() - 1.) < 0.05) num_normal++;
    if (std::fabs(p.x - p.y) < 0.05 && std::fabs(p.y - p.z) < 0.05) num_height++;
  }
  if (num_height > num_normal) return ImageMap::create(height_to_normal_map(*image));
  return ImageMap::create(std::move(image));
}

// ---- materials ------------------------------------------------------------------------------------------
static Materials make_mat(int kind, Textures albedo, Textures normal, double param, Materials m1 = nullptr,
                          Materials m2 = nullptr) {
  auto m = std::make_shared<Material>();
  m->kind = kind;
  m->albedo = std::move(albedo);
  m->normal = std::move(normal);
  m->param = param;
  m->m1 = std::move(m1);
  m->m2 = std::move(m2);
  return m;
}
Materials Lambertian::create(Textures albedo, Textures normal) { return make_mat(SOL_MAT_LAMBERTIAN, albedo, normal, 0); }
Materials Metal::create(Textures albedo, Textures normal, double fuzz) { return make_mat(SOL_MAT_METAL, albedo, normal, fuzz); }
Materials Dielectric::create(Textures albedo, Textures normal, double ior) { return make_mat(SOL_MAT_DIELECTRIC, albedo, normal, ior); }
Materials DiffuseLight::create(double r, double g, double b, double half_length) {
  // attenuation_factor = attenuation_half_length.map(|a| 1. / a)   (mod.rs:338)
  double af = std::isnan(half_length) ? std::numeric_limits<double>::quiet_NaN() : 1. / half_length;
  return make_mat(SOL_MAT_DIFFUSE_LIGHT, SolidColor::create(r, g, b), nullptr, af);
}
Materials Blend::create(Materials m1, Materials m2, double f) { return make_mat(SOL_MAT_BLEND, nullptr, nullptr, f, m1, m2); }

// ---- hittables ------------------------------------------------------------------------------------------
Hittables Sphere::create(const Vec3& center, double radius, Materials mat) {  // sphere.rs:25-36
  auto h = std::make_shared<Hittable>();
  h->kind = SOL_REF_SPHERE;
  h->center = center;
  h->radius = radius;
  h->mat = std::move(mat);
  Vec3 r(radius, radius, radius);
  h->b_box = Aabb::new_from_2_points(center - r, center + r);
  return h;
}

Hittables Quad::create(const Vec3& q0, const Vec3& u0, const Vec3& v0, Materials mat, const Transformer& t) {  // quad.rs:34-66
  auto h = std::make_shared<Hittable>();
  h->kind = SOL_REF_QUAD;
  Vec3 q = t.transform(q0, false), u = t.transform(u0, true), v = t.transform(v0, true);
  h->b_box = Aabb()
                 .combine(Aabb::new_from_2_points(q, q + u))
                 .combine(Aabb::new_from_2_points(q, q + v))
                 .combine(Aabb::new_from_2_points(q, q + u + v))
                 .pad_if_needed();
  Vec3 n = u.cross(v);
  Vec3 normal = n.unit();
  h->q = q; h->u = u; h->v = v; h->normal = normal;
  h->d = normal.dot(q);
  h->w = n / n.dot(n);
  h->area = n.length();
  h->mat = std::move(mat);
  return h;
}

std::vector<Hittables> Quad::new_box(const Vec3& a, const Vec3& b, Materials mat, const Transformer& t) {  // quad.rs:69-129
  std::vector<Hittables> sides;
  Vec3 mn(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z));
  Vec3 mx(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z));
  Vec3 dx(mx.x - mn.x, 0., 0.), dy(0., mx.y - mn.y, 0.), dz(0., 0., mx.z - mn.z);
  sides.push_back(Quad::create(Vec3(mn.x, mn.y, mx.z), dx, dy, mat, t));
  sides.push_back(Quad::create(Vec3(mx.x, mn.y, mx.z), dz.neg(), dy, mat, t));
  sides.push_back(Quad::create(Vec3(mx.x, mn.y, mn.z), dx.neg(), dy, mat, t));
  sides.push_back(Quad::create(Vec3(mn.x, mn.y, mn.z), dz, dy, mat, t));
  sides.push_back(Quad::create(Vec3(mn.x, mx.y, mx.z), dx, dz.neg(), mat, t));
  sides.push_back(Quad::create(Vec3(mn.x, mn.y, mn.z), dx, dz, mat, t));
  return sides;
}

Hittables Triangle::create(const Vec3& v0, const Vec3& v1, const Vec3& v2, Materials mat, const Transformer& t) {
  return new_with_tex_coords(v0, v1, v2, Uv(), Uv(), Uv(), std::move(mat), t);
}
Hittables Triangle::new_with_tex_coords(const Vec3& a0, const Vec3& a1, const Vec3& a2, Uv uv0, Uv uv1, Uv uv2,
                                        Materials mat, const Transformer& t) {  // triangle.rs:53-96
  auto h = std::make_shared<Hittable>();
  h->kind = SOL_REF_TRIANGLE;
  Vec3 v0 = t.transform(a0, false), v1 = t.transform(a1, false), v2 = t.transform(a2, false);
  h->b_box = Aabb::new_from_3_points(v0, v1, v2).pad_if_needed();
  Vec3 v0v1 = v1 - v0, v0v2 = v2 - v0;
  Vec3 n = v0v1.cross(v0v2);
  h->normal = n.unit();
  h->area = n.length() / 2.;
  Vec3 dp1 = v1 - v0, dp2 = v2 - v0;
  Uv duv1 = uv1 - uv0, duv2 = uv2 - uv0;
  float r = 1.0f / (duv1.u * duv2.v - duv1.v * duv2.u);  // f32 arithmetic: Uv is f32 (geo/mod.rs:15-20)
  h->tangent = ((dp1 * (double)duv2.v - dp2 * (double)duv1.v) * (double)r).unit();
  h->bi_tangent = ((dp2 * (double)duv1.u - dp1 * (double)duv2.u) * (double)r).unit();
  h->v0 = v0; h->v0v1 = v0v1; h->v0v2 = v0v2;
  h->uv0 = uv0; h->uv1 = uv1; h->uv2 = uv2;
  h->mat = std::move(mat);
  return h;
}

Hittables ConstantMedium::create(Hittables boundary, double density, const Vec3& color) {  // constant_medium.rs:24-31
  auto h = std::make_shared<Hittable>();
  h->kind = SOL_REF_MEDIUM;
  h->b_box = boundary->b_box;
  h->boundary = std::move(boundary);
  h->negative_inverse_density = -1. / density;
  h->mat = make_mat(SOL_MAT_ISOTROPIC, SolidColor::new_from_vec3(color), nullptr, 0);
  return h;
}

// ---- BVH build (src/hittable/bvh.rs:84-162) ---------------------------------------------------------------
// The list is sorted level by level on the centres of the boxes: the centres are computed once and travel with their hittable, so that the million-element
// sorts of an OBJ model compare neighbouring doubles instead of chasing two pointers per comparison (the same keys, the same stable order, the same tree:
// 2.1 -> 0.6 s for the 1.09 M-triangle statue).
namespace {
struct BvhEntry {
  Hittables h;
  double c[3];
};
}  // namespace

static std::pair<double, double> bounding_box_spread(const std::vector<BvhEntry>& list, size_t lo, size_t hi, int axis) {
  double mn = std::numeric_limits<double>::infinity(), mx = -std::numeric_limits<double>::infinity();
  for (size_t i = lo; i < hi; ++i) {
    double c = list[i].c[axis];
    mn = std::fmin(mn, c);
    mx = std::fmax(mx, c);
  }
  return {mx - mn, (mn + mx) * 0.5};
}

static size_t sort_hittables_by_center(std::vector<BvhEntry>& list, size_t lo, size_t hi, double center, int axis) {
  // The reference uses sort_unstable_by (order of equal keys unspecified); a stable sort is one valid
  // outcome and keeps the build deterministic. total_cmp order == numeric order for the finite centres here.
  if (hi - lo <= 24) {  // (most calls are on a handful of elements: a stable insertion sort without std::stable_sort's buffer allocation - the same order)
    for (size_t i = lo + 1; i < hi; ++i) {
      size_t j = i;
      if (!(list[i].c[axis] < list[i - 1].c[axis])) continue;
      BvhEntry e = std::move(list[i]);
      while (j > lo && e.c[axis] < list[j - 1].c[axis]) { list[j] = std::move(list[j - 1]); --j; }
      list[j] = std::move(e);
    }
  } else {
    std::stable_sort(list.begin() + lo, list.begin() + hi, [axis](const BvhEntry& a, const BvhEntry& b) { return a.c[axis] < b.c[axis]; });
  }
  size_t i = 0;
  for (size_t k = lo; k < hi; ++k, ++i)
    if (list[k].c[axis] >= center) return i;
  return i;
}

static size_t sort_hittables_slice_by_most_spread_axis(std::vector<BvhEntry>& list, size_t lo, size_t hi) {
  auto [xs, xc] = bounding_box_spread(list, lo, hi, 0);
  auto [ys, yc] = bounding_box_spread(list, lo, hi, 1);
  auto [zs, zc] = bounding_box_spread(list, lo, hi, 2);
  size_t center;
  if (xs >= ys && xs >= zs) center = sort_hittables_by_center(list, lo, hi, xc, 0);
  else if (ys >= xs && ys >= zs) center = sort_hittables_by_center(list, lo, hi, yc, 1);
  else center = sort_hittables_by_center(list, lo, hi, zc, 2);
  size_t len = hi - lo;
  if (center == 0 || center == len) center = len / 2;  // could not split: split up the middle index
  return center;
}

static std::shared_ptr<Hittable> new_bvh(std::vector<BvhEntry>& list, size_t lo, size_t hi, int par_depth) {
  auto b = std::make_shared<Hittable>();
  b->kind = SOL_REF_NODE;
  size_t len = hi - lo;
  if (len == 1) {
    b->left = {2, list[lo].h};
    b->right = {0, nullptr};
    b->b_box = list[lo].h->b_box;
  } else if (len == 2) {
    b->left = {2, list[lo].h};
    b->right = {2, list[lo + 1].h};
    b->b_box = list[lo].h->b_box.combine(list[lo + 1].h->b_box);
  } else {
    size_t mid = sort_hittables_slice_by_most_spread_axis(list, lo, hi);
    std::shared_ptr<Hittable> l, r;
    if (par_depth > 0 && len > 4096) {  // rayon::join (bvh.rs:100-103)
      auto fut = std::async(std::launch::async, [&] { return new_bvh(list, lo, lo + mid, par_depth - 1); });
      r = new_bvh(list, lo + mid, hi, par_depth - 1);
      l = fut.get();
    } else {
      l = new_bvh(list, lo, lo + mid, 0);
      r = new_bvh(list, lo + mid, hi, 0);
    }
    b->b_box = l->b_box.combine(r->b_box);
    b->left = {1, l};
    b->right = {1, r};
  }
  return b;
}

Hittables Bvh::create(std::vector<Hittables> list) {
  if (list.empty()) {  // bvh.rs:62-68: both children None, default (empty) box
    auto b = std::make_shared<Hittable>();
    b->kind = SOL_REF_NODE;
    return b;
  }
  std::vector<BvhEntry> entries(list.size());
  for (size_t i = 0; i < list.size(); ++i) {
    if (!list[i]) throw std::runtime_error("Bvh::new: null hittable");
    const Vec3 c = list[i]->b_box.center();
    entries[i].c[0] = c.axis(0); entries[i].c[1] = c.axis(1); entries[i].c[2] = c.axis(2);
    entries[i].h = std::move(list[i]);
  }
  // (sub-trees are independent: as many levels of rayon::join as the host has cores for, at most 32 tasks)
  int par_depth = 0;
  for (unsigned hc = std::max(1u, std::thread::hardware_concurrency()); (1u << par_depth) < hc && par_depth < 5; ++par_depth) {}
  return new_bvh(entries, 0, entries.size(), par_depth);
}

static void collect_lights(const Hittables& h, std::vector<Hittables>& out) {
  if (!h) return;
  switch (h->kind) {
    case SOL_REF_NODE:  // bvh.rs:186-193
      if (h->left.kind) collect_lights(h->left.item, out);
      if (h->right.kind) collect_lights(h->right.item, out);
      break;
    case SOL_REF_MEDIUM: break;  // constant_medium.rs:85-87
    default:
      if (h->mat && h->mat->is_light()) out.push_back(h);
  }
}
std::vector<Hittables> get_lights(const Hittables& h) {
  std::vector<Hittables> out;
  collect_lights(h, out);
  return out;
}

// ---- camera (src/camera.rs:47-74) ---------------------------------------------------------------------------
static void put3(double* d, const Vec3& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
SolCamera camera_new(size_t image_width, size_t image_height, const CameraConfig& c) {
  double aspect_ratio = (double)image_width / (double)image_height;
  double theta = degrees_to_radians(c.vertical_fov_degrees);
  double h = std::tan(theta / 2.);
  double view_port_height = 2. * h;
  double view_port_width = aspect_ratio * view_port_height;
  Vec3 look_v = c.look_from - c.look_at;
  double focus_distance = look_v.length();
  Vec3 w = look_v.unit();
  Vec3 u = c.up.unit().cross(w).unit();
  Vec3 v = w.cross(u);
  Vec3 horizontal = (u * view_port_width) * focus_distance;
  Vec3 vertical = (v * view_port_height) * focus_distance;
  Vec3 llc = c.look_from - (horizontal / 2.) - (vertical / 2.) - (w * focus_distance);
  SolCamera cam{};
  put3(cam.origin, c.look_from);
  put3(cam.lower_left_corner, llc);
  put3(cam.horizontal, horizontal);
  put3(cam.vertical, vertical);
  put3(cam.u, u);
  put3(cam.v, v);
  cam.lens_radius = c.aperture_size / 2.;
  return cam;
}

bool RenderImageStrategy::should_generate_image(uint32_t sample, uint32_t total, double now_s, double last_s) const {
  switch (kind) {  // src/renderer/mod.rs:100-118
    case EverySample: return true;
    case Interval: return sample == total || (now_s - last_s) > interval_seconds;
    default: return sample == total;
  }
}

// ---- flatten -------------------------------------------------------------------------------------------------
namespace {
struct Flattener {
  explicit Flattener(FlatScene& f) : fs(f) {}
  FlatScene& fs;
  std::unordered_map<const Texture*, int32_t> tex_ids;
  std::unordered_map<const Material*, int32_t> mat_ids;
  std::unordered_map<const Hittable*, uint32_t> prim_refs;  // emitted primitives that are lights (what get_lights collects: the lights lookup)
  const Material* last_mat = nullptr;                        // (a mesh's triangles share one material: the last lookup answers most)
  int32_t last_mat_id = -1;
  uint32_t dfs = 0;
  uint32_t depth = 0;

  static SolAabb box(const Aabb& b) { return SolAabb{{b.x.min, b.x.max, b.y.min, b.y.max, b.z.min, b.z.max}}; }

  int32_t texture(const Textures& t) {
    if (!t) return -1;
    auto it = tex_ids.find(t.get());
    if (it != tex_ids.end()) return it->second;
    SolTexture s{};
    s.kind = t->kind;
    if (t->kind == SOL_TEX_IMAGE) {
      s.width = t->image->width;
      s.height = t->image->height;
      s.texel_offset = fs.texels.size();
      fs.texels.insert(fs.texels.end(), t->image->data.begin(), t->image->data.end());
    } else {
      s.rgb[0] = t->color.x; s.rgb[1] = t->color.y; s.rgb[2] = t->color.z;
    }
    int32_t id = (int32_t)fs.textures.size();
    fs.textures.push_back(s);
    tex_ids[t.get()] = id;
    return id;
  }

  int32_t material(const Materials& m) {
    if (!m) throw std::runtime_error("flatten: hittable without material");
    if (m.get() == last_mat) return last_mat_id;
    auto it = mat_ids.find(m.get());
    if (it != mat_ids.end()) { last_mat = m.get(); last_mat_id = it->second; return it->second; }
    SolMaterial s{};
    s.kind = m->kind;
    s.albedo_tex = texture(m->albedo);
    s.normal_tex = texture(m->normal);
    s.param = m->param;
    s.m1 = s.m2 = -1;
    if (m->kind == SOL_MAT_BLEND) {
      s.m1 = material(m->m1);
      s.m2 = material(m->m2);
    }
    int32_t id = (int32_t)fs.materials.size();
    fs.materials.push_back(s);
    mat_ids[m.get()] = id;
    return id;
  }

  uint32_t item(const BvhItem& it) {
    if (it.kind == 0) return SOL_MAKE_REF(SOL_REF_NONE, 0);
    return hittable(it.item);
  }

  // Returns the reference of `h`; Bvh (also when nested in a Leaf) becomes a node.
  uint32_t hittable(const Hittables& h) {
    if (!h) throw std::runtime_error("flatten: null hittable");
    switch (h->kind) {
      case SOL_REF_NODE: {
        uint32_t idx = (uint32_t)fs.nodes.size();
        fs.nodes.push_back(SolBvhNode{});
        depth++;
        fs.max_depth_nodes = std::max(fs.max_depth_nodes, depth);
        uint32_t l = item(h->left);
        uint32_t r = item(h->right);
        depth--;
        fs.nodes[idx].bbox = box(h->b_box);
        fs.nodes[idx].left = l;
        fs.nodes[idx].right = r;
        return SOL_MAKE_REF(SOL_REF_NODE, idx);
      }
      case SOL_REF_SPHERE: {
        SolSphere s{};
        put3(s.center, h->center);
        s.radius = h->radius;
        s.bbox = box(h->b_box);
        s.material = material(h->mat);
        s.dfs_index = dfs++;
        fs.spheres.push_back(s);
        uint32_t r = SOL_MAKE_REF(SOL_REF_SPHERE, fs.spheres.size() - 1);
        if (h->mat->is_light()) prim_refs[h.get()] = r;
        return r;
      }
      case SOL_REF_QUAD: {
        SolQuad s{};
        put3(s.q, h->q); put3(s.u, h->u); put3(s.v, h->v); put3(s.normal, h->normal); put3(s.w, h->w);
        s.d = h->d;
        s.area = h->area;
        s.bbox = box(h->b_box);
        s.material = material(h->mat);
        s.dfs_index = dfs++;
        fs.quads.push_back(s);
        uint32_t r = SOL_MAKE_REF(SOL_REF_QUAD, fs.quads.size() - 1);
        if (h->mat->is_light()) prim_refs[h.get()] = r;
        return r;
      }
      case SOL_REF_TRIANGLE: {
        SolTriangle s{};
        put3(s.v0, h->v0); put3(s.v0v1, h->v0v1); put3(s.v0v2, h->v0v2);
        put3(s.normal, h->normal); put3(s.tangent, h->tangent); put3(s.bi_tangent, h->bi_tangent);
        s.area = h->area;
        s.uv0[0] = h->uv0.u; s.uv0[1] = h->uv0.v; s.uv1[0] = h->uv1.u; s.uv1[1] = h->uv1.v;
        s.uv2[0] = h->uv2.u; s.uv2[1] = h->uv2.v;
        s.bbox = box(h->b_box);
        s.material = material(h->mat);
        s.dfs_index = dfs++;
        fs.triangles.push_back(s);
        uint32_t r = SOL_MAKE_REF(SOL_REF_TRIANGLE, fs.triangles.size() - 1);
        if (h->mat->is_light()) prim_refs[h.get()] = r;
        return r;
      }
      case SOL_REF_MEDIUM: {
        SolMedium s{};
        s.dfs_index = dfs++;
        s.negative_inverse_density = h->negative_inverse_density;
        s.bbox = box(h->b_box);
        s.material = material(h->mat);
        uint32_t idx = (uint32_t)fs.mediums.size();
        fs.mediums.push_back(s);
        uint32_t b = hittable(h->boundary);  // boundary sub-tree: reachable only through the medium
        fs.mediums[idx].boundary = b;
        return SOL_MAKE_REF(SOL_REF_MEDIUM, idx);
      }
      default: throw std::runtime_error("flatten: unknown hittable kind");
    }
  }
};
}  // namespace

std::unique_ptr<FlatScene> flatten(const Scene& scene) {
  if (!scene.world) throw std::runtime_error("flatten: scene without world");
  auto fs = std::make_unique<FlatScene>();
  Flattener f(*fs);
  uint32_t root = f.hittable(scene.world);
  for (auto& l : get_lights(scene.world)) {
    auto it = f.prim_refs.find(l.get());
    if (it == f.prim_refs.end()) throw std::runtime_error("flatten: light not found among primitives");
    fs->lights.push_back(it->second);
  }
  SolSceneDesc& d = fs->desc;
  d.abi_version = SOL_ABI_VERSION;
  d.width = (uint32_t)scene.render_config.width;
  d.height = (uint32_t)scene.render_config.height;
  d.shader_kind = scene.render_config.shader.kind;
  d.max_depth = scene.render_config.shader.max_depth;
  d.root = root;
  d.background[0] = scene.background_color.x;
  d.background[1] = scene.background_color.y;
  d.background[2] = scene.background_color.z;
  d.camera = camera_new(scene.render_config.width, scene.render_config.height, scene.camera);
  d.nodes = fs->nodes.data();           d.n_nodes = (uint32_t)fs->nodes.size();
  d.spheres = fs->spheres.data();       d.n_spheres = (uint32_t)fs->spheres.size();
  d.quads = fs->quads.data();           d.n_quads = (uint32_t)fs->quads.size();
  d.triangles = fs->triangles.data();   d.n_triangles = (uint32_t)fs->triangles.size();
  d.mediums = fs->mediums.data();       d.n_mediums = (uint32_t)fs->mediums.size();
  d.materials = fs->materials.data();   d.n_materials = (uint32_t)fs->materials.size();
  d.textures = fs->textures.data();     d.n_textures = (uint32_t)fs->textures.size();
  d.texels = fs->texels.data();         d.n_texel_bytes = fs->texels.size();
  d.lights = fs->lights.data();         d.n_lights = (uint32_t)fs->lights.size();
  d.env_texels = scene.environment.empty() ? nullptr : scene.environment.data();  // (owned by the Scene, which outlives the FlatScene)
  d.env_width = scene.env_width; d.env_height = scene.env_height; d.env_scale = scene.env_scale;
  return fs;
}

// ---- Nop post-processor arithmetic (src/util/rgb_color.rs:14-35) --------------------------------------------
void to_rgb_color(const double col[3], uint32_t spp, uint8_t out[3]) {
  double scale = 1.0 / (double)spp;
  for (int c = 0; c < 3; ++c) {
    double v = std::sqrt(scale * col[c]);
    if (v < -0.999) v = -0.999;  // Interval::clamp; NaN passes through both tests, `as u8` of NaN = 0
    if (v > 0.999) v = 0.999;
    double s = 256. * v;
    out[c] = std::isnan(s) ? 0 : (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));  // Rust `as u8` saturates
  }
}

PostProcessors BloomPostProcessor::create(double kernel_size_fraction, double threshold, double max_intensity) {
  if (!(kernel_size_fraction >= 0. && kernel_size_fraction <= 0.5))
    throw std::invalid_argument("kernel_size_fraction must be between 0 and 0.5");
  PostProcessors p;
  p.kind = PostProcessors::Bloom;
  p.kernel_size_fraction = kernel_size_fraction;
  p.threshold = std::isnan(threshold) ? Vec3{1., 1., 1.}.length() : threshold;
  p.max_intensity = std::isnan(max_intensity) ? std::numeric_limits<double>::max() : max_intensity;
  return p;
}

// ---- ray_trace (src/lib.rs:93-99, src/renderer/mod.rs:140-162,209-358) ---------------------------------------
std::string ray_trace(const Scene& scene, const std::function<void(RenderProgress&&)>& output,
                      const std::function<bool()>& abort, int device) {
  return ray_trace(scene, output, abort, std::vector<int>{device});
}

// The same on several GPUs of the node, from this one process (no reference analogue: its Rayon pool has no devices): the frame's 8x8
// blocks are dealt out over `devices` (by their cost in the creation probe: SOL_OPT_BALANCED_PARTITION + sol_scene_set_partition), every device holds the scene (created in parallel, one
// thread each) and renders its blocks of every batch concurrently; an image is gathered into the first device (sol_gather_local: peer copies) and
// post-processed there. The picture does not depend on n (RNG key = (seed, pixel, sample)). A device may be named more than once.
std::string ray_trace(const Scene& scene, const std::function<void(RenderProgress&&)>& output,
                      const std::function<bool()>& abort, const std::vector<int>& devices) {
  // Renderer::new
  if (get_lights(scene.world).empty()) return "Scene should have at least one light";
  if (devices.empty()) return "ray_trace: no device named";
  std::unique_ptr<FlatScene> fs;
  try {
    fs = flatten(scene);
  } catch (const std::exception& e) {
    return e.what();
  }
  const int n_dev = (int)devices.size();
  std::vector<SolScene*> devs((size_t)n_dev, nullptr);
  struct Guard { std::vector<SolScene*>& v; ~Guard() { for (SolScene* s : v) sol_scene_destroy(s); } } guard{devs};
  if (n_dev == 1) {
    if (sol_scene_create(&fs->desc, devices[0], &devs[0]) != SOL_OK) return sol_last_error();
  } else {
    std::vector<std::string> errs((size_t)n_dev);
    std::vector<std::thread> pool;
    for (int i = 0; i < n_dev; ++i)
      pool.emplace_back([&, i] {  // (sol_last_error is per thread: read where the call was made)
        // (the blocks are dealt out by their cost in the creation probe - deterministic, the same table on every device: sol_gather_local compares the
        // checksums - instead of b mod n: the devices finish a batch within a block's cost of each other)
        if (sol_scene_create(&fs->desc, devices[(size_t)i], &devs[(size_t)i]) != SOL_OK ||
            sol_scene_set_option(devs[(size_t)i], SOL_OPT_BALANCED_PARTITION, 1) != SOL_OK || sol_scene_set_partition(devs[(size_t)i], i, n_dev) != SOL_OK) {
          errs[(size_t)i] = sol_last_error();
          if (errs[(size_t)i].empty()) errs[(size_t)i] = "scene creation failed";
        }
      });
    for (auto& t : pool) t.join();
    for (const std::string& e : errs)
      if (!e.empty()) return e;
  }
  SolScene* const dev = devs[0];

  const RenderConfig& rc = scene.render_config;
  const uint32_t spp = rc.samples_per_pixel;
  const size_t npix = rc.width * rc.height;
  using clk = std::chrono::steady_clock;
  auto t0 = clk::now();
  auto secs = [&](clk::time_point t) { return std::chrono::duration<double>(t - t0).count(); };
  double last_image_time = -1e300;  // SystemTime::UNIX_EPOCH in the reference

  // Passes are batched on the device; one RenderProgress per sample index is still emitted (tests drain the
  // channel, tests/integration_tests.rs:316-321) and abort is polled between batches (renderer/mod.rs:237).
  uint32_t batch = rc.render_image_strategy.kind == RenderImageStrategy::EverySample ? 1u : 16u;
  // OnlyFinal shows nothing before the end, so a batch only bounds how long an abort waits: it doubles (multiples of 16, where the
  // sums do not depend on the split - DESIGN.md 3) while a batch takes less than ~50 ms, so that a small scene is not rendered
  // in launches of a few milliseconds each (the reference's profiling workload: 64-sample batches reach 2/3 of one launch's rate).
  // Interval: the same, with a batch kept below half the interval so that images still come when they are due.
  const bool only_final = rc.render_image_strategy.kind == RenderImageStrategy::OnlyFinal;
  const bool interval = rc.render_image_strategy.kind == RenderImageStrategy::Interval;
  const double grow_below = only_final ? 0.05 : interval ? std::min(0.05, 0.5 * rc.render_image_strategy.interval_seconds) : 0.0;
  if (only_final) batch = 64u;
  uint32_t batch_cap = 0xFFFFFFF0u;
  for (SolScene* d : devs) batch_cap = std::min(batch_cap, std::max(16u, sol_max_samples_per_call(d) / 16u * 16u));
  uint32_t done = 0;
  while (done < spp) {
    if (abort && abort()) return "";
    uint32_t n = std::min(std::min(batch, batch_cap), spp - done);
    const double t_batch = secs(clk::now());
    for (SolScene* d : devs)  // (asynchronous: the devices render their blocks of this batch side by side)
      if (sol_render(d, done, n, rc.seed) != SOL_OK) return sol_last_error();
    for (SolScene* d : devs)
      if (sol_sync(d) != SOL_OK) return sol_last_error();
    if (secs(clk::now()) - t_batch < grow_below && batch < batch_cap) batch *= 2u;
    for (uint32_t s = done + 1; s <= done + n; ++s) {
      double now = secs(clk::now());
      RenderProgress p;
      bool last_of_batch = (s == done + n);
      if (last_of_batch && rc.render_image_strategy.should_generate_image(s, spp, now, last_image_time)) {
        last_image_time = now;
        if (abort && abort()) return "";
        // post-processor chain (renderer/mod.rs:307-337) on the device: every processor but the last transforms the sums
        // (intermediate_post_process), the last one produces the image; an empty list produces none
        if (!rc.post_processors.empty()) {
          void* img = nullptr;
          if ((n_dev == 1 ? sol_resolve_image(dev, &img) : sol_gather_local(devs.data(), n_dev, &img)) != SOL_OK) return sol_last_error();
          for (size_t k = 0; k + 1 < rc.post_processors.size(); ++k) {
            const PostProcessors& pp = rc.post_processors[k];
            if (pp.kind == PostProcessors::Bloom &&
                sol_bloom(dev, img, s, pp.kernel_size_fraction, pp.threshold, pp.max_intensity) != SOL_OK)
              return sol_last_error();  // Nop: intermediate_post_process is the identity (nop.rs:36-46)
          }
          const PostProcessors& last = rc.post_processors.back();
          p.render_image.resize(npix * 3);
          int rcode = last.kind == PostProcessors::Bloom
                          ? sol_bloom_rgb8(dev, img, s, last.kernel_size_fraction, last.threshold, last.max_intensity, p.render_image.data())
                          : sol_tonemap_rgb8(dev, img, s, p.render_image.data());
          if (rcode != SOL_OK) return sol_last_error();
          p.has_image = true;
          p.width = (uint32_t)rc.width;
          p.height = (uint32_t)rc.height;
        }
      }
      double elapsed = std::max(now, 1e-9);
      p.progress = (double)s / (double)spp;
      p.fps = (double)s / elapsed;                                   // calculate_fps (mod.rs:367-373)
      p.estimated_time_left_s = elapsed / (float)s * (float)(spp - s);  // calculate_estimated_time_left
      if (output) output(std::move(p));
    }
    done += n;
  }
  return "";
}

}  // namespace solstrale
