// solstrale_host_c.cpp -- C wrappers (include/solstrale_host.h) over the C++ host mirror.
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/solstrale_host.h"
#include "solstrale.hpp"

using namespace solstrale;

static thread_local std::string g_err;

struct SolhBuilder {
  std::vector<std::shared_ptr<Transformer>> transforms;
  std::vector<Textures> textures;
  std::vector<Materials> materials;
  std::vector<Hittables> hittables;
  std::unique_ptr<FlatScene> flat;
  Scene scene;
  NopTransformer nop;

  const Transformer& tf(int id) const {
    if (id < 0) return nop;
    if ((size_t)id >= transforms.size()) throw std::runtime_error("bad transform id");
    return *transforms[id];
  }
  Textures tex(int id, bool optional) const {
    if (id < 0) {
      if (optional) return nullptr;
      throw std::runtime_error("texture id required");
    }
    if ((size_t)id >= textures.size()) throw std::runtime_error("bad texture id");
    return textures[id];
  }
  Materials mat(int id) const {
    if (id < 0 || (size_t)id >= materials.size()) throw std::runtime_error("bad material id");
    return materials[id];
  }
  Hittables hit(int id) const {
    if (id < 0 || (size_t)id >= hittables.size()) throw std::runtime_error("bad hittable id");
    return hittables[id];
  }
};

template <typename F>
static int guarded(F&& f) {
  try {
    return f();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

static Vec3 v3(const double* p) { return Vec3(p[0], p[1], p[2]); }

extern "C" {

SolhBuilder* solh_builder_new(void) { return new SolhBuilder(); }
void solh_builder_free(SolhBuilder* b) { delete b; }
const char* solh_last_error(void) { return g_err.c_str(); }

int solh_transform(SolhBuilder* b, int n_ops, const int* kinds, const double* params) {
  return guarded([&] {
    std::vector<std::shared_ptr<Transformer>> ops;
    for (int i = 0; i < n_ops; ++i) {
      const double* p = params + 3 * i;
      switch (kinds[i]) {
        case 0: ops.push_back(std::make_shared<Translation>(v3(p))); break;
        case 1: ops.push_back(std::make_shared<RotationX>(p[0])); break;
        case 2: ops.push_back(std::make_shared<RotationY>(p[0])); break;
        case 3: ops.push_back(std::make_shared<RotationZ>(p[0])); break;
        case 4: ops.push_back(std::make_shared<Scale>(p[0])); break;
        default: throw std::runtime_error("bad transform kind");
      }
    }
    b->transforms.push_back(std::make_shared<Transformations>(std::move(ops)));
    return (int)b->transforms.size() - 1;
  });
}

int solh_solid_color(SolhBuilder* b, double r, double g, double bl) {
  return guarded([&] {
    b->textures.push_back(SolidColor::create(r, g, bl));
    return (int)b->textures.size() - 1;
  });
}
static std::shared_ptr<RgbImage> make_image(uint32_t w, uint32_t h, const uint8_t* rgb8) {
  if (!rgb8 || !w || !h) throw std::runtime_error("bad image");
  auto im = std::make_shared<RgbImage>();
  im->width = w;
  im->height = h;
  im->data.assign(rgb8, rgb8 + (size_t)w * h * 3);
  return im;
}
int solh_image_map(SolhBuilder* b, uint32_t w, uint32_t h, const uint8_t* rgb8) {
  return guarded([&] {
    b->textures.push_back(ImageMap::create(make_image(w, h, rgb8)));
    return (int)b->textures.size() - 1;
  });
}
int solh_normal_texture(SolhBuilder* b, uint32_t w, uint32_t h, const uint8_t* rgb8) {
  return guarded([&] {
    b->textures.push_back(normal_texture_from_bump_map(make_image(w, h, rgb8)));
    return (int)b->textures.size() - 1;
  });
}

int solh_lambertian(SolhBuilder* b, int a, int n) {
  return guarded([&] {
    b->materials.push_back(Lambertian::create(b->tex(a, false), b->tex(n, true)));
    return (int)b->materials.size() - 1;
  });
}
int solh_metal(SolhBuilder* b, int a, int n, double fuzz) {
  return guarded([&] {
    b->materials.push_back(Metal::create(b->tex(a, false), b->tex(n, true), fuzz));
    return (int)b->materials.size() - 1;
  });
}
int solh_dielectric(SolhBuilder* b, int a, int n, double ior) {
  return guarded([&] {
    b->materials.push_back(Dielectric::create(b->tex(a, false), b->tex(n, true), ior));
    return (int)b->materials.size() - 1;
  });
}
int solh_diffuse_light(SolhBuilder* b, double r, double g, double bl, double half_length) {
  return guarded([&] {
    b->materials.push_back(DiffuseLight::create(r, g, bl, half_length));
    return (int)b->materials.size() - 1;
  });
}
int solh_blend(SolhBuilder* b, int m1, int m2, double f) {
  return guarded([&] {
    b->materials.push_back(Blend::create(b->mat(m1), b->mat(m2), f));
    return (int)b->materials.size() - 1;
  });
}

int solh_sphere(SolhBuilder* b, const double c[3], double radius, int material) {
  return guarded([&] {
    b->hittables.push_back(Sphere::create(v3(c), radius, b->mat(material)));
    return (int)b->hittables.size() - 1;
  });
}
int solh_quad(SolhBuilder* b, const double q[3], const double u[3], const double v[3], int material, int transform) {
  return guarded([&] {
    b->hittables.push_back(Quad::create(v3(q), v3(u), v3(v), b->mat(material), b->tf(transform)));
    return (int)b->hittables.size() - 1;
  });
}
int solh_box(SolhBuilder* b, const double a[3], const double bb[3], int material, int transform) {
  return guarded([&] {
    auto sides = Quad::new_box(v3(a), v3(bb), b->mat(material), b->tf(transform));
    int first = (int)b->hittables.size();
    for (auto& s : sides) b->hittables.push_back(s);
    return first;
  });
}
int solh_triangle(SolhBuilder* b, const double v0[3], const double v1[3], const double v2[3], const float uv[6],
                  int material, int transform) {
  return guarded([&] {
    Uv a, c, d;
    if (uv) { a = Uv(uv[0], uv[1]); c = Uv(uv[2], uv[3]); d = Uv(uv[4], uv[5]); }
    b->hittables.push_back(Triangle::new_with_tex_coords(v3(v0), v3(v1), v3(v2), a, c, d, b->mat(material), b->tf(transform)));
    return (int)b->hittables.size() - 1;
  });
}
int solh_triangles(SolhBuilder* b, uint32_t n, const double* vtx, const float* uvs, const int* mats, int transform) {
  return guarded([&] {
    int first = (int)b->hittables.size();
    const Transformer& t = b->tf(transform);
    b->hittables.reserve(b->hittables.size() + n);
    for (uint32_t i = 0; i < n; ++i) {
      const double* p = vtx + 9 * (size_t)i;
      Uv a, c, d;
      if (uvs) { const float* q = uvs + 6 * (size_t)i; a = Uv(q[0], q[1]); c = Uv(q[2], q[3]); d = Uv(q[4], q[5]); }
      b->hittables.push_back(Triangle::new_with_tex_coords(v3(p), v3(p + 3), v3(p + 6), a, c, d, b->mat(mats[i]), t));
    }
    return first;
  });
}
int solh_spheres(SolhBuilder* b, uint32_t n, const double* centers, const double* radii, const int* mats) {
  return guarded([&] {
    int first = (int)b->hittables.size();
    for (uint32_t i = 0; i < n; ++i) b->hittables.push_back(Sphere::create(v3(centers + 3 * (size_t)i), radii[i], b->mat(mats[i])));
    return first;
  });
}
int solh_constant_medium(SolhBuilder* b, int boundary, double density, const double color[3]) {
  return guarded([&] {
    b->hittables.push_back(ConstantMedium::create(b->hit(boundary), density, v3(color)));
    return (int)b->hittables.size() - 1;
  });
}
int solh_bvh(SolhBuilder* b, int n, const int* ids) {
  return guarded([&] {
    std::vector<Hittables> list;
    list.reserve(n);
    for (int i = 0; i < n; ++i) list.push_back(b->hit(ids[i]));
    b->hittables.push_back(Bvh::create(std::move(list)));
    return (int)b->hittables.size() - 1;
  });
}
int solh_bvh_range(SolhBuilder* b, int first, int n) {
  return guarded([&] {
    if (first < 0 || n < 0 || (size_t)first + n > b->hittables.size()) throw std::runtime_error("bad hittable range");
    std::vector<Hittables> list(b->hittables.begin() + first, b->hittables.begin() + first + n);
    b->hittables.push_back(Bvh::create(std::move(list)));
    return (int)b->hittables.size() - 1;
  });
}

const SolSceneDesc* solh_finish(SolhBuilder* b, int world, uint32_t width, uint32_t height, uint32_t shader_kind,
                                uint32_t max_depth, const double background[3], double vfov, double aperture,
                                const double look_from[3], const double look_at[3], const double up[3]) {
  try {
    Scene& s = b->scene;
    s.world = b->hit(world);
    s.camera.vertical_fov_degrees = vfov;
    s.camera.aperture_size = aperture;
    s.camera.look_from = v3(look_from);
    s.camera.look_at = v3(look_at);
    s.camera.up = v3(up);
    s.background_color = v3(background);
    s.render_config.width = width;
    s.render_config.height = height;
    s.render_config.shader = Shaders{shader_kind, max_depth};
    b->flat = flatten(s);
    return &b->flat->desc;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
int solh_environment(SolhBuilder* b, uint32_t width, uint32_t height, const float* rgb, double scale) {
  return guarded([&] {
    if (!rgb || !width || !height) throw std::runtime_error("environment: empty map");
    b->scene.environment.assign(rgb, rgb + (size_t)width * height * 3);
    b->scene.env_width = width; b->scene.env_height = height; b->scene.env_scale = scale;
    return 0;
  });
}
uint32_t solh_tree_depth(const SolhBuilder* b) { return b->flat ? b->flat->max_depth_nodes : 0; }

int solh_ray_trace(SolhBuilder* b, uint32_t spp, uint64_t seed, int strategy, double interval_seconds, int device,
                   solh_progress_fn progress, solh_abort_fn abort_cb, void* user) {
  return solh_ray_trace_devices(b, spp, seed, strategy, interval_seconds, 1, &device, progress, abort_cb, user);
}

int solh_ray_trace_devices(SolhBuilder* b, uint32_t spp, uint64_t seed, int strategy, double interval_seconds, int n_devices, const int* devices,
                           solh_progress_fn progress, solh_abort_fn abort_cb, void* user) {
  return guarded([&] {
    if (!b->scene.world) throw std::runtime_error("solh_ray_trace: call solh_finish first");
    if (n_devices < 1 || n_devices > 64 || !devices) throw std::runtime_error("solh_ray_trace_devices: 1 .. 64 devices");
    Scene& s = b->scene;
    s.render_config.samples_per_pixel = spp;
    s.render_config.seed = seed;
    s.render_config.render_image_strategy.kind =
        strategy == 0 ? RenderImageStrategy::EverySample : (strategy == 1 ? RenderImageStrategy::Interval : RenderImageStrategy::OnlyFinal);
    s.render_config.render_image_strategy.interval_seconds = interval_seconds;
    std::string err = ray_trace(
        s,
        [&](RenderProgress&& p) {
          if (progress)
            progress(user, p.progress, p.fps, p.estimated_time_left_s, p.has_image ? p.render_image.data() : nullptr, p.width, p.height);
        },
        [&]() { return abort_cb ? abort_cb(user) != 0 : false; }, std::vector<int>(devices, devices + n_devices));
    if (!err.empty()) throw std::runtime_error(err);
    return 0;
  });
}

int solh_load_obj(SolhBuilder* b, const char* path, const char* filename, int transform, int default_material,
                  solh_image_decoder decoder, void* user) {
  return guarded([&] {
    ImageDecoder dec = [&](const std::string& p, const char* what) -> std::shared_ptr<const RgbImage> {
      uint32_t w = 0, h = 0;
      const uint8_t* data = nullptr;
      const int rc = decoder ? decoder(user, p.c_str(), &w, &h, &data) : 1;
      if (rc == 1) throw std::runtime_error(std::string("Failed to open ") + what + " texture " + p + ": No such file or directory (os error 2)");
      if (rc != 0 || !data || !w || !h) throw std::runtime_error(std::string("Failed to decode ") + what + " texture " + p + ": unsupported or corrupt image");
      auto img = std::make_shared<RgbImage>();
      img->width = w; img->height = h;
      img->data.assign(data, data + (size_t)w * h * 3);
      return img;
    };
    Materials dm = default_material < 0 ? nullptr : b->mat(default_material);
    b->hittables.push_back(Obj(path, filename).load(b->tf(transform), dm, dec));
    return (int)b->hittables.size() - 1;
  });
}

int solh_set_post_processors(SolhBuilder* b, int n, const int* kinds, const double* params) {
  return guarded([&] {
    std::vector<PostProcessors> pp;
    for (int i = 0; i < n; ++i) {
      if (kinds[i] == 0) pp.push_back(NopPostProcessor::create());
      else if (kinds[i] == 1) pp.push_back(BloomPostProcessor::create(params[3 * i], params[3 * i + 1], params[3 * i + 2]));
      else throw std::runtime_error("solh_set_post_processors: unknown post-processor kind");
    }
    b->scene.render_config.post_processors = std::move(pp);
    return 0;
  });
}

void solh_abi_sizes(uint32_t out[11]) {
  const size_t s[11] = {sizeof(SolAabb), sizeof(SolBvhNode), sizeof(SolSphere), sizeof(SolQuad), sizeof(SolTriangle),
                        sizeof(SolMedium), sizeof(SolMaterial), sizeof(SolTexture), sizeof(SolCamera), sizeof(SolSceneDesc),
                        sizeof(SolStats)};
  for (int i = 0; i < 11; ++i) out[i] = (uint32_t)s[i];
}

void solh_to_rgb_color(const double col[3], uint32_t spp, uint8_t out[3]) { to_rgb_color(col, spp, out); }

}  // extern "C"
