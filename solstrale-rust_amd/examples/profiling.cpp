// profiling.cpp -- the reference's profiling binary (src/bin/profiling.rs:14-37) on the device path, as a native program:
// no Python, no torch - solstrale.hpp (the C++ mirror of Scene / RenderConfig / ray_trace) over the C ABI of
// include/solstrale_hip.h, which is what a Rust host would bind. The workload is the reference's:
// create_test_scene (tests/scenes.rs:17-122) at 800 x 400 with 1000 samples per pixel, ray_trace() draining the progress
// stream, the last image saved.
//
//   profiling [--width W] [--height H] [--spp N] [--seed S] [--device D | --devices D0,D1,..] [--texture tex.ppm] [--out out.ppm] [--repeat R]
//   (--devices: the same picture from this one process on several GPUs of the node - ray_trace(scene, output, abort, devices))
//
// The reference loads resources/textures/tex.jpg; file decoding is the caller's business on this side of the boundary
// (DESIGN.md 10), so the ground texture comes as a binary PPM (P6) - tests/test_gpu_examples.py converts the reference's
// file - or, without --texture, as a procedural checker (then the picture is not the reference's, the workload's shape is).
// Prints one JSON line: samples, seconds inside ray_trace (scene creation included, as in the reference's binary), Msamples/s.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../host/solstrale.hpp"

using namespace solstrale;

namespace {

std::shared_ptr<const RgbImage> read_ppm(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::string magic;
  f >> magic;
  if (magic != "P6") throw std::runtime_error(path + ": not a binary PPM (P6)");
  long field[3];
  for (long& v : field) {  // width, height, maximum - comment lines may stand between them
    for (;;) {
      f >> std::ws;
      if (f.peek() != '#') break;
      std::string comment;
      std::getline(f, comment);
    }
    if (!(f >> v)) throw std::runtime_error(path + ": bad PPM header");
  }
  if (field[0] < 1 || field[1] < 1 || field[0] > 65536 || field[1] > 65536 || field[2] != 255) throw std::runtime_error(path + ": unsupported PPM (8-bit RGB only)");
  f.get();  // the single white-space byte before the pixels
  auto img = std::make_shared<RgbImage>();
  img->width = (uint32_t)field[0];
  img->height = (uint32_t)field[1];
  img->data.resize((size_t)img->width * img->height * 3);
  f.read((char*)img->data.data(), (std::streamsize)img->data.size());
  if ((size_t)f.gcount() != img->data.size()) throw std::runtime_error(path + ": truncated PPM");
  return img;
}

std::shared_ptr<const RgbImage> checker(uint32_t n) {
  auto img = std::make_shared<RgbImage>();
  img->width = img->height = n;
  img->data.resize((size_t)n * n * 3);
  for (uint32_t y = 0; y < n; ++y)
    for (uint32_t x = 0; x < n; ++x) {
      const bool on = ((x / 32) ^ (y / 32)) & 1u;
      uint8_t* p = &img->data[((size_t)y * n + x) * 3];
      p[0] = on ? 220 : 40;
      p[1] = on ? 200 : 60;
      p[2] = on ? 160 : 90;
    }
  return img;
}

void append(std::vector<Hittables>& world, std::vector<Hittables> more) {
  for (auto& h : more) world.push_back(std::move(h));
}

// create_test_scene, tests/scenes.rs:17-122 (the constants are the reference's fixture; the order of `world` decides the tree)
Scene test_scene(const RenderConfig& config, std::shared_ptr<const RgbImage> ground_image) {
  const NopTransformer nop;
  const Materials ground = Lambertian::create(ImageMap::create(std::move(ground_image)));
  const Materials glass = Dielectric::create(SolidColor::create(1., 1., 1.), nullptr, 1.5);
  const Materials light = DiffuseLight::create(10., 10., 10.);
  const Materials red = Lambertian::create(SolidColor::create(1., 0., 0.));

  std::vector<Hittables> world;
  world.push_back(Quad::create({-5., 0., -15.}, {20., 0., 0.}, {0., 0., 20.}, ground, nop));
  world.push_back(Sphere::create({-1., 1., 0.}, 1., glass));
  append(world, Quad::new_box({0., 0., -.5}, {1., 2., .5}, red, RotationY(15.)));
  world.push_back(ConstantMedium::create(Bvh::create(Quad::new_box({0., 0., -.5}, {1., 2., .5}, red, Translation({0., 0., 1.}))), 0.1, {1., 1., 1.}));
  append(world, Quad::new_box({-1., 2., 0.}, {-.5, 2.5, .5}, red, nop));

  std::vector<Hittables> slivers;  // 5 x 5 x 5 thin triangles in a Bvh of their own
  for (int ii = 0; ii < 10; ii += 2)
    for (int jj = 0; jj < 10; jj += 2)
      for (int kk = 0; kk < 10; kk += 2) {
        const double i = ii * 0.1, j = jj * 0.1, k = kk * 0.1;
        slivers.push_back(Triangle::create({i, j + 0.05, k + 0.8}, {i, j, k + 0.8}, {i, j + 0.05, k}, red, nop));
      }
  world.push_back(Bvh::create(std::move(slivers)));
  world.push_back(Triangle::create({1., 0.1, 2.}, {3., 0.1, 2.}, {2., 0.1, 1.}, red, nop));

  // lights: a sphere, a rotated and lifted quad, a triangle
  world.push_back(Sphere::create({10., 5., 10.}, 10., light));
  const Transformations lifted({std::make_shared<RotationY>(45.), std::make_shared<Translation>(Vec3{-1., 10., -1.})});
  world.push_back(Quad::create({0., 0., 0.}, {2., 0., 0.}, {0., 0., 2.}, light, lifted));
  world.push_back(Triangle::create({-2., 1., -3.}, {0., 1., -3.}, {-1., 2., -3.}, light, nop));

  Scene scene;
  scene.world = Bvh::create(std::move(world));
  scene.camera = CameraConfig{20., 0.1, {-5., 3., 6.}, {.25, 1., 0.}, {0., 1., 0.}};
  scene.background_color = {.2, .3, .5};
  scene.render_config = config;
  return scene;
}

}  // namespace

int main(int argc, char** argv) {
  RenderConfig config;  // src/bin/profiling.rs:15-20
  config.width = 800;
  config.height = 400;
  config.samples_per_pixel = 1000;
  std::string texture, out = "out.ppm";
  int repeat = 1;
  std::vector<int> devices{0};
  for (int a = 1; a < argc; ++a) {
    const std::string k = argv[a];
    auto value = [&]() -> const char* {
      if (a + 1 >= argc) { std::fprintf(stderr, "profiling: %s needs a value\n", k.c_str()); std::exit(2); }
      return argv[++a];
    };
    if (k == "--width") config.width = (size_t)std::strtoul(value(), nullptr, 10);
    else if (k == "--height") config.height = (size_t)std::strtoul(value(), nullptr, 10);
    else if (k == "--spp") config.samples_per_pixel = (uint32_t)std::strtoul(value(), nullptr, 10);
    else if (k == "--seed") config.seed = std::strtoull(value(), nullptr, 0);
    else if (k == "--device") devices = {std::atoi(value())};
    else if (k == "--devices") {
      devices.clear();
      for (const char* p = value(); *p;) { devices.push_back(std::atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; }
    }
    else if (k == "--texture") texture = value();
    else if (k == "--out") out = value();
    else if (k == "--repeat") repeat = std::atoi(value());
    else { std::fprintf(stderr, "profiling: unknown option %s\n", k.c_str()); return 2; }
  }
  try {
    const Scene scene = test_scene(config, texture.empty() ? checker(512) : read_ppm(texture));
    RenderProgress last;
    uint32_t events = 0;
    double best = 1e300;
    for (int r = 0; r < std::max(repeat, 1); ++r) {  // (--repeat: the first ray_trace of a process also loads the code objects)
      events = 0;
      const auto t0 = std::chrono::steady_clock::now();
      const std::string err = ray_trace(scene, [&](RenderProgress&& p) {
        ++events;
        if (p.has_image) last = std::move(p);
      }, nullptr, devices);
      if (!err.empty()) { std::fprintf(stderr, "profiling: %s\n", err.c_str()); return 1; }
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    if (!last.has_image) { std::fprintf(stderr, "profiling: ray_trace produced no image\n"); return 1; }
    std::ofstream f(out, std::ios::binary);
    f << "P6\n" << last.width << " " << last.height << "\n255\n";
    f.write((const char*)last.render_image.data(), (std::streamsize)last.render_image.size());
    if (!f) { std::fprintf(stderr, "profiling: cannot write %s\n", out.c_str()); return 1; }
    const double samples = (double)config.width * (double)config.height * (double)config.samples_per_pixel;
    std::printf("{\"workload\": \"reference profiling binary (create_test_scene)\", \"width\": %zu, \"height\": %zu, \"spp\": %u, \"progress_events\": %u, "
                "\"ray_trace_s\": %.4f, \"msamples_per_s\": %.1f, \"devices\": %zu, \"texture\": \"%s\", \"image\": \"%s\"}\n",
                config.width, config.height, config.samples_per_pixel, events, best, samples / best / 1e6, devices.size(),
                texture.empty() ? "procedural checker" : texture.c_str(), out.c_str());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "profiling: %s\n", e.what());
    return 1;
  }
  return 0;
}
