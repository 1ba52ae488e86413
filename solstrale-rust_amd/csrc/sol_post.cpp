// sol_post.cpp -- what follows a render on the device: un-permute of tile buffers into the row-major image, the Nop tone-map
// (src/post/nop.rs, src/util/rgb_color.rs:14-35) and BloomPostProcessor (src/post/bloom.rs:76-150); kernels in sol_aux.hip.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "sol_scene.h"

extern "C" {

int sol_unpermute(SolScene* s, const void* gathered, int world, void* image) {
  if (!s || !gathered || !image) return sol_fail(SOL_EINVAL, "null argument");
  if (world != s->world) return sol_fail(SOL_EINVAL, "world %d differs from the scene's partition (%d)", world, s->world);
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(sol_launch_unpermute((const float*)gathered, (float*)image, s->S.width, s->S.height, s->blocks_x, (uint32_t)world,
                               0xFFFFFFFFu, s->acc_floats, s->slot_of_block, s->stream));
  return SOL_OK;
}

int sol_tonemap_rgb8(SolScene* s, const void* image, uint32_t spp, uint8_t* out) {
  if (!s || !image || !out || spp == 0) return sol_fail(SOL_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(s->device));
  const uint32_t n = s->S.width * s->S.height * 3;
  HIP_TRY(sol_launch_tonemap((const float*)image, s->rgb8, n, spp, s->stream));
  HIP_TRY(hipMemcpyAsync(out, s->rgb8, n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return SOL_OK;
}

int sol_resolve_image(SolScene* s, void** image_dev) {
  if (!s || !image_dev) return sol_fail(SOL_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(sol_launch_unpermute(s->acc, s->image, s->S.width, s->S.height, s->blocks_x, (uint32_t)s->world, (uint32_t)s->rank,
                               s->acc_floats, s->slot_of_block, s->stream));
  *image_dev = s->image;
  return SOL_OK;
}

}  // extern "C"

// create_gaussian_blur_weights (src/util/gaussian.rs:3-25)
static std::vector<double> gaussian_blur_weights(size_t kernel_size, double std_dev) {
  std::vector<double> w(kernel_size);
  const double mean = (double)(kernel_size - 1) / 2.0;
  double sum = 0.0;
  for (size_t i = 0; i < kernel_size; ++i) {
    const double a = ((double)i - mean) / std_dev;
    w[i] = std::exp(-0.5 * a * a);
  }
  for (size_t i = 0; i < kernel_size; ++i) sum += w[i];  // iter().sum(): left to right from 0.0
  for (size_t i = 0; i < kernel_size; ++i) w[i] /= sum;
  return w;
}

extern "C" {

int sol_gaussian_blur_weights(uint32_t kernel_size, double std_dev, double* out) {
  if (!out || kernel_size == 0) return sol_fail(SOL_EINVAL, "bad argument");
  std::vector<double> w = gaussian_blur_weights(kernel_size, std_dev);
  std::memcpy(out, w.data(), w.size() * sizeof(double));
  return SOL_OK;
}

}  // extern "C"

static int bloom_impl(SolScene* s, void* image, uint32_t spp, double ksf, double threshold, double max_intensity, uint8_t* out) {
  if (!s || !image || spp == 0) return sol_fail(SOL_EINVAL, "bad argument");
  if (!(ksf >= 0.0 && ksf <= 0.5)) return sol_fail(SOL_EINVAL, "kernel_size_fraction must be between 0 and 0.5");  // bloom.rs:33-37
  HIP_TRY(hipSetDevice(s->device));
  const uint32_t W = s->S.width, H = s->S.height;
  const size_t n = (size_t)W * H * 3;
  // bloom.rs:86-91
  const double thr = threshold * (double)spp, maxi = max_intensity * (double)spp;
  const size_t k = (size_t)(ksf * (double)W) * 2 + 1;
  std::vector<double> w = gaussian_blur_weights(k, (double)k / 5.0);
  if (!s->bloom_a) HIP_TRY(hipMalloc((void**)&s->bloom_a, n * sizeof(double)));
  if (!s->bloom_b) HIP_TRY(hipMalloc((void**)&s->bloom_b, n * sizeof(double)));
  if (k > s->bloom_w_cap) {
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->bloom_w) hipFree(s->bloom_w);
    s->bloom_w = nullptr; s->bloom_w_cap = 0;
    HIP_TRY(hipMalloc((void**)&s->bloom_w, k * sizeof(double)));
    s->bloom_w_cap = k;
  }
  HIP_TRY(hipMemcpyAsync(s->bloom_w, w.data(), k * sizeof(double), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));  // `w` is pageable host memory about to go out of scope
  HIP_TRY(sol_launch_bloom((float*)image, s->bloom_a, s->bloom_b, s->bloom_w, (uint32_t)k, W, H, thr, maxi, out ? s->rgb8 : nullptr, spp,
                           s->stream));
  if (out) {
    HIP_TRY(hipMemcpyAsync(out, s->rgb8, n, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  return SOL_OK;
}

extern "C" {

int sol_bloom(SolScene* s, void* image, uint32_t spp, double ksf, double threshold, double max_intensity) {
  return bloom_impl(s, image, spp, ksf, threshold, max_intensity, nullptr);
}
int sol_bloom_rgb8(SolScene* s, const void* image, uint32_t spp, double ksf, double threshold, double max_intensity, uint8_t* out) {
  if (!out) return sol_fail(SOL_EINVAL, "bad argument");
  return bloom_impl(s, const_cast<void*>(image), spp, ksf, threshold, max_intensity, out);
}

}  // extern "C"
