// sol_aux.hip -- the small kernels around the render kernel: chunk resolve, tile un-permute (rank-0 side of the multi-GPU
// gather), the Nop post-processor, and the function-level evaluation hook used by the parity tests.
#include <hip/hip_runtime.h>

#include "sol_launch.h"
#include "sol_shade.h"

// acc[i] += sum over chunks (in chunk order) of partial[chunk][i]
__global__ void __launch_bounds__(256) sol_resolve_kernel(float* __restrict__ acc, const float* __restrict__ partial,
                                                          uint32_t n_floats, uint32_t n_chunks) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_floats; i += gridDim.x * blockDim.x) {
    float s = acc[i];  // ((acc + c0) + c1) + ...: the same association whether the chunks come from one call or several
    for (uint32_t k = 0; k < n_chunks; ++k) s += partial[(size_t)k * n_floats + i];
    acc[i] = s;
  }
}

// compact per-rank tile buffers -> row-major image (row 0 = top). gathered = world buffers of `stride` floats.
__global__ void __launch_bounds__(256) sol_unpermute_kernel(const float* __restrict__ gathered, float* __restrict__ image,
                                                            uint32_t width, uint32_t height, uint32_t blocks_x, uint32_t world,
                                                            uint32_t only_rank, size_t stride, const uint32_t* __restrict__ slot_of_block) {
  const uint32_t npix = width * height;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
    const uint32_t y = p / width, x = p - y * width;
    const uint32_t b = (y / SOL_TILE) * blocks_x + (x / SOL_TILE);
    uint32_t r = b % world, lb = b / world;
    if (slot_of_block) {  // balanced partition: owner * blocks per buffer + local block
      const uint32_t per = (uint32_t)(stride / 192u), sl = slot_of_block[b];
      r = sl / per; lb = sl - r * per;
    }
    const uint32_t slot = lb * 64u + (y % SOL_TILE) * SOL_TILE + (x % SOL_TILE);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (only_rank == 0xFFFFFFFFu || only_rank == r) {
      const float* src = gathered + (only_rank == 0xFFFFFFFFu ? (size_t)r * stride : 0) + (size_t)slot * 3;
      v0 = src[0]; v1 = src[1]; v2 = src[2];
    }
    image[(size_t)p * 3] = v0; image[(size_t)p * 3 + 1] = v1; image[(size_t)p * 3 + 2] = v2;
  }
}

// Nop post-processor: to_rgb_color (src/util/rgb_color.rs:14-35). The reference computes it in f64 from f64 sums; the
// device sums are fp32, converted to double here so the rounding of sqrt/clamp/scale matches the host arithmetic.
__global__ void __launch_bounds__(256) sol_tonemap_kernel(const float* __restrict__ image, uint8_t* __restrict__ rgb,
                                                          uint32_t n, uint32_t spp) {
  const double scale = 1.0 / (double)spp;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double v = sqrt(scale * (double)image[i]);
    if (v < -0.999) v = -0.999;
    if (v > 0.999) v = 0.999;
    double sc = 256.0 * v;
    rgb[i] = isnan(sc) ? (uint8_t)0 : (uint8_t)(sc < 0.0 ? 0.0 : (sc > 255.0 ? 255.0 : sc));
  }
}

// ---- BloomPostProcessor (src/post/bloom.rs:76-150) in f64, the reference's arithmetic and summation order ----------------
// Pixel sums are the device's fp32 sums promoted to double. Buffers `a`, `b` hold W*H*3 doubles.
// 1. bright pass (bloom.rs:93-106): keep a pixel whose length reaches the threshold, scaled back to max_intensity.
__global__ void __launch_bounds__(256) sol_bloom_bright_kernel(const float* __restrict__ image, double* __restrict__ out, uint32_t npix,
                                                               double threshold, double max_intensity) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
    const double x = (double)image[(size_t)p * 3], y = (double)image[(size_t)p * 3 + 1], z = (double)image[(size_t)p * 3 + 2];
    const double len = sqrt(x * x + y * y + z * z);  // Vec3::length (vec3.rs:282-295)
    double ox = 0.0, oy = 0.0, oz = 0.0;
    if (len >= threshold) {
      if (len > max_intensity) {  // p.unit() * max_intensity
        ox = (x / len) * max_intensity; oy = (y / len) * max_intensity; oz = (z / len) * max_intensity;
      } else {
        ox = x; oy = y; oz = z;
      }
    }
    out[(size_t)p * 3] = ox; out[(size_t)p * 3 + 1] = oy; out[(size_t)p * 3 + 2] = oz;
  }
}
// 2./3. separable blur (bloom.rs:108-142): col += get_pixel_safe(..) * weights[i] for i = 0..k-1 in this order, coordinates
// clamped to the image (bloom.rs:153-158). VERTICAL selects the axis. Consecutive lanes read consecutive pixels for every tap
// (coalesced 24-B records); the k-fold re-reads are served by L1/L2.
template <bool VERTICAL>
__global__ void __launch_bounds__(256) sol_bloom_blur_kernel(const double* __restrict__ in, double* __restrict__ out, uint32_t width,
                                                             uint32_t height, const double* __restrict__ weights, uint32_t k) {
  const int half = (int)(k / 2);
  const uint32_t npix = width * height;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
    const int x = (int)(p % width), y = (int)(p / width);
    double cx = 0.0, cy = 0.0, cz = 0.0;
    for (uint32_t i = 0; i < k; ++i) {
      int sx = x, sy = y;
      if (VERTICAL) sy = min(max(y + (int)i - half, 0), (int)height - 1);
      else sx = min(max(x + (int)i - half, 0), (int)width - 1);
      const double* s = in + ((size_t)sy * width + sx) * 3;
      const double w = weights[i];
      cx = cx + s[0] * w; cy = cy + s[1] * w; cz = cz + s[2] * w;
    }
    out[(size_t)p * 3] = cx; out[(size_t)p * 3 + 1] = cy; out[(size_t)p * 3 + 2] = cz;
  }
}
// The same blur, LDS-tiled (round 3): one workgroup owns SOL_BLUR_TILE consecutive outputs of one row, loads them and their k - 1
// neighbours ONCE into LDS (coordinates clamped to the row: get_pixel_safe) and every thread forms SOL_BLUR_PER_THREAD adjacent outputs
// from it - an input value read from LDS feeds that many accumulators. Each output is still  col += pixel * w[i]  for i = 0 .. k-1
// in this order, in f64: bit-identical to the kernel above (tests/test_post.py compares both with the numpy restatement). The
// vertical pass runs as transpose -> the same row kernel -> transpose (two 200-MB copies at 4K against k-fold strided reads).
// MI355X, tests/tools/bloom_bench.py: 4K, k = 1537: 40.9 -> 11.4 ms; 1080p, k = 769: 3.55 -> 1.77 ms ("Small kernels").
#define SOL_BLUR_PER_THREAD 4
#define SOL_BLUR_TILE (256 * SOL_BLUR_PER_THREAD)
#define SOL_BLUR_MAX_K 1665  // (SOL_BLUR_TILE + k - 1) * 24 bytes must fit 64 KiB of LDS; longer kernels use the untiled form
__global__ void __launch_bounds__(256) sol_bloom_blur_rows_kernel(const double* __restrict__ in, double* __restrict__ out, uint32_t width,
                                                                  uint32_t height, const double* __restrict__ weights, uint32_t k) {
  extern __shared__ double tile[];  // (SOL_BLUR_TILE + k - 1) pixels x 3
  const int half = (int)(k / 2);
  const uint32_t tiles_x = (width + SOL_BLUR_TILE - 1) / SOL_BLUR_TILE;
  const uint32_t row = blockIdx.x / tiles_x, x0 = (blockIdx.x - row * tiles_x) * SOL_BLUR_TILE;
  const uint32_t span = SOL_BLUR_TILE + k - 1;
  const double* src = in + (size_t)row * width * 3;
  for (uint32_t j = threadIdx.x; j < span; j += 256u) {
    const int sx = min(max((int)x0 + (int)j - half, 0), (int)width - 1);
    tile[3 * j] = src[(size_t)sx * 3]; tile[3 * j + 1] = src[(size_t)sx * 3 + 1]; tile[3 * j + 2] = src[(size_t)sx * 3 + 2];
  }
  __syncthreads();
  // thread t forms outputs x0 + t + 256 * m, m = 0 .. PER_THREAD-1 (adjacent threads, adjacent LDS addresses)
  double acc[SOL_BLUR_PER_THREAD][3];
#pragma unroll
  for (int m = 0; m < SOL_BLUR_PER_THREAD; ++m) acc[m][0] = acc[m][1] = acc[m][2] = 0.0;
  for (uint32_t i = 0; i < k; ++i) {
    const double w = weights[i];
#pragma unroll
    for (int m = 0; m < SOL_BLUR_PER_THREAD; ++m) {
      const double* s = tile + 3 * (threadIdx.x + 256u * m + i);
      acc[m][0] = acc[m][0] + s[0] * w; acc[m][1] = acc[m][1] + s[1] * w; acc[m][2] = acc[m][2] + s[2] * w;
    }
  }
#pragma unroll
  for (int m = 0; m < SOL_BLUR_PER_THREAD; ++m) {
    const uint32_t x = x0 + threadIdx.x + 256u * m;
    if (x < width) {
      double* o = out + ((size_t)row * width + x) * 3;
      o[0] = acc[m][0]; o[1] = acc[m][1]; o[2] = acc[m][2];
    }
  }
}
// out[x][y] = in[y][x] for W x H pixels of three doubles (32 x 32 tiles through LDS)
__global__ void __launch_bounds__(256) sol_transpose3_kernel(const double* __restrict__ in, double* __restrict__ out, uint32_t width, uint32_t height) {
  __shared__ double t[32][33 * 3];
  const uint32_t bx = blockIdx.x * 32u, by = blockIdx.y * 32u, tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;
  for (uint32_t r = ty; r < 32u; r += 8u)
    if (bx + tx < width && by + r < height) {
      const double* s = in + ((size_t)(by + r) * width + bx + tx) * 3;
      t[r][3 * tx] = s[0]; t[r][3 * tx + 1] = s[1]; t[r][3 * tx + 2] = s[2];
    }
  __syncthreads();
  for (uint32_t r = ty; r < 32u; r += 8u)
    if (by + tx < height && bx + r < width) {
      double* o = out + ((size_t)(bx + r) * height + by + tx) * 3;
      o[0] = t[tx][3 * r]; o[1] = t[tx][3 * r + 1]; o[2] = t[tx][3 * r + 2];
    }
}
// 4. pixel + blurred (bloom.rs:144-148); then either back to the fp32 image (intermediate_post_process, rounded to fp32:
// the device keeps fp32 sums) or straight through to_rgb_color in f64 (post_process).
__global__ void __launch_bounds__(256) sol_bloom_finish_kernel(float* __restrict__ image, const double* __restrict__ blurred,
                                                               uint8_t* __restrict__ rgb, uint32_t n, uint32_t spp) {
  const double scale = 1.0 / (double)spp;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double sum = (double)image[i] + blurred[i];
    if (rgb) {
      double v = sqrt(scale * sum);
      if (v < -0.999) v = -0.999;
      if (v > 0.999) v = 0.999;
      double sc = 256.0 * v;
      rgb[i] = isnan(sc) ? (uint8_t)0 : (uint8_t)(sc < 0.0 ? 0.0 : (sc > 255.0 ? 255.0 : sc));
    } else {
      image[i] = (float)sum;
    }
  }
}

hipError_t sol_launch_bloom(float* image, double* a, double* b, const double* weights, uint32_t k, uint32_t width, uint32_t height,
                            double threshold, double max_intensity, uint8_t* rgb, uint32_t spp, hipStream_t stream) {
  const uint32_t npix = width * height;
  uint32_t grid = (npix + 255u) / 256u;
  if (grid > 8192u) grid = 8192u;
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL(sol_bloom_bright_kernel, dim3(grid), dim3(256), 0, stream, image, a, npix, threshold, max_intensity);
  static const bool untiled = false;  // (the untiled kernels remain for kernels longer than SOL_BLUR_MAX_K taps)
  if (!untiled && k <= SOL_BLUR_MAX_K) {
    const size_t lds = (size_t)(SOL_BLUR_TILE + k - 1) * 3 * sizeof(double);
    hipError_t e = hipFuncSetAttribute((const void*)sol_bloom_blur_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e != hipSuccess) return e;
    const uint32_t tiles_w = (width + SOL_BLUR_TILE - 1) / SOL_BLUR_TILE, tiles_h = (height + SOL_BLUR_TILE - 1) / SOL_BLUR_TILE;
    // rows: a -> b; columns: b^T (= a) -> rows of the transposed image -> b, transposed back -> a
    hipLaunchKernelGGL(sol_bloom_blur_rows_kernel, dim3(tiles_w * height), dim3(256), lds, stream, a, b, width, height, weights, k);
    hipLaunchKernelGGL(sol_transpose3_kernel, dim3((width + 31u) / 32u, (height + 31u) / 32u), dim3(256), 0, stream, b, a, width, height);
    hipLaunchKernelGGL(sol_bloom_blur_rows_kernel, dim3(tiles_h * width), dim3(256), lds, stream, a, b, height, width, weights, k);
    hipLaunchKernelGGL(sol_transpose3_kernel, dim3((height + 31u) / 32u, (width + 31u) / 32u), dim3(256), 0, stream, b, a, height, width);
  } else {
    hipLaunchKernelGGL((sol_bloom_blur_kernel<false>), dim3(grid), dim3(256), 0, stream, a, b, width, height, weights, k);
    hipLaunchKernelGGL((sol_bloom_blur_kernel<true>), dim3(grid), dim3(256), 0, stream, b, a, width, height, weights, k);
  }
  hipLaunchKernelGGL(sol_bloom_finish_kernel, dim3(grid), dim3(256), 0, stream, image, a, rgb, npix * 3u, spp);
  return hipGetLastError();
}

// ---- launch wrappers (called from sol_api.cpp) ----
hipError_t sol_launch_resolve(float* acc, const float* partial, uint32_t n_floats, uint32_t n_chunks, hipStream_t stream) {
  uint32_t grid = (n_floats + 255u) / 256u;
  if (grid > 4096u) grid = 4096u;
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL(sol_resolve_kernel, dim3(grid), dim3(256), 0, stream, acc, partial, n_floats, n_chunks);
  return hipGetLastError();
}

hipError_t sol_launch_unpermute(const float* gathered, float* image, uint32_t width, uint32_t height, uint32_t blocks_x,
                                uint32_t world, uint32_t only_rank, size_t stride, const uint32_t* slot_of_block, hipStream_t stream) {
  uint32_t grid = (width * height + 255u) / 256u;
  if (grid > 4096u) grid = 4096u;
  hipLaunchKernelGGL(sol_unpermute_kernel, dim3(grid), dim3(256), 0, stream, gathered, image, width, height, blocks_x, world,
                     only_rank, stride, slot_of_block);
  return hipGetLastError();
}

hipError_t sol_launch_tonemap(const float* image, uint8_t* rgb, uint32_t n, uint32_t spp, hipStream_t stream) {
  uint32_t grid = (n + 255u) / 256u;
  if (grid > 4096u) grid = 4096u;
  hipLaunchKernelGGL(sol_tonemap_kernel, dim3(grid), dim3(256), 0, stream, image, rgb, n, spp);
  return hipGetLastError();
}

// ---- function-level evaluation (sol_eval): the device functions of sol_math.h / sol_trace.h on arrays of inputs, so that
// tests can pin the fp32 arithmetic contract bit for bit against the CPU restatement -------------------------------------
__global__ void __launch_bounds__(256) sol_eval_kernel(uint32_t fn, const float* __restrict__ in, uint32_t n, uint32_t is,
                                                       float* __restrict__ out, uint32_t os) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* x = in + (size_t)i * is;
  float* y = out + (size_t)i * os;
  switch (fn) {
    case 0: {  // arithmetic: a/b, sqrt|a|, 1/a, a*b+c (unfused), fmax, fmin, floor
      float a = x[0], b = x[1], c = x[2];
      y[0] = a / b; y[1] = sol_sqrt(fabsf(a)); y[2] = 1.0f / a; y[3] = a * b + c; y[4] = fmaxf(a, b); y[5] = fminf(a, b);
      y[6] = floorf(a);
      break;
    }
    case 1: {  // elementary functions: r in [0,1), x in [-1,1], y
      float c, s;
      sincos2pi(x[0], c, s);
      y[0] = c; y[1] = s; y[2] = acos_r(x[1]); y[3] = atan2_r(x[2], x[1]); y[4] = log_r(x[0]);
      break;
    }
    case 2: {  // rng: seed_lo, seed_hi, pixel, sample, counter (bit patterns)
      Rng r;
      rng_init(r, __float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
      y[0] = __uint_as_float(rng_bits(r, __float_as_uint(x[4])));
      y[1] = u32_to_unit(rng_bits(r, __float_as_uint(x[4])));
      break;
    }
    case 3: {  // vectors: v(3), n(3), ior -> unit(v), reflect(v,n), refract(v,n,ior), onb_new(v)
      f3 v = mk3(x[0], x[1], x[2]), nn = mk3(x[3], x[4], x[5]);
      f3 u = unit3(v), rf = reflect3(v, nn), rr = refract3(v, nn, x[6]);
      Onb o = onb_new(v);
      y[0] = u.x; y[1] = u.y; y[2] = u.z; y[3] = rf.x; y[4] = rf.y; y[5] = rf.z; y[6] = rr.x; y[7] = rr.y; y[8] = rr.z;
      y[9] = o.tangent.x; y[10] = o.tangent.y; y[11] = o.tangent.z; y[12] = o.bi_tangent.x; y[13] = o.bi_tangent.y;
      y[14] = o.bi_tangent.z; y[15] = o.normal.x; y[16] = o.normal.y; y[17] = o.normal.z;
      break;
    }
    case 4: {  // sphere: center(3), radius, o(3), d(3), tmin, tmax -> hit, t
      DSphere S; S.cx = x[0]; S.cy = x[1]; S.cz = x[2]; S.radius = x[3]; S.dfs = 0; S.mat = 0; S.pad0 = S.pad1 = 0;
      float t = 0.f;
      bool h = sphere_test(S, mk3(x[4], x[5], x[6]), mk3(x[7], x[8], x[9]), x[10], x[11], x[12], t);
      y[0] = h ? 1.f : 0.f; y[1] = h ? t : 0.f;
      break;
    }
    case 5: {  // quad: n(3), d, q(3), w(3), u(3), v(3), o(3), dir(3), tmin, tmax -> hit, t, u, v
      DQuad Q; Q.nx = x[0]; Q.ny = x[1]; Q.nz = x[2]; Q.d = x[3]; Q.qx = x[4]; Q.qy = x[5]; Q.qz = x[6];
      Q.wx = x[7]; Q.wy = x[8]; Q.wz = x[9]; Q.ux = x[10]; Q.uy = x[11]; Q.uz = x[12]; Q.vx = x[13]; Q.vy = x[14]; Q.vz = x[15];
      Q.dfs = 0; Q.mat = 0; Q.area = 0; Q.pad = 0;
      float t = 0.f, u = 0.f, v = 0.f;
      bool h = quad_test(Q, mk3(x[16], x[17], x[18]), mk3(x[19], x[20], x[21]), x[22], x[23], t, u, v);
      y[0] = h ? 1.f : 0.f; y[1] = h ? t : 0.f; y[2] = h ? u : 0.f; y[3] = h ? v : 0.f;
      break;
    }
    case 6: {  // triangle: v0(3), e1(3), e2(3), o(3), dir(3), tmin, tmax -> hit, t, u, v
      DTri T; T.v0x = x[0]; T.v0y = x[1]; T.v0z = x[2]; T.e1x = x[3]; T.e1y = x[4]; T.e1z = x[5]; T.e2x = x[6]; T.e2y = x[7];
      T.e2z = x[8]; T.dfs = 0; T.mat = 0; T.area = 0.f;
      float t = 0.f, u = 0.f, v = 0.f;
      bool h = tri_test(T, mk3(x[9], x[10], x[11]), mk3(x[12], x[13], x[14]), x[15], x[16], t, u, v);
      y[0] = h ? 1.f : 0.f; y[1] = h ? t : 0.f; y[2] = h ? u : 0.f; y[3] = h ? v : 0.f;
      break;
    }
    case 7: {  // slab: box(6: xmin,xmax,ymin,ymax,zmin,zmax), o(3), dir(3) -> hit, t_entry
      f3 d = mk3(x[9], x[10], x[11]);
      f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
      float te = 0.f;
      bool h = slab(x[0], x[1], x[2], x[3], x[4], x[5], mk3(x[6], x[7], x[8]), inv, __builtin_signbitf(inv.x),
                    __builtin_signbitf(inv.y), __builtin_signbitf(inv.z), te);
      y[0] = h ? 1.f : 0.f; y[1] = te;
      break;
    }
    case 8: {  // sampling: seed bits, pixel, sample -> random_cosine_direction (3), random_in_unit_sphere (3), counter after
      Rng r;
      rng_init(r, __float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
      f3 c = random_cosine_direction(r);
      f3 s = random_in_unit_sphere(r);
      y[0] = c.x; y[1] = c.y; y[2] = c.z; y[3] = s.x; y[4] = s.y; y[5] = s.z; y[6] = __uint_as_float(r.ctr);
      break;
    }
    default: break;
  }
}

hipError_t sol_launch_eval(uint32_t fn, const float* in, uint32_t n, uint32_t is, float* out, uint32_t os, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(sol_eval_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, fn, in, n, is, out, os);
  return hipGetLastError();
}
