// sol_kernels.hip -- the gfx950 kernels of the path-tracing hot path (wave64, no MFMA: branchy traversal + fp32
// shading). One persistent kernel carries a path from camera ray to termination:
//
//   regenerate : lanes without a path take the next sample of their work item; lanes without a work item fetch one
//                through a wave-aggregated atomic (ballot + popcount prefix, one atomic per wave);
//   intersect  : ordered BVH traversal with an LDS-resident per-lane stack (sol_trace.h);
//   shade      : material scatter, light/BSDF mixture pdf, throughput update or termination (sol_shade.h).
//
// Every lane owns one (pixel, 16-sample chunk) work item at a time and sums its samples in registers in sample order,
// so no float atomics touch the accumulator and the image is a pure function of (scene, seed), bit-identical for any
// tile partition. Replaces src/renderer/mod.rs:241-291 (row tasks) and everything under ray_color (:164-206).
#include <hip/hip_runtime.h>

#include "sol_launch.h"
#include "sol_shade.h"

// The recursion ray_color <-> shade (src/renderer/shader.rs:62-125) flattened exactly: with per-level factors
// a_d >= 0, a composition of  x -> a*x  (ScatterBasic) and  x -> min(a*x, 3) with NaN -> 0  (ScatterPdf) is always
// x -> min(A*x, C); a Pdf level does C <- min(C, 3*A), A <- A*a; a Basic level does A <- A*a.
struct PathState {
  f3 o, d, inv;
  f3 A, C;
  float acc_len;
  uint32_t depth;
  bool pdf_seen;
};

template <bool COUNT, bool MEDIUM>
__global__ void __launch_bounds__(SOL_WG)
sol_render_kernel(const DevScene S, const RenderParams P, float* __restrict__ acc, float* __restrict__ partial,
                  uint32_t* __restrict__ work_counter, uint32_t* __restrict__ spill, DevCounters* __restrict__ dcnt) {
  __shared__ uint32_t lds_stack[SOL_LDS_STACK * SOL_WG];
  const uint32_t tid = threadIdx.x;
  const uint32_t gtid = blockIdx.x * SOL_WG + tid;
  const uint32_t lane = tid & 63u;
  Stack st;
  st.lds = lds_stack + tid;
  st.spill = spill + gtid;
  st.stride = P.total_threads;
  Counters cnt = {};
  const float inf = __builtin_huge_valf();
  const uint32_t slots = P.n_local_blocks * 64u;

  bool have_item = false, alive = false;
  uint32_t px = 0, py = 0, slot = 0, chunk = 0, s = 0, s_end = 0;
  f3 sum = mk3(0.f, 0.f, 0.f);
  Rng rng = {0, 0, 0};
  PathState ps = {};

  for (;;) {
    // ---- work fetch: one atomic per wave, lanes take consecutive items (an aligned wave = one 8x8 pixel block) ----
    if (!have_item) {
      const unsigned long long need = __ballot(1);
      const uint32_t leader = (uint32_t)__ffsll((long long)need) - 1u;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(work_counter, (uint32_t)__popcll(need));
      base = __shfl(base, (int)leader);
      const uint32_t item = base + (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
      if (item >= P.n_items) break;  // no work left for this lane
      chunk = item / slots;
      slot = item - chunk * slots;
      const uint32_t lb = slot >> 6, pin = slot & 63u;
      const uint32_t b = lb * P.world + P.rank;
      const uint32_t by = b / P.blocks_x, bx = b - by * P.blocks_x;
      px = bx * SOL_TILE + (pin & 7u);
      py = by * SOL_TILE + (pin >> 3);
      if (px >= S.width || py >= S.height) continue;  // padding pixel of an edge block
      s = P.first_sample + chunk * SOL_CHUNK;
      s_end = min(s + SOL_CHUNK, P.first_sample + P.n_samples);
      sum = mk3(0.f, 0.f, 0.f);
      have_item = true;
      alive = false;
    }
    // ---- generate: pixel jitter + Camera::get_ray (src/renderer/mod.rs:263-265, src/camera.rs:77-89) ----
    if (!alive) {
      rng_init(rng, P.seed_lo, P.seed_hi, py * S.width + px, s);
      if (COUNT) cnt.samples++;
      const uint32_t y_ref = (S.height - 1u) - py;
      float u = ((float)px + rnd(rng)) / (float)(S.width - 1u);
      float v = ((float)y_ref + rnd(rng)) / (float)(S.height - 1u);
      f3 offset = mk3(0.f, 0.f, 0.f);
      if (S.cam.lens_radius > 0.0f) {
        f3 rd = mk3(0.f, 0.f, 0.f);
        for (int it = 0; it < 80; ++it) {  // random_in_unit_disc (vec3.rs:400-412)
          rd.x = rnd_range(rng, -1.0f, 1.0f);
          rd.y = rnd_range(rng, -1.0f, 1.0f);
          if (len2(rd) < 1.0f) break;
        }
        rd = rd * S.cam.lens_radius;
        offset = mk3(S.cam.ux, S.cam.uy, S.cam.uz) * rd.x + mk3(S.cam.wx, S.cam.wy, S.cam.wz) * rd.y;
      }
      const f3 org = mk3(S.cam.ox, S.cam.oy, S.cam.oz);
      ps.d = mk3(S.cam.llx, S.cam.lly, S.cam.llz) + mk3(S.cam.hx, S.cam.hy, S.cam.hz) * u +
             mk3(S.cam.vx, S.cam.vy, S.cam.vz) * v - org - offset;
      ps.o = org + offset;
      ps.inv = mk3(1.0f / ps.d.x, 1.0f / ps.d.y, 1.0f / ps.d.z);  // Ray::new (geo/mod.rs:277-285)
      ps.A = mk3(1.f, 1.f, 1.f);
      ps.C = mk3(inf, inf, inf);
      ps.acc_len = 0.0f;
      ps.depth = 0;
      ps.pdf_seen = false;
      alive = true;
    }
    // ---- intersect: world.hit(ray, RAY_INTERVAL) (src/renderer/mod.rs:165) ----
    Hit h;
    closest_hit<COUNT, MEDIUM>(S, ps.o, ps.d, ps.inv, RAY_MIN_F, inf, S.root, S.rxmin, S.rxmax, S.rymin, S.rymax, S.rzmin,
                               S.rzmax, h, st, 0, rng, ps.depth, cnt);
    if (COUNT) cnt.rays++;
    // ---- shade ----
    bool terminal = true;
    f3 x = mk3(S.bgx, S.bgy, S.bgz);  // miss: background (src/renderer/mod.rs:197-204)
    bool has_af = false;
    float af = 0.0f, path_len = 0.0f;
    if (SOL_REF_KIND(h.ref) != SOL_REF_NONE) {
      Surface sf;
      build_surface<COUNT>(S, ps.o, ps.d, h, rng, ps.depth, sf);
      sf.normal = transformed_normal<COUNT>(S, sf.mat, sf, rng, cnt);  // RayHit::new (material/mod.rs:50)
      Scatter sc;
      if (S.shader != SOL_SHADER_PATH_TRACING) {
        if (S.shader == SOL_SHADER_NORMAL) {  // shader.rs:165-172
          x = sf.normal;
        } else {  // Albedo (shader.rs:141-151), Simple (shader.rs:192-214)
          scatter<COUNT>(S, ps.d, sf, rng, sc, cnt);
          x = sc.color;
          if (S.shader == SOL_SHADER_SIMPLE && sc.type != SCATTER_EMISSION)
            x = sc.color * (dot3(sf.normal, mk3(1.f, 1.f, -1.f)) * 0.5f + 0.75f);
        }
      } else if (ps.depth >= S.max_depth) {  // shader.rs:70-72
        x = mk3(0.f, 0.f, 0.f);
      } else {
        const float total = sf.t + ps.acc_len;  // shader.rs:74
        scatter<COUNT>(S, ps.d, sf, rng, sc, cnt);
        if (sc.type == SCATTER_EMISSION) {  // shader.rs:78-84
          x = sc.color; has_af = sc.has_af; af = sc.af; path_len = total;
        } else {
          bool go_on = true;
          if (sc.type == SCATTER_PDF) {  // shader.rs:95-104 + filter :109-125
            const f3 a = sc.color * sc.probability;
            if (!(a.x > 0.0f || a.y > 0.0f || a.z > 0.0f)) {
              // a == 0 (or NaN): this level returns exactly 0 whatever lies beyond it -> the sample contributes 0
              x = mk3(0.f, 0.f, 0.f);
              go_on = false;
            } else {
              ps.C = mk3(fminf(ps.C.x, ps.A.x * 3.0f), fminf(ps.C.y, ps.A.y * 3.0f), fminf(ps.C.z, ps.A.z * 3.0f));
              ps.A = ps.A * a;
              ps.pdf_seen = true;
            }
          } else {  // ScatterBasic (shader.rs:85-94)
            ps.A = ps.A * sc.color;
          }
          if (go_on) {
            terminal = false;
            ps.o = sf.p;
            ps.d = sc.dir;
            ps.inv = mk3(1.0f / sc.dir.x, 1.0f / sc.dir.y, 1.0f / sc.dir.z);
            ps.acc_len = total;
            ps.depth++;
          }
        }
      }
    }
    if (terminal) {
      f3 c = x;
      if (S.shader == SOL_SHADER_PATH_TRACING) {
        c = ps.A * x;
        if (ps.pdf_seen) {
          c.x = isnan(c.x) ? 0.0f : fminf(c.x, ps.C.x);
          c.y = isnan(c.y) ? 0.0f : fminf(c.y, ps.C.y);
          c.z = isnan(c.z) ? 0.0f : fminf(c.z, ps.C.z);
        }
        if (has_af) c = (c * 1.0f) / (1.0f + af * path_len);  // get_attenuated_color (material/mod.rs:127-131)
      }
      sum = sum + c;  // add_row_data (src/renderer/mod.rs:361-365): sums, not means
      alive = false;
      s++;
      if (s == s_end) {
        if (P.n_chunks == 1) {
          float* a = acc + (size_t)slot * 3;
          a[0] += sum.x; a[1] += sum.y; a[2] += sum.z;
        } else {
          float* a = partial + ((size_t)chunk * slots + slot) * 3;
          a[0] = sum.x; a[1] = sum.y; a[2] = sum.z;
        }
        have_item = false;
      }
    }
  }
  if (COUNT) {
    atomicAdd(&dcnt->samples, (unsigned long long)cnt.samples);
    atomicAdd(&dcnt->rays, (unsigned long long)cnt.rays);
    atomicAdd(&dcnt->node_visits, (unsigned long long)cnt.node_visits);
    atomicAdd(&dcnt->sphere_tests, (unsigned long long)cnt.sphere_tests);
    atomicAdd(&dcnt->quad_tests, (unsigned long long)cnt.quad_tests);
    atomicAdd(&dcnt->triangle_tests, (unsigned long long)cnt.triangle_tests);
    atomicAdd(&dcnt->shades, (unsigned long long)cnt.shades);
    atomicAdd(&dcnt->texel_fetches, (unsigned long long)cnt.texel_fetches);
    atomicMax(&dcnt->max_stack, (unsigned long long)cnt.max_stack);
  }
}

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
                                                            uint32_t only_rank, size_t stride) {
  const uint32_t npix = width * height;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
    const uint32_t y = p / width, x = p - y * width;
    const uint32_t b = (y / SOL_TILE) * blocks_x + (x / SOL_TILE);
    const uint32_t r = b % world, lb = b / world;
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

// ---- launch wrappers (called from sol_api.cpp) ---------------------------------------------------------------
template <bool COUNT, bool MEDIUM>
static hipError_t launch_render_t(const DevScene& S, const RenderParams& P, float* acc, float* partial, uint32_t* work,
                                  uint32_t* spill, DevCounters* cnt, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL((sol_render_kernel<COUNT, MEDIUM>), dim3(grid), dim3(SOL_WG), 0, stream, S, P, acc, partial, work, spill, cnt);
  return hipGetLastError();
}

hipError_t sol_launch_render(const DevScene& S, const RenderParams& P, float* acc, float* partial, uint32_t* work,
                             uint32_t* spill, DevCounters* cnt, uint32_t grid, bool count, bool medium, hipStream_t stream) {
  if (count) return medium ? launch_render_t<true, true>(S, P, acc, partial, work, spill, cnt, grid, stream)
                           : launch_render_t<true, false>(S, P, acc, partial, work, spill, cnt, grid, stream);
  return medium ? launch_render_t<false, true>(S, P, acc, partial, work, spill, cnt, grid, stream)
                : launch_render_t<false, false>(S, P, acc, partial, work, spill, cnt, grid, stream);
}

int sol_render_blocks_per_cu(bool count, bool medium) {
  int n = 0;
  hipError_t e;
  if (count) e = medium ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sol_render_kernel<true, true>, SOL_WG, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sol_render_kernel<true, false>, SOL_WG, 0);
  else e = medium ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sol_render_kernel<false, true>, SOL_WG, 0)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sol_render_kernel<false, false>, SOL_WG, 0);
  if (e != hipSuccess || n < 1) n = 1;
  return n;
}

hipError_t sol_launch_resolve(float* acc, const float* partial, uint32_t n_floats, uint32_t n_chunks, hipStream_t stream) {
  uint32_t grid = (n_floats + 255u) / 256u;
  if (grid > 4096u) grid = 4096u;
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL(sol_resolve_kernel, dim3(grid), dim3(256), 0, stream, acc, partial, n_floats, n_chunks);
  return hipGetLastError();
}

hipError_t sol_launch_unpermute(const float* gathered, float* image, uint32_t width, uint32_t height, uint32_t blocks_x,
                                uint32_t world, uint32_t only_rank, size_t stride, hipStream_t stream) {
  uint32_t grid = (width * height + 255u) / 256u;
  if (grid > 4096u) grid = 4096u;
  hipLaunchKernelGGL(sol_unpermute_kernel, dim3(grid), dim3(256), 0, stream, gathered, image, width, height, blocks_x, world,
                     only_rank, stride);
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
      bool h = sphere_test(S, mk3(x[4], x[5], x[6]), mk3(x[7], x[8], x[9]), x[10], x[11], t);
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
      T.e2z = x[8]; T.dfs = 0; T.mat = 0; T.pad = 0;
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
