// sol_path.h -- one path vertex at a time: camera-ray generation and the flattened ray_color <-> shade recursion.
// Shared by every render kernel so that all of them execute the identical fp32 operation sequence.
#pragma once
#include "sol_shade.h"

// The recursion ray_color <-> shade (src/renderer/shader.rs:62-125) flattened exactly: with per-level factors a_d >= 0, a
// composition of  x -> a*x  (ScatterBasic) and  x -> min(a*x, 3) with NaN -> 0  (ScatterPdf) is always x -> min(A*x, C);
// a Pdf level does C <- min(C, 3*A), A <- A*a; a Basic level does A <- A*a (DESIGN.md "Flattened recursion").
struct Path {
  f3 o, d;        // current ray (Ray::new's cached reciprocal is recomputed where the ray is traced)
  f3 A, C;
  float acc_len;  // accumulated_ray_length (sum of parametric t, shader.rs:74)
  uint32_t depth;
  bool pdf_seen;
  uint32_t from;  // fp32 rule 8: the quad / triangle the current ray leaves, as 2^31 | its depth-first index (0: a camera ray, a sphere,
                  // a medium) - sol_self_hit. (The dfs index, not the reference: a pre-split triangle has one record per part, all with its dfs index.)
  Rng rng;
};

// fp32 rule 8: a ray does not hit the FLAT primitive it leaves. A line meets a plane once and the ray starts on it - in f64 the second "hit" lies at
// t ~ 1e-11, far below RAY_MIN, and never counts; in fp32 the start point is a few 1e-5 off its plane at coordinates of hundreds and a grazing ray finds
// the plane again just above RAY_MIN (C1: 7 samples in a million went into the box they had just left and came back black; found by the 1024-spp
// run of the f64 gate). True when the closest hit `h` of a finished world search is that primitive: the caller searches again BEHIND it
// (tmin = sol_behind(h.t); the oracle's float instantiation does the same: oracle.cpp ray_color).
#ifndef SOL_RULE8
#define SOL_RULE8 1  // (0: an A/B build without the rule - tests/tools/variants.py -, for pricing it; its frames are not the oracle's)
#endif
DEV bool sol_self_hit(uint32_t from, const Hit& h) { return SOL_RULE8 && from != 0u && (0x80000000u | h.dfs) == from && SOL_REF_KIND(h.ref) != SOL_REF_NONE; }
DEV float sol_behind(float t) { return __uint_as_float(__float_as_uint(t) + 1u); }  // (the next float above a positive finite t: std::nextafter's)
// Path::from of a ray that starts on the primitive of hit `h`. The product kernel's hits carry their dfs index; the A/B kernels park a hit without it
// (Hit::dfs = SOL_DFS_UNKNOWN on adoption): then it is read from the primitive's record.
#define SOL_DFS_UNKNOWN 0xFFFFFFFFu
DEV uint32_t sol_flat_from(const DevScene& S, const Hit& h) {
  const uint32_t kind = SOL_REF_KIND(h.ref), idx = SOL_REF_INDEX(h.ref);
  if (kind != SOL_REF_TRIANGLE && kind != SOL_REF_QUAD) return 0u;
  uint32_t dfs = h.dfs;
  if (dfs == SOL_DFS_UNKNOWN) dfs = kind == SOL_REF_TRIANGLE ? ldg_u32(&S.tris[idx].dfs) : ldg_u32(&S.quads[idx].dfs);
  return 0x80000000u | dfs;
}

// Pixel jitter + Camera::get_ray (src/renderer/mod.rs:263-265, src/camera.rs:77-89) for sample `s` of pixel (px, py),
// py counted from the image top.
template <bool COUNT>
DEV void generate_path(const DevScene& S, uint32_t seed_lo, uint32_t seed_hi, uint32_t px, uint32_t py, uint32_t s, Path& p,
                       Counters& cnt) {
  rng_init(p.rng, seed_lo, seed_hi, py * S.width + px, s);
  if (COUNT) cnt.samples++;
  const uint32_t y_ref = (S.height - 1u) - py;
  float u = ((float)px + rnd(p.rng)) / (float)(S.width - 1u);
  float v = ((float)y_ref + rnd(p.rng)) / (float)(S.height - 1u);
  f3 offset = mk3(0.f, 0.f, 0.f);
  if (S.cam.lens_radius > 0.0f) {
    f3 rd = mk3(0.f, 0.f, 0.f);
    for (int it = 0; it < 80; ++it) {  // random_in_unit_disc (vec3.rs:400-412); the bound is never reached
      rd.x = rnd_range(p.rng, -1.0f, 1.0f);
      rd.y = rnd_range(p.rng, -1.0f, 1.0f);
      if (len2(rd) < 1.0f) break;
    }
    rd = rd * S.cam.lens_radius;
    offset = mk3(S.cam.ux, S.cam.uy, S.cam.uz) * rd.x + mk3(S.cam.wx, S.cam.wy, S.cam.wz) * rd.y;
  }
  const f3 org = mk3(S.cam.ox, S.cam.oy, S.cam.oz);
  p.d = mk3(S.cam.llx, S.cam.lly, S.cam.llz) + mk3(S.cam.hx, S.cam.hy, S.cam.hz) * u + mk3(S.cam.vx, S.cam.vy, S.cam.vz) * v - org -
        offset;
  p.o = org + offset;
  const float inf = __builtin_huge_valf();
  p.A = mk3(1.f, 1.f, 1.f);
  p.C = mk3(inf, inf, inf);
  p.acc_len = 0.0f;
  p.depth = 0;
  p.pdf_seen = false;
  p.from = 0u;
}

DEV f3 ray_inverse(f3 d) { return mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }  // Ray::new (geo/mod.rs:277-285)

// EXTENSION, not in the reference (include/solstrale_hip.h, SolSceneDesc::env_*): radiance of a ray that hits nothing, from a
// latitude-longitude map; the arithmetic of oracle.cpp env_color in fp32.
DEV f3 env_color(const DevScene& S, f3 dir) {
  const f3 n = unit3(dir);
  const float theta = acos_r(-n.y);
  const float phi = -atan2_r(n.z, n.x) + SOL_PI;
  const float u = phi / (2.0f * SOL_PI), v = theta / SOL_PI;
  const float x = u * ((float)S.env_w - 1.0f), y = (1.0f - v) * ((float)S.env_h - 1.0f);
  uint32_t xi = x >= 0.0f ? (x < 4294967296.0f ? (uint32_t)x : 0xFFFFFFFFu) : 0u;
  uint32_t yi = y >= 0.0f ? (y < 4294967296.0f ? (uint32_t)y : 0xFFFFFFFFu) : 0u;
  xi = min(xi, S.env_w - 1u);
  yi = min(yi, S.env_h - 1u);
  const float* p = S.env + ((size_t)yi * S.env_w + xi) * 3;
  return mk3(__uint_as_float(ldg_u32(p)), __uint_as_float(ldg_u32(p + 1)), __uint_as_float(ldg_u32(p + 2))) * S.env_scale;
}

// Processes the result `h` of world.hit(ray) for the path `p`. Returns true when the sample is finished, with its colour
// (get_attenuated_color applied) in `contrib`; returns false when the path continues with the new ray in p.o / p.d.
template <bool COUNT, bool STRICT = false>
DEV bool shade_vertex(const DevScene& S, Path& p, const Hit& h, f3& contrib, Counters& cnt) {
  bool terminal = true;
  f3 x = mk3(S.bgx, S.bgy, S.bgz);  // miss: background (src/renderer/mod.rs:197-204)
  if (S.env != nullptr && SOL_REF_KIND(h.ref) == SOL_REF_NONE) x = env_color(S, p.d);  // (extension: environment map)
  bool has_af = false;
  float af = 0.0f, path_len = 0.0f;
  if (SOL_REF_KIND(h.ref) != SOL_REF_NONE) {
    phase_tick<COUNT>(cnt, 1);
    Surface sf;
    build_surface<COUNT>(S, p.o, p.d, h, p.rng, p.depth, sf);
    sf.normal = transformed_normal<COUNT>(S, sf.mat, sf, p.rng, cnt);  // RayHit::new (material/mod.rs:50)
    Scatter sc;
    if (S.shader != SOL_SHADER_PATH_TRACING) {
      if (S.shader == SOL_SHADER_NORMAL) {  // shader.rs:165-172
        x = sf.normal;
      } else {  // Albedo (shader.rs:141-151), Simple (shader.rs:192-214)
        scatter<COUNT, STRICT>(S, p.d, sf, p.rng, sc, cnt);
        x = sc.color;
        if (S.shader == SOL_SHADER_SIMPLE && sc.type != SCATTER_EMISSION)
          x = sc.color * (dot3(sf.normal, mk3(1.f, 1.f, -1.f)) * 0.5f + 0.75f);
      }
    } else if (p.depth >= S.max_depth) {  // shader.rs:70-72
      x = mk3(0.f, 0.f, 0.f);
    } else {
      const float total = sf.t + p.acc_len;  // shader.rs:74
      scatter<COUNT, STRICT>(S, p.d, sf, p.rng, sc, cnt);
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
            p.C = mk3(fminf(p.C.x, p.A.x * 3.0f), fminf(p.C.y, p.A.y * 3.0f), fminf(p.C.z, p.A.z * 3.0f));
            p.A = p.A * a;
            p.pdf_seen = true;
          }
        } else {  // ScatterBasic (shader.rs:85-94)
          p.A = p.A * sc.color;
        }
        if (go_on) {
          terminal = false;
          p.o = sf.p;
          p.d = sc.dir;
          p.acc_len = total;
          p.depth++;
          p.from = SOL_RULE8 ? sol_flat_from(S, h) : 0u;  // (rule 8: the scattered ray starts on this primitive)
        }
      }
    }
  }
  if (terminal) {
    f3 c = x;
    if (S.shader == SOL_SHADER_PATH_TRACING) {
      c = p.A * x;
      if (p.pdf_seen) {
        c.x = isnan(c.x) ? 0.0f : fminf(c.x, p.C.x);
        c.y = isnan(c.y) ? 0.0f : fminf(c.y, p.C.y);
        c.z = isnan(c.z) ? 0.0f : fminf(c.z, p.C.z);
      }
      if (has_af) c = (c * 1.0f) / (1.0f + af * path_len);  // get_attenuated_color (material/mod.rs:127-131)
    }
    contrib = c;
  }
  return terminal;
}

// Work item -> pixel. Items are ordered so that an aligned run of 64 is one 8x8 pixel block of this rank.
struct Item {
  uint32_t px, py, slot, chunk;
};
DEV bool decode_item(const DevScene& S, const RenderParams& P, uint32_t item, Item& it) {
  const uint32_t slots = P.n_local_blocks * 64u;
  it.chunk = item / slots;
  it.slot = item - it.chunk * slots;
  const uint32_t lb = it.slot >> 6, pin = it.slot & 63u;
  const uint32_t b = S.block_of_local ? ldg_u32(S.block_of_local + lb) : lb * P.world + P.rank;
  const uint32_t by = b / P.blocks_x, bx = b - by * P.blocks_x;
  it.px = bx * SOL_TILE + (pin & 7u);
  it.py = by * SOL_TILE + (pin >> 3);
  return it.px < S.width && it.py < S.height;  // false: padding pixel of an edge block
}
// The product kernel's order of the same items. A 16-sample item of a pixel inside glass can be 800 rays long, ~40 ms of one
// lane's time, whatever else the GPU does: taken last it IS the tail of the launch (C5: 40 of 220 ms). So the blocks a counted
// probe found heavy come first, block-major (all chunks of the heaviest block, then the next, ..), and the rest follows in
// the chunk-major order of decode_item. Images do not depend on the order.
DEV bool decode_item_ordered(const DevScene& S, const RenderParams& P, uint32_t item, Item& it) {
  const uint32_t pair = item >> 6, pin = item & 63u;
  const uint32_t heavy_pairs = S.n_first * P.n_chunks;
  uint32_t k;
  if (pair < heavy_pairs) {
    k = pair / P.n_chunks;
    it.chunk = pair - k * P.n_chunks;
  } else {
    const uint32_t q = pair - heavy_pairs, rest = P.n_traced_blocks - S.n_first;
    it.chunk = q / rest;
    k = S.n_first + (q - it.chunk * rest);
  }
  const uint32_t lb = S.block_order ? ldg_u32(S.block_order + k) : k;
  it.slot = lb * 64u + pin;
  const uint32_t b = S.block_of_local ? ldg_u32(S.block_of_local + lb) : lb * P.world + P.rank;
  const uint32_t by = b / P.blocks_x, bx = b - by * P.blocks_x;
  it.px = bx * SOL_TILE + (pin & 7u);
  it.py = by * SOL_TILE + (pin >> 3);
  return it.px < S.width && it.py < S.height;  // false: padding pixel of an edge block
}

// Writes a finished chunk sum: straight into the accumulator when the call has one chunk, else into the partial plane.
DEV void write_chunk(const RenderParams& P, float* __restrict__ acc, float* __restrict__ partial, uint32_t slot, uint32_t chunk,
                     f3 sum) {
  if (P.n_chunks == 1) {
    float* a = acc + (size_t)slot * 3;
    a[0] += sum.x; a[1] += sum.y; a[2] += sum.z;
  } else {
    float* a = partial + ((size_t)chunk * (P.n_local_blocks * 64u) + slot) * 3;
    a[0] = sum.x; a[1] = sum.y; a[2] = sum.z;
  }
}

DEV void flush_counters(const Counters& cnt, DevCounters* __restrict__ dcnt) {
  atomicAdd(&dcnt->samples, (unsigned long long)cnt.samples);
  atomicAdd(&dcnt->rays, (unsigned long long)cnt.rays);
  atomicAdd(&dcnt->node_visits, (unsigned long long)cnt.node_visits);
  atomicAdd(&dcnt->sphere_tests, (unsigned long long)cnt.sphere_tests);
  atomicAdd(&dcnt->quad_tests, (unsigned long long)cnt.quad_tests);
  atomicAdd(&dcnt->triangle_tests, (unsigned long long)cnt.triangle_tests);
  atomicAdd(&dcnt->shades, (unsigned long long)cnt.shades);
  atomicAdd(&dcnt->texel_fetches, (unsigned long long)cnt.texel_fetches);
  atomicMax(&dcnt->max_stack, (unsigned long long)cnt.max_stack);
  for (int k = 0; k < 6; ++k) atomicAdd(&dcnt->phase[k], (unsigned long long)cnt.phase[k]);
  atomicAdd(&dcnt->primary_hits, (unsigned long long)cnt.primary_hits);
  for (int k = 0; k < 6; ++k) atomicAdd(&dcnt->path_len[k], (unsigned long long)cnt.path_len[k]);
}
// (counted builds) a sample ended after `rays` rays: its bin in the path-length histogram
DEV void count_path(Counters& cnt, uint32_t rays) {
  cnt.path_len[rays <= 1u ? 0 : rays == 2u ? 1 : rays <= 4u ? 2 : rays <= 8u ? 3 : rays <= 16u ? 4 : 5]++;
}
