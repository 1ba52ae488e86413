// sol_trace.h -- primitive intersection and BVH closest-hit traversal on the device.
//
// Results are those of the reference's `Bvh::hit` (src/hittable/bvh.rs:165-180): the primitive with the smallest
// ray parameter t in the interval, and among exactly equal t the one LAST in depth-first leaf order (the
// reference's right-hand search uses the inclusive interval [min, t_left], src/util/interval.rs:67-69). The cost
// is not the reference's: children are visited near-first, boxes beyond the current best t are culled, and the
// per-lane stack lives in LDS ([level][lane] layout: lane l always hits bank l%32, conflict-free).
#pragma once
#include "../../include/solstrale_hip.h"
#include "sol_math.h"
#include "sol_types.h"

#define REF_DONE 0xFFFFFFFFu
#define ALMOST_ZERO_F 1e-8f  // src/geo/vec3.rs:21
#define RAY_MIN_F 0.001f     // RAY_INTERVAL.min (src/util/interval.rs:25-28)

struct Hit {
  float t;
  uint32_t ref;  // SOL_REF_NONE kind when nothing was hit
  uint32_t dfs;
  float u, v;    // barycentrics (triangle) / planar coordinates (quad), the reference's f32 values
};

struct Counters {
  uint32_t samples, rays, node_visits, sphere_tests, quad_tests, triangle_tests, shades, texel_fetches, max_stack;
};

struct Stack {
  uint32_t* lds;     // base of this workgroup's [SOL_LDS_STACK][SOL_WG] array, already offset by the lane
  uint32_t* spill;   // base of the global spill area, already offset by the global thread id
  uint32_t stride;   // total threads (spill stride between levels)
};
DEV void stack_push(const Stack& s, int& sp, uint32_t v) {
  if (sp < SOL_LDS_STACK) s.lds[sp * SOL_WG] = v;
  else s.spill[(size_t)(sp - SOL_LDS_STACK) * s.stride] = v;
  sp++;
}
DEV uint32_t stack_pop(const Stack& s, int& sp) {
  sp--;
  if (sp < SOL_LDS_STACK) return s.lds[sp * SOL_WG];
  return s.spill[(size_t)(sp - SOL_LDS_STACK) * s.stride];
}

// Aabb::hit (src/geo/mod.rs:159-188): slab test over [0, inf); fmaxf/fminf return the non-NaN operand like Rust's
// f64::max/min. Returns the entry parameter in t_entry; hit iff t_min < t_max.
DEV bool slab(float xmin, float xmax, float ymin, float ymax, float zmin, float zmax, f3 o, f3 inv, bool sx, bool sy,
              bool sz, float& t_entry) {
  float t_min = 0.0f, t_max = __builtin_huge_valf();
  t_min = fmaxf(((sx ? xmax : xmin) - o.x) * inv.x, t_min);
  t_max = fminf(((sx ? xmin : xmax) - o.x) * inv.x, t_max);
  t_min = fmaxf(((sy ? ymax : ymin) - o.y) * inv.y, t_min);
  t_max = fminf(((sy ? ymin : ymax) - o.y) * inv.y, t_max);
  t_min = fmaxf(((sz ? zmax : zmin) - o.z) * inv.z, t_min);
  t_max = fminf(((sz ? zmin : zmax) - o.z) * inv.z, t_max);
  t_entry = t_min;
  return t_min < t_max;
}

// Triangle::hit (src/hittable/triangle.rs:119-173), geometric part.
DEV bool tri_test(const DTri& T, f3 o, f3 d, float tmin, float tmax, float& t, float& u, float& v) {
  f3 v0 = mk3(T.v0x, T.v0y, T.v0z), e1 = mk3(T.e1x, T.e1y, T.e1z), e2 = mk3(T.e2x, T.e2y, T.e2z);
  f3 p_vec = cross3(d, e2);
  float det = dot3(e1, p_vec);
  float inv_det = 1.0f / det;
  f3 t_vec = o - v0;
  f3 q_vec = cross3(t_vec, e1);
  u = dot3(t_vec, p_vec) * inv_det;
  v = dot3(d, q_vec) * inv_det;
  t = dot3(e2, q_vec) * inv_det;
  bool ok = !(fabsf(det) < ALMOST_ZERO_F);
  ok = ok && (u >= 0.0f && u <= 1.0f);
  ok = ok && !(v < 0.0f || u + v > 1.0f);
  ok = ok && (tmin <= t && t <= tmax);
  return ok;
}
// Quad::hit (src/hittable/quad.rs:150-194), geometric part.
DEV bool quad_test(const DQuad& Q, f3 o, f3 d, float tmin, float tmax, float& t, float& u, float& v) {
  f3 n = mk3(Q.nx, Q.ny, Q.nz);
  float denom = dot3(n, d);
  if (fabsf(denom) < ALMOST_ZERO_F) return false;
  t = (Q.d - dot3(n, o)) / denom;
  if (!(tmin <= t && t <= tmax)) return false;
  f3 hp = o + d * t;
  f3 planar = hp - mk3(Q.qx, Q.qy, Q.qz);
  f3 w = mk3(Q.wx, Q.wy, Q.wz);
  u = dot3(w, cross3(planar, mk3(Q.vx, Q.vy, Q.vz)));
  v = dot3(w, cross3(mk3(Q.ux, Q.uy, Q.uz), planar));
  return (u >= 0.0f && u <= 1.0f) && (v >= 0.0f && v <= 1.0f);
}
// Sphere::hit (src/hittable/sphere.rs:64-108), geometric part.
DEV bool sphere_test(const DSphere& S, f3 o, f3 d, float tmin, float tmax, float& t) {
  f3 oc = o - mk3(S.cx, S.cy, S.cz);
  float a = len2(d);
  float half_b = dot3(oc, d);
  float c = len2(oc) - S.radius * S.radius;
  float disc = half_b * half_b - a * c;
  if (disc < 0.0f) return false;
  float sqrt_d = sol_sqrt(disc);
  float root = (-half_b - sqrt_d) / a;
  if (!(tmin <= root && root <= tmax)) {
    root = (-half_b + sqrt_d) / a;
    if (!(tmin <= root && root <= tmax)) return false;
  }
  t = root;
  return true;
}

DEV bool better(float t, uint32_t dfs, const Hit& h) {
  return t < h.t || (t == h.t && (SOL_REF_KIND(h.ref) == SOL_REF_NONE || dfs > h.dfs));
}

template <bool COUNT, bool MEDIUM>
DEV void closest_hit(const DevScene& S, f3 o, f3 d, f3 inv, float tmin, float tmax, uint32_t root, float bxmin,
                     float bxmax, float bymin, float bymax, float bzmin, float bzmax, Hit& h, const Stack& st, int sp_base,
                     const Rng& rng, uint32_t depth, Counters& cnt);

// ConstantMedium::hit (src/hittable/constant_medium.rs:35-79). Its draws come from the sub-stream
// 0x40000000 + (depth<<20 | medium<<8) + i of the path's generator (DESIGN.md "RNG"), so the outcome does not
// depend on when the tree search reaches the medium.
template <bool COUNT>
DEV bool medium_test(const DevScene& S, uint32_t midx, f3 o, f3 d, f3 inv, float tmin, float tmax, float& t_out,
                     const Stack& st, int sp, const Rng& rng, uint32_t depth, Counters& cnt) {
  const DMedium M = S.mediums[midx];
  const float inf = __builtin_huge_valf();
  Hit h1, h2;
  h1.t = inf; h1.ref = 0; h1.dfs = 0; h1.u = h1.v = 0.f;
  closest_hit<COUNT, false>(S, o, d, inv, -inf, inf, M.boundary, M.bxmin, M.bxmax, M.bymin, M.bymax, M.bzmin, M.bzmax, h1,
                            st, sp, rng, depth, cnt);
  if (SOL_REF_KIND(h1.ref) == SOL_REF_NONE) return false;
  h2.t = inf; h2.ref = 0; h2.dfs = 0; h2.u = h2.v = 0.f;
  closest_hit<COUNT, false>(S, o, d, inv, h1.t + 0.0001f, inf, M.boundary, M.bxmin, M.bxmax, M.bymin, M.bymax, M.bzmin,
                            M.bzmax, h2, st, sp, rng, depth, cnt);
  if (SOL_REF_KIND(h2.ref) == SOL_REF_NONE) return false;
  float t1 = fmaxf(h1.t, tmin);
  float t2 = fminf(h2.t, tmax);
  if (t1 >= t2) return false;
  t1 = fmaxf(t1, 0.0f);
  float r_length = len3(d);
  float distance_inside = (t2 - t1) * r_length;
  uint32_t c = 0x40000000u + (((depth & 0x3FFu) << 20) | ((midx & 0xFFFu) << 8));
  float hit_distance = M.nid * log_r(u32_to_unit(rng_bits(rng, c)));
  if (hit_distance > distance_inside) return false;
  t_out = t1 + hit_distance / r_length;
  return true;
}
// The random unit normal of a medium hit (same sub-stream, draws 1..).
DEV f3 medium_normal(const Rng& rng, uint32_t midx, uint32_t depth) {
  uint32_t c = 0x40000000u + (((depth & 0x3FFu) << 20) | ((midx & 0xFFFu) << 8)) + 1u;
  f3 p = mk3(0.f, 0.f, 0.f);
  for (int it = 0; it < 80; ++it) {  // random_in_unit_sphere (vec3.rs:380-392); the bound is never reached
    p.x = u32_to_unit(rng_bits(rng, c++)) * 2.0f + -1.0f;
    p.y = u32_to_unit(rng_bits(rng, c++)) * 2.0f + -1.0f;
    p.z = u32_to_unit(rng_bits(rng, c++)) * 2.0f + -1.0f;
    if (len2(p) < 1.0f) break;
  }
  return unit3(p);
}

template <bool COUNT, bool MEDIUM>
DEV void closest_hit(const DevScene& S, f3 o, f3 d, f3 inv, float tmin, float tmax, uint32_t root, float bxmin,
                     float bxmax, float bymin, float bymax, float bzmin, float bzmax, Hit& h, const Stack& st, int sp_base,
                     const Rng& rng, uint32_t depth, Counters& cnt) {
  const bool sx = __builtin_signbitf(inv.x), sy = __builtin_signbitf(inv.y), sz = __builtin_signbitf(inv.z);
  h.t = tmax;
  h.ref = SOL_MAKE_REF(SOL_REF_NONE, 0);
  h.dfs = 0;
  h.u = h.v = 0.0f;
  uint32_t cur = root;
  if (SOL_REF_KIND(root) == SOL_REF_NODE) {  // Bvh::hit starts with its own box (bvh.rs:166)
    float te;
    if (!slab(bxmin, bxmax, bymin, bymax, bzmin, bzmax, o, inv, sx, sy, sz, te)) return;
  }
  int sp = sp_base;
  for (;;) {
    const uint32_t kind = SOL_REF_KIND(cur);
    const uint32_t idx = SOL_REF_INDEX(cur);
    if (kind == SOL_REF_NODE) {
      const float4* np = reinterpret_cast<const float4*>(S.nodes + idx);
      const float4 a = np[0], b = np[1], c = np[2];
      const uint4 r = *reinterpret_cast<const uint4*>(np + 3);
      if (COUNT) cnt.node_visits++;
      // Culling: a box whose entry parameter lies beyond the best hit cannot hold a better one. The slab test clamps the
      // entry to 0 (origin inside the box), and a search over (-inf, inf) (constant-medium boundary) accepts hits at
      // negative t, so boxes entered at 0 are never culled.
      const float cull_t = fmaxf(h.t, 0.0f);
      float tl, tr;
      bool hl = slab(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, sx, sy, sz, tl) && tl <= cull_t;
      bool hr = slab(b.z, b.w, c.x, c.y, c.z, c.w, o, inv, sx, sy, sz, tr) && tr <= cull_t;
      if (tmin < 0.0f) {
        // r.z bit0/bit1: that child's box is the primitive's own box, which the reference does not test (a two-leaf
        // `Bvh` tests only its union box, bvh.rs:91-96). For t >= 0 the extra test is a sound cull; a primitive hit at
        // negative t lies outside the forward slab, so here the child is taken whenever the node was reached.
        if (r.z & 1u) { hl = true; tl = 0.0f; }
        if (r.z & 2u) { hr = true; tr = 0.0f; }
      }
      if (hl && hr) {
        const bool lfirst = tl <= tr;
        stack_push(st, sp, lfirst ? r.y : r.x);
        if (COUNT) cnt.max_stack = max(cnt.max_stack, (uint32_t)sp);
        cur = lfirst ? r.x : r.y;
        continue;
      }
      if (hl) { cur = r.x; continue; }
      if (hr) { cur = r.y; continue; }
    } else if (kind == SOL_REF_TRIANGLE) {
      const float4* tp = reinterpret_cast<const float4*>(S.tris + idx);
      const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
      DTri T;
      T.v0x = p0.x; T.v0y = p0.y; T.v0z = p0.z; T.e1x = p0.w; T.e1y = p1.x; T.e1z = p1.y; T.e2x = p1.z; T.e2y = p1.w; T.e2z = p2.x;
      const uint32_t dfs = __float_as_uint(p2.y);
      if (COUNT) cnt.triangle_tests++;
      float t, u, v;
      if (tri_test(T, o, d, tmin, h.t, t, u, v) && better(t, dfs, h)) { h.t = t; h.ref = cur; h.dfs = dfs; h.u = u; h.v = v; }
    } else if (kind == SOL_REF_SPHERE) {
      const DSphere Sp = S.spheres[idx];
      if (COUNT) cnt.sphere_tests++;
      float t;
      if (sphere_test(Sp, o, d, tmin, h.t, t) && better(t, Sp.dfs, h)) { h.t = t; h.ref = cur; h.dfs = Sp.dfs; }
    } else if (kind == SOL_REF_QUAD) {
      const DQuad Q = S.quads[idx];
      if (COUNT) cnt.quad_tests++;
      float t, u, v;
      if (quad_test(Q, o, d, tmin, h.t, t, u, v) && better(t, Q.dfs, h)) { h.t = t; h.ref = cur; h.dfs = Q.dfs; h.u = u; h.v = v; }
    } else if (MEDIUM && kind == SOL_REF_MEDIUM) {
      float t;
      const uint32_t dfs = S.mediums[idx].dfs;
      if (medium_test<COUNT>(S, idx, o, d, inv, tmin, h.t, t, st, sp, rng, depth, cnt) && better(t, dfs, h)) {
        h.t = t; h.ref = cur; h.dfs = dfs;
      }
    }
    if (sp == sp_base) break;
    cur = stack_pop(st, sp);
  }
}
