// sol_trace.h -- primitive intersection and BVH closest-hit traversal on the device.
//
// Results are those of the reference's `Bvh::hit` (src/hittable/bvh.rs:165-180): the primitive with the smallest
// ray parameter t in the interval, and among exactly equal t the one LAST in depth-first leaf order (the
// reference's right-hand search uses the inclusive interval [min, t_left], src/util/interval.rs:67-69). The cost
// is not the reference's: children are visited near-first, boxes beyond the current best t are culled, and the
// per-lane stack lives in LDS ([level][lane] layout: lane l always hits bank l%32, conflict-free).
//
// The search is written as a resumable stepper (Trav / trav_begin / trav_step) so that a render kernel can refill idle
// lanes with new rays between steps; closest_hit() is the run-to-completion form.
#pragma once
#include "../../include/solstrale_hip.h"
#include "sol_math.h"
#include "sol_types.h"

// The world (searches with t >= 0.001) is walked through the 8-wide quantised tree; -DSOL_WORLD_BINARY=true builds the
// A/B variant that walks the 2-wide DNode tree instead (same results).
#define SOL_WORLD_ROOT(S) (SOL_WORLD_BINARY ? (S).root : (S).wroot)

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
  uint32_t phase[6];  // lane-utilisation instrumentation: (active lanes, 64 per executing wave) for traverse / shade / generate
};

// Lane-utilisation instrumentation (COUNT builds only): every active lane counts itself, the first active lane of the
// wave counts the 64 slots of this execution.
template <bool COUNT>
DEV void phase_tick(Counters& cnt, int ph) {
  if (COUNT) {
    cnt.phase[2 * ph]++;
    const unsigned long long m = __ballot(1);
    if ((int)__lane_id() == __ffsll((long long)m) - 1) cnt.phase[2 * ph + 1] += 64u;
  }
}

struct Stack {
  uint32_t* lds;     // base of this workgroup's [depth][SOL_WG] array, already offset by the lane
  uint32_t* spill;   // base of the global spill area, already offset by the global thread id
  uint32_t stride;   // total threads (spill stride between levels)
  int depth;         // entries per lane kept in LDS; deeper entries go to the spill area
};
DEV void stack_push(const Stack& s, int& sp, uint32_t v) {
  if (sp < s.depth) s.lds[sp * SOL_WG] = v;
  else s.spill[(size_t)(sp - s.depth) * s.stride] = v;
  sp++;
}
DEV void stack_store(const Stack& s, int level, uint32_t v) {
  if (level < s.depth) s.lds[level * SOL_WG] = v;
  else s.spill[(size_t)(level - s.depth) * s.stride] = v;
}
DEV uint32_t stack_pop(const Stack& s, int& sp) {
  sp--;
  if (sp < s.depth) return s.lds[sp * SOL_WG];
  return s.spill[(size_t)(sp - s.depth) * s.stride];
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
// fp32 contract, spheres: a root counts only if its hit point lies inside the sphere's own box (centre +- radius, widened
// by `slack` = half the box pad). In fp32 the quadratic suffers catastrophic cancellation for distant origins with
// unnormalised directions (camera at 800 units, |d| = 800: half_b^2 and a*c agree to 7 digits), and reports "hits" whose
// point is up to 0.02 units off the sphere - outside every box that bounds it. Whether such a phantom is seen would then
// depend on which boxes a traversal happens to test. With this rule a hit is always inside all of its boxes, so every
// conservative BVH layout (reference-shaped, two-leaf extra boxes, 8-wide quantised) returns the same hit. In f64 the rule
// never fires.
DEV bool sphere_root_ok(const DSphere& S, f3 o, f3 d, float root, float tmin, float tmax, float slack) {
  if (!(tmin <= root && root <= tmax)) return false;
  const f3 hp = o + d * root;
  const float lim = S.radius + slack;
  return fabsf(hp.x - S.cx) <= lim && fabsf(hp.y - S.cy) <= lim && fabsf(hp.z - S.cz) <= lim;
}
// Sphere::hit (src/hittable/sphere.rs:64-108), geometric part.
DEV bool sphere_test(const DSphere& S, f3 o, f3 d, float tmin, float tmax, float slack, float& t) {
  f3 oc = o - mk3(S.cx, S.cy, S.cz);
  float a = len2(d);
  float half_b = dot3(oc, d);
  float c = len2(oc) - S.radius * S.radius;
  float disc = half_b * half_b - a * c;
  if (disc < 0.0f) return false;
  float sqrt_d = sol_sqrt(disc);
  float root = (-half_b - sqrt_d) / a;
  if (!sphere_root_ok(S, o, d, root, tmin, tmax, slack)) {
    root = (-half_b + sqrt_d) / a;
    if (!sphere_root_ok(S, o, d, root, tmin, tmax, slack)) return false;
  }
  t = root;
  return true;
}

DEV bool better(float t, uint32_t dfs, const Hit& h) {
  return t < h.t || (t == h.t && (SOL_REF_KIND(h.ref) == SOL_REF_NONE || dfs > h.dfs));
}

// Visit order of the 8-wide tree: kWideFar[octant] = {slots 0-3, slots 4-7}, byte i = mask of the slots j that come after
// slot i for a ray of that direction octant, i.e. (j ^ octant) > (i ^ octant).
static __device__ const uint2 kWideFar[8] = {
    {0xF0F8FCFEu, 0x0080C0E0u}, {0xF4F0FDFCu, 0x4000D0C0u}, {0xF3FBF0F2u, 0x30B00020u}, {0xF7F3F1F0u, 0x70301000u},
    {0x00080C0Eu, 0x0F8FCFEFu}, {0x04000D0Cu, 0x4F0FDFCFu}, {0x030B0002u, 0x3FBF0F2Fu}, {0x07030100u, 0x7F3F1F0Fu}};

// State of one closest-hit search.
struct Trav {
  f3 o, d, inv;
  float tmin;
  uint32_t cur;  // reference being visited, REF_DONE when the search is over
  int sp, sp_base;
  uint2 far;     // kWideFar[ray octant] (8-wide searches only)
  Hit h;
};

// Starts a search of `root` over [tmin, tmax]; (bxmin..bzmax) is root's own box, tested first when root is a node
// (Bvh::hit, bvh.rs:166).
DEV void trav_begin(Trav& t, f3 o, f3 d, float tmin, float tmax, uint32_t root, float bxmin, float bxmax, float bymin,
                    float bymax, float bzmin, float bzmax, int sp_base) {
  t.o = o; t.d = d; t.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // Ray::new (geo/mod.rs:277-285)
  t.tmin = tmin;
  t.h.t = tmax;
  t.h.ref = SOL_MAKE_REF(SOL_REF_NONE, 0);
  t.h.dfs = 0;
  t.h.u = t.h.v = 0.0f;
  t.sp = t.sp_base = sp_base;
  t.cur = root;
  t.far = make_uint2(0u, 0u);
  // A ray with a NaN in its origin or direction cannot hit anything: every primitive test ends in a comparison with NaN,
  // which is false (the reference returns None the same way, after visiting every box - Aabb::hit ignores NaN). Such rays
  // occur a few times per 10^8 samples; without this exit one lane walks the whole tree and tests every primitive.
  if (isnan(d.x) || isnan(d.y) || isnan(d.z) || isnan(o.x) || isnan(o.y) || isnan(o.z)) { t.cur = REF_DONE; return; }
  if (SOL_REF_KIND(root) == SOL_REF_NODE || SOL_REF_KIND(root) == SOL_REF_WIDE) {
    float te;
    if (!slab(bxmin, bxmax, bymin, bymax, bzmin, bzmax, o, t.inv, __builtin_signbitf(t.inv.x), __builtin_signbitf(t.inv.y),
              __builtin_signbitf(t.inv.z), te))
      t.cur = REF_DONE;
    if (SOL_REF_KIND(root) == SOL_REF_WIDE)
      t.far = kWideFar[(__builtin_signbitf(t.inv.x) ? 4u : 0u) | (__builtin_signbitf(t.inv.y) ? 2u : 0u) | (__builtin_signbitf(t.inv.z) ? 1u : 0u)];
  }
}

template <bool COUNT>
DEV bool medium_test(const DevScene& S, uint32_t midx, f3 o, f3 d, float tmin, float tmax, float& t_out, const Stack& st,
                     int sp, const Rng& rng, uint32_t depth, Counters& cnt);

// Decodes child `i` (compile-time) of a wide node and tests it; sets bit i of `hits` when the child must be visited.
// The slab test runs in t-space: plane q of an axis is crossed at  t = A + q * B  with  A = (origin - o) * inv,
// B = scale * inv  (one FMA per plane), and the near / far byte arrays of each axis are chosen once per node from the sign of
// the ray direction. This is NOT the reference's (b - o) * inv sequence and need not be: the boxes are culls, and the
// builder pads them by twice the fp32 box pad, three times the worst rounding error of this evaluation.
// NaN (0 * inf for axis-parallel rays) is ignored by fmaxf / fminf, i.e. treated as "no constraint": conservative.
#define SOL_WIDE_CHILD(i, nxw, nyw, nzw, fxw, fyw, fzw, refv)                                                          \
  {                                                                                                                     \
    const float tnx = fmaf((float)(((nxw) >> (8 * ((i) & 3))) & 0xFFu), bx, ax), tfx = fmaf((float)(((fxw) >> (8 * ((i) & 3))) & 0xFFu), bx, ax); \
    const float tny = fmaf((float)(((nyw) >> (8 * ((i) & 3))) & 0xFFu), by, ay), tfy = fmaf((float)(((fyw) >> (8 * ((i) & 3))) & 0xFFu), by, ay); \
    const float tnz = fmaf((float)(((nzw) >> (8 * ((i) & 3))) & 0xFFu), bz, az), tfz = fmaf((float)(((fzw) >> (8 * ((i) & 3))) & 0xFFu), bz, az); \
    const float te = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));                                                         \
    const float tx = fminf(fminf(tfx, tfy), fminf(tfz, cull_t));                                                        \
    hits |= (te <= tx) ? (1u << (i)) : 0u; /* an empty slot has an inverted box; at worst it yields a NONE ref (no-op) */ \
  }
// Second pass. Children are visited in order of k = slot ^ octant (slots are octants of the node's split planes, so this
// is front to back along the ray's direction signs). The nearest hit child becomes the current reference, the others go
// on the stack with the nearest of them on top: level = sp + (number of hit children that are farther). "Farther than
// slot i" is a fixed slot set per ray octant (kWideFar, one byte per slot), so a level costs and + popcount.
// Fast form (whole node fits in the LDS part of the stack): branch-free - a child that was not hit is stored to a scratch
// level above everything live, the nearest child is stored too (at sp + n - 1, the level the stack does not keep).
#define SOL_WIDE_PLACE_FAST(i, refv, farw)                                                                   \
  {                                                                                                          \
    const uint32_t pos = __popc(hits & (((farw) >> (8 * ((i) & 3))) & 0xFFu));                                \
    const uint32_t lvl = (hits & (1u << (i))) ? pos : 8u;                                                     \
    base[lvl * SOL_WG] = (refv);                                                                             \
    nearest = (lvl == n_hit - 1u) ? (refv) : nearest;                                                        \
  }
#define SOL_WIDE_PLACE(i, refv, farw)                                                                        \
  if (hits & (1u << (i))) {                                                                                  \
    const uint32_t pos = __popc(hits & (((farw) >> (8 * ((i) & 3))) & 0xFFu));                                \
    if (pos == n_hit - 1u) cur = (refv);                                                                     \
    else stack_store(st, t.sp + (int)pos, (refv));                                                           \
  }

// One step: visits the node or primitive t.cur, then moves to the next reference (near child, or popped from the stack).
// BINARY selects which inner-node kind this search walks: the 2-wide DNode tree (constant-medium boundaries, whose
// search interval includes negative t) or the 8-wide DWide tree (the world).
template <bool COUNT, bool MEDIUM, bool BINARY>
DEV void trav_step(const DevScene& S, Trav& t, const Stack& st, const Rng& rng, uint32_t depth, Counters& cnt) {
  uint32_t cur = t.cur;
  uint32_t kind = SOL_REF_KIND(cur);
  phase_tick<COUNT>(cnt, 0);
  // Part 1: an inner node, if the lane is at one; its nearest hit child (or, when no child is hit, the top of the stack)
  // becomes `cur`. Part 2: a primitive, if `cur` is one now. A lane so advances by up to two visits per step, while a wave
  // whose lanes are spread over nodes and primitives pays for both parts on every step anyway.
  if (!BINARY && kind == SOL_REF_WIDE) {
    const uint32_t idx = SOL_REF_INDEX(cur);
    const bool sx = __builtin_signbitf(t.inv.x), sy = __builtin_signbitf(t.inv.y), sz = __builtin_signbitf(t.inv.z);
    const float4* wp = reinterpret_cast<const float4*>(S.wides + idx);
    const float4 h = wp[0];
    const uint4 qa = *reinterpret_cast<const uint4*>(wp + 1), qb = *reinterpret_cast<const uint4*>(wp + 2);
    const uint4 qc = *reinterpret_cast<const uint4*>(wp + 3);
    const uint4 ra = *reinterpret_cast<const uint4*>(wp + 4), rb = *reinterpret_cast<const uint4*>(wp + 5);
    if (COUNT) cnt.node_visits++;
    const uint32_t meta = __float_as_uint(h.w);
    const float scx = __uint_as_float((meta & 0xFFu) << 23), scy = __uint_as_float(((meta >> 8) & 0xFFu) << 23);
    const float scz = __uint_as_float(((meta >> 16) & 0xFFu) << 23);
#ifdef SOL_NO_TCULL
    const float cull_t = __builtin_huge_valf();
#else
    const float cull_t = t.h.t;  // t >= tmin > 0 in a world search
#endif
    uint32_t hits = 0u;
    // An exactly zero direction component gives inv = inf, and A + q * B = -inf + inf = NaN for every plane: "no constraint",
    // i.e. the ray would visit every node. Clamped to +-1e30 the axis becomes the containment test it should be (origin
    // inside the slab: planes at -+huge; outside: both planes at the same huge sign -> culled).
    const float ivx = __builtin_amdgcn_fmed3f(t.inv.x, -1e30f, 1e30f), ivy = __builtin_amdgcn_fmed3f(t.inv.y, -1e30f, 1e30f);
    const float ivz = __builtin_amdgcn_fmed3f(t.inv.z, -1e30f, 1e30f);
    const float ax = (h.x - t.o.x) * ivx, bx = scx * ivx;
    const float ay = (h.y - t.o.y) * ivy, by = scy * ivy;
    const float az = (h.z - t.o.z) * ivz, bz = scz * ivz;
    // q words: qa = {lo_x[0..3], lo_x[4..7], lo_y[0..3], lo_y[4..7]}, qb = {lo_z.., lo_z.., hi_x.., hi_x..},
    //          qc = {hi_y.., hi_y.., hi_z.., hi_z..}; near = the plane the ray meets first on that axis
    const uint32_t nx0 = sx ? qb.z : qa.x, nx1 = sx ? qb.w : qa.y, fx0 = sx ? qa.x : qb.z, fx1 = sx ? qa.y : qb.w;
    const uint32_t ny0 = sy ? qc.x : qa.z, ny1 = sy ? qc.y : qa.w, fy0 = sy ? qa.z : qc.x, fy1 = sy ? qa.w : qc.y;
    const uint32_t nz0 = sz ? qc.z : qb.x, nz1 = sz ? qc.w : qb.y, fz0 = sz ? qb.x : qc.z, fz1 = sz ? qb.y : qc.w;
    SOL_WIDE_CHILD(0, nx0, ny0, nz0, fx0, fy0, fz0, ra.x)
    SOL_WIDE_CHILD(1, nx0, ny0, nz0, fx0, fy0, fz0, ra.y)
    SOL_WIDE_CHILD(2, nx0, ny0, nz0, fx0, fy0, fz0, ra.z)
    SOL_WIDE_CHILD(3, nx0, ny0, nz0, fx0, fy0, fz0, ra.w)
    SOL_WIDE_CHILD(4, nx1, ny1, nz1, fx1, fy1, fz1, rb.x)
    SOL_WIDE_CHILD(5, nx1, ny1, nz1, fx1, fy1, fz1, rb.y)
    SOL_WIDE_CHILD(6, nx1, ny1, nz1, fx1, fy1, fz1, rb.z)
    SOL_WIDE_CHILD(7, nx1, ny1, nz1, fx1, fy1, fz1, rb.w)
    // keeps the loads of the references with the loads of the boxes (the compiler would sink them behind the branch, a
    // second dependent memory round trip per node)
    asm volatile("" ::"v"(ra.x), "v"(ra.y), "v"(ra.z), "v"(ra.w), "v"(rb.x), "v"(rb.y), "v"(rb.z), "v"(rb.w));
    if (hits != 0u) {
      const uint32_t n_hit = __popc(hits);
      if (t.sp + 9 <= st.depth) {
        uint32_t* base = st.lds + t.sp * SOL_WG;
        uint32_t nearest = 0u;
        SOL_WIDE_PLACE_FAST(0, ra.x, t.far.x) SOL_WIDE_PLACE_FAST(1, ra.y, t.far.x) SOL_WIDE_PLACE_FAST(2, ra.z, t.far.x)
        SOL_WIDE_PLACE_FAST(3, ra.w, t.far.x) SOL_WIDE_PLACE_FAST(4, rb.x, t.far.y) SOL_WIDE_PLACE_FAST(5, rb.y, t.far.y)
        SOL_WIDE_PLACE_FAST(6, rb.z, t.far.y) SOL_WIDE_PLACE_FAST(7, rb.w, t.far.y)
        cur = nearest;
      } else {  // the node may reach the spill area: generic stores
        SOL_WIDE_PLACE(0, ra.x, t.far.x) SOL_WIDE_PLACE(1, ra.y, t.far.x) SOL_WIDE_PLACE(2, ra.z, t.far.x)
        SOL_WIDE_PLACE(3, ra.w, t.far.x) SOL_WIDE_PLACE(4, rb.x, t.far.y) SOL_WIDE_PLACE(5, rb.y, t.far.y)
        SOL_WIDE_PLACE(6, rb.z, t.far.y) SOL_WIDE_PLACE(7, rb.w, t.far.y)
      }
      t.sp += (int)n_hit - 1;
      if (COUNT) cnt.max_stack = max(cnt.max_stack, (uint32_t)t.sp);
    } else {
      cur = (t.sp == t.sp_base) ? REF_DONE : stack_pop(st, t.sp);
    }
    kind = SOL_REF_KIND(cur);
  } else if (BINARY && kind == SOL_REF_NODE) {
    const uint32_t idx = SOL_REF_INDEX(cur);
    const bool sx = __builtin_signbitf(t.inv.x), sy = __builtin_signbitf(t.inv.y), sz = __builtin_signbitf(t.inv.z);
    const float4* np = reinterpret_cast<const float4*>(S.nodes + idx);
    const float4 a = np[0], b = np[1], c = np[2];
    const uint4 r = *reinterpret_cast<const uint4*>(np + 3);
    if (COUNT) cnt.node_visits++;
    // Culling: a box whose entry parameter lies beyond the best hit cannot hold a better one. The slab test clamps the
    // entry to 0 (origin inside the box), and a search over (-inf, inf) (constant-medium boundary) accepts hits at
    // negative t, so boxes entered at 0 are never culled.
#ifdef SOL_NO_TCULL
    const float cull_t = __builtin_huge_valf();
#else
    const float cull_t = fmaxf(t.h.t, 0.0f);
#endif
    float tl, tr;
    bool hl = slab(a.x, a.y, a.z, a.w, b.x, b.y, t.o, t.inv, sx, sy, sz, tl) && tl <= cull_t;
    bool hr = slab(b.z, b.w, c.x, c.y, c.z, c.w, t.o, t.inv, sx, sy, sz, tr) && tr <= cull_t;
    if (t.tmin < 0.0f) {
      // r.z bit0/bit1: that child's box is the primitive's own box, which the reference does not test (a two-leaf
      // `Bvh` tests only its union box, bvh.rs:91-96). For t >= 0 the extra test is a sound cull; a primitive hit at
      // negative t lies outside the forward slab, so here the child is taken whenever the node was reached.
      if (r.z & 1u) { hl = true; tl = 0.0f; }
      if (r.z & 2u) { hr = true; tr = 0.0f; }
    }
    if (hl && hr) {
      const bool lfirst = tl <= tr;
      stack_push(st, t.sp, lfirst ? r.y : r.x);
      if (COUNT) cnt.max_stack = max(cnt.max_stack, (uint32_t)t.sp);
      cur = lfirst ? r.x : r.y;
    } else if (hl) {
      cur = r.x;
    } else if (hr) {
      cur = r.y;
    } else {
      cur = (t.sp == t.sp_base) ? REF_DONE : stack_pop(st, t.sp);
    }
    kind = SOL_REF_KIND(cur);
  }
  const bool is_inner = cur != REF_DONE && kind == (BINARY ? SOL_REF_NODE : SOL_REF_WIDE);
#if SOL_PRIM_MIN > 1
  // Postponed primitive tests (world search only): the lanes holding a primitive wait - keep it as their current reference -
  // while fewer than SOL_PRIM_MIN of the wave's lanes do and some lane still has an inner node to visit next turn; the
  // primitive part then runs with more lanes enabled. Results do not depend on the order of the tests.
  unsigned long long inner_m = 0ull, prim_m = 0ull;
  if (!BINARY) {
    inner_m = __ballot(is_inner);
    prim_m = __ballot(cur != REF_DONE && !is_inner);  // (counting per primitive kind instead was measured: no better)
  }
#endif
  if (cur == REF_DONE || is_inner) { t.cur = cur; return; }
#if SOL_PRIM_MIN > 1
  if (!BINARY && inner_m != 0ull && (int)__popcll(prim_m) < SOL_PRIM_MIN) { t.cur = cur; return; }
#endif
  const uint32_t idx = SOL_REF_INDEX(cur);
  if (kind == SOL_REF_TRIANGLE) {
    const float4* tp = reinterpret_cast<const float4*>(S.tris + idx);
    const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
    DTri T;
    T.v0x = p0.x; T.v0y = p0.y; T.v0z = p0.z; T.e1x = p0.w; T.e1y = p1.x; T.e1z = p1.y; T.e2x = p1.z; T.e2y = p1.w; T.e2z = p2.x;
    const uint32_t dfs = __float_as_uint(p2.y);
    if (COUNT) cnt.triangle_tests++;
    float tt, u, v;
    if (tri_test(T, t.o, t.d, t.tmin, t.h.t, tt, u, v) && better(tt, dfs, t.h)) { t.h.t = tt; t.h.ref = cur; t.h.dfs = dfs; t.h.u = u; t.h.v = v; }
  } else if (kind == SOL_REF_SPHERE) {
    const DSphere Sp = S.spheres[idx];
    if (COUNT) cnt.sphere_tests++;
    float tt;
    if (sphere_test(Sp, t.o, t.d, t.tmin, t.h.t, S.sphere_slack, tt) && better(tt, Sp.dfs, t.h)) { t.h.t = tt; t.h.ref = cur; t.h.dfs = Sp.dfs; }
  } else if (kind == SOL_REF_QUAD) {
    const DQuad Q = S.quads[idx];
    if (COUNT) cnt.quad_tests++;
    float tt, u, v;
    if (quad_test(Q, t.o, t.d, t.tmin, t.h.t, tt, u, v) && better(tt, Q.dfs, t.h)) { t.h.t = tt; t.h.ref = cur; t.h.dfs = Q.dfs; t.h.u = u; t.h.v = v; }
  } else if (MEDIUM && kind == SOL_REF_MEDIUM) {
    float tt;
    const uint32_t dfs = S.mediums[idx].dfs;
    if (medium_test<COUNT>(S, idx, t.o, t.d, t.tmin, t.h.t, tt, st, t.sp, rng, depth, cnt) && better(tt, dfs, t.h)) {
      t.h.t = tt; t.h.ref = cur; t.h.dfs = dfs;
    }
  }
  t.cur = (t.sp == t.sp_base) ? REF_DONE : stack_pop(st, t.sp);
}

// Run-to-completion form.
template <bool COUNT, bool MEDIUM, bool BINARY>
DEV void closest_hit(const DevScene& S, f3 o, f3 d, float tmin, float tmax, uint32_t root, float bxmin, float bxmax,
                     float bymin, float bymax, float bzmin, float bzmax, Hit& h, const Stack& st, int sp_base, const Rng& rng,
                     uint32_t depth, Counters& cnt) {
  Trav t;
  trav_begin(t, o, d, tmin, tmax, root, bxmin, bxmax, bymin, bymax, bzmin, bzmax, sp_base);
  while (t.cur != REF_DONE) trav_step<COUNT, MEDIUM, BINARY>(S, t, st, rng, depth, cnt);
  h = t.h;
}

// ConstantMedium::hit (src/hittable/constant_medium.rs:35-79). Its draws come from the sub-stream
// 0x40000000 + (depth<<20 | medium<<8) + i of the path's generator (DESIGN.md "RNG"), so the outcome does not
// depend on when the tree search reaches the medium.
template <bool COUNT>
DEV bool medium_test(const DevScene& S, uint32_t midx, f3 o, f3 d, float tmin, float tmax, float& t_out, const Stack& st,
                     int sp, const Rng& rng, uint32_t depth, Counters& cnt) {
  const DMedium M = S.mediums[midx];
  const float inf = __builtin_huge_valf();
  Hit h1, h2;
  closest_hit<COUNT, false, true>(S, o, d, -inf, inf, M.boundary, M.bxmin, M.bxmax, M.bymin, M.bymax, M.bzmin, M.bzmax, h1, st, sp,
                            rng, depth, cnt);
  if (SOL_REF_KIND(h1.ref) == SOL_REF_NONE) return false;
  closest_hit<COUNT, false, true>(S, o, d, h1.t + 0.0001f, inf, M.boundary, M.bxmin, M.bxmax, M.bymin, M.bymax, M.bzmin, M.bzmax,
                            h2, st, sp, rng, depth, cnt);
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
