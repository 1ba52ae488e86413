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

// The world (searches with t >= 0.001) is walked through the 7-wide quantised tree (DWide); the boundary searches of a constant
// medium, whose interval includes negative t, walk the reference-shaped 2-wide tree (DNode).

#define REF_DONE 0xFFFFFFFFu
#define SOL_INV_CLAMP 1e18f  // |1 / direction| as the 7-wide node test sees it (wide_node_test: why 1e18)
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
  uint32_t primary_hits;  // samples whose camera ray hit something
  uint32_t path_len[6];   // samples by the number of rays of their path: 1, 2, 3-4, 5-8, 9-16, 17 and more
};

// Lane-utilisation instrumentation (COUNT builds only): every active lane counts itself, the first active lane of the
// wave counts the 64 slots of this execution.
template <bool COUNT>
DEV void phase_tick(Counters& cnt, int ph) {
  if (COUNT) {
    cnt.phase[2 * ph]++;
    const unsigned long long m = sol_ballot(true);
    if ((int)__lane_id() == __ffsll((long long)m) - 1) cnt.phase[2 * ph + 1] += 64u;
  }
}

// The two halves of the stack live in different address spaces and are typed so: a pointer chosen between them at run time
// would be generic, and the pop of EVERY step a flat_load (vmcnt + lgkmcnt, aperture check) instead of a ds_read.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
#define SOL_OCT_TABLE_BYTES 1024  // [octant][7-bit slot mask] -> the mask in visit order (sol_fill_oct_table)
struct Stack {
  lds_u32* lds;              // base of this workgroup's [depth][SOL_WG] array, already offset by the lane
  SOL_AS1 uint32_t* spill;   // base of the global spill area, already offset by the global thread id
  uint32_t stride;           // total threads (spill stride between levels)
  int depth;                 // entries per lane kept in LDS; deeper entries go to the spill area. A kernel built for trees
                             // that fit the LDS stack sets it to a huge constant: every spill branch folds away
  // The three scene fields every step of a world search needs, copied out of the DevScene record once per kernel (sol_search_
  // context): read through the record, the compiler re-loads them with s_load + s_waitcnt in front of EVERY node fetch (it has
  // no scalar registers left to keep them), a scalar-cache round trip on the critical path of each visit.
  const DWide* wides;
  const DTri* tris;
  uint32_t wide_emin;
  // Slot mask -> visit-order mask of a 7-wide node (bit p <- bit p ^ octant) as a 1 KiB table in LDS, or nullptr-equivalent
  // (oct_table_on = false): three conditional butterfly stages, 14 four-cycle vector instructions per node visit. The kernel is
  // bound by vector-instruction issue and its LDS pipe idles, so the render kernel trades them for one ds_read_u8.
  const lds_u8* oct_table;
  bool oct_table_on;
};
// Fills a workgroup's octant table (all threads call it; the caller synchronises the workgroup before the first search).
DEV void sol_fill_oct_table(lds_u8* tbl, uint32_t tid, uint32_t n_threads) {
  for (uint32_t i = tid; i < SOL_OCT_TABLE_BYTES; i += n_threads) {
    const uint32_t oct = i >> 7, m = i & 127u;
    uint32_t r = 0u;
    for (uint32_t p = 0; p < 8u; ++p) r |= ((m >> (p ^ oct)) & 1u) << p;
    tbl[i] = (uint8_t)r;
  }
}
// Fills the scene fields of a search context. PIN keeps the node fields in vector registers (3 VGPRs) by hiding where they came from.
template <bool PIN>
DEV void sol_search_context(Stack& st, const DevScene& S) {
  unsigned long long w = (unsigned long long)S.wides, tr = (unsigned long long)S.tris;
  uint32_t e = S.wide_emin;
  if (PIN) asm volatile("" : "+v"(w), "+v"(e));  // (the triangle pointer too would cost the kernel its last registers: 3 spills)
  st.wides = (const DWide*)w;
  st.tris = (const DTri*)tr;
  st.wide_emin = e + 24u;  // (the node test reads plane bytes as the halves q * 2^-24: the scales carry the 2^24)
  st.oct_table = nullptr;
  st.oct_table_on = false;
}
#define SOL_NO_SPILL 0x3FFFFFFF  // Stack::depth of a kernel built for searches that fit the LDS stack (a compile-time constant there)
DEV void stack_store(const Stack& s, int level, uint32_t v) {
  if (s.depth >= SOL_NO_SPILL || level < s.depth) s.lds[level * SOL_WG] = v;
  else s.spill[(size_t)(level - s.depth) * s.stride] = v;
}
DEV void stack_push(const Stack& s, int& sp, uint32_t v) {
  stack_store(s, sp, v);
  sp++;
}
// Pops one two-dword entry (pushed as `first`, then `second`): in a kernel without a spill area one address and one
// ds_read2st64_b32 instead of two of each.
DEV void stack_pop2(const Stack& s, int& sp, uint32_t& first, uint32_t& second) {
  sp -= 2;
  if (s.depth >= SOL_NO_SPILL) {
    const lds_u32* p = s.lds + sp * SOL_WG;
    first = p[0];
    second = p[SOL_WG];
  } else {
    first = sp < s.depth ? s.lds[sp * SOL_WG] : s.spill[(size_t)(sp - s.depth) * s.stride];
    second = sp + 1 < s.depth ? s.lds[(sp + 1) * SOL_WG] : s.spill[(size_t)(sp + 1 - s.depth) * s.stride];
  }
}
DEV uint32_t stack_pop(const Stack& s, int& sp) {
  sp--;
  uint32_t v;
  if (s.depth >= SOL_NO_SPILL || sp < s.depth) v = s.lds[sp * SOL_WG];
  else v = s.spill[(size_t)(sp - s.depth) * s.stride];
  return v;
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
  // fp32 contract, sixth rule (oracle.cpp hit_sphere): the discriminant from the distance of the centre to the ray, the roots as q / a and
  // c / q - the reference's half_b^2 - a c loses the digits of a distant origin (C2 +3.5 % for it, every other workload unchanged)
  const float k = half_b / a;
  const f3 l = oc - d * k;  // from the centre to the ray's closest point
  const float disc1 = S.radius * S.radius - len2(l);
  if (disc1 < 0.0f) return false;
  const float sq = sol_sqrt(a * disc1);
  const float q = half_b >= 0.0f ? -half_b - sq : -half_b + sq;
  const float rq = q / a, rc = c / q;
  float root = half_b >= 0.0f ? rq : rc;
  if (!sphere_root_ok(S, o, d, root, tmin, tmax, slack)) {
    root = half_b >= 0.0f ? rc : rq;
    if (!sphere_root_ok(S, o, d, root, tmin, tmax, slack)) return false;
  }
  t = root;
  return true;
}

DEV bool better(float t, uint32_t dfs, const Hit& h) {
  return t < h.t || (t == h.t && (SOL_REF_KIND(h.ref) == SOL_REF_NONE || dfs > h.dfs));
}
// fp32 contract, scenes with needle triangles (include/solstrale_hip.h): the ray's point o + t*d and the triangle's point
// v0 + u*e1 + v*e2 of a hit agree within `delta` (0.8 box pads) in every coordinate - the operation order of oracle.cpp, hit_triangle.
DEV bool tri_hit_consistent(const DTri& T, f3 o, f3 d, float t, float u, float v, float delta) {
  const f3 p = o + d * t;
  const f3 q = mk3(T.v0x, T.v0y, T.v0z) + mk3(T.e1x, T.e1y, T.e1z) * u + mk3(T.e2x, T.e2y, T.e2z) * v;
  const f3 dl = p - q;
  return fabsf(dl.x) <= delta && fabsf(dl.y) <= delta && fabsf(dl.z) <= delta;
}

// State of one closest-hit search.
//
// Searches of the 2-wide tree (constant-medium boundaries; BINARY) keep one reference per stack entry and `cur` = the
// reference being visited. Searches of the 7-wide tree (the world) keep GROUPS, after Ylitie, Karras & Laine (2017, sec. 5):
//   node group  g0 = base_inner | ordered hits << 24,  g1 = the node's meta word (imask)  - the inner children of one node that the ray's slab test
//               hit and that have not been visited yet. "Ordered": bit p stands for slot p ^ octant, so the lowest set bit is
//               the nearest child. Visiting a child pushes what is left of its parent's group: ONE two-dword stack entry per
//               level instead of one entry per child (the first layout: eight ds_write_b32 and ~70 placement instructions per
//               visit).
//   prim group  pg = base_prim | hit leaf slots << 24  (leaf kind and lmask: g1, the same node's meta word) - the hit primitives of
//               the node visited last; they are tested before the search descends further.
// `cur` is only the status of such a search: REF_DONE when it is over, 0 while it runs.
//
// ConstantMedium::hit (constant_medium.rs:35-79) is two closest-hit searches of the medium's boundary. A world search that reaches
// a medium keeps them as a SUB-STATE (MedSearch, kernels built with MEDIUM only) and advances it one boundary step per turn of the
// wave's search loop, beside the other lanes' world steps, instead of running both searches to completion inside one primitive
// test while the rest of the wave idles (round 3: lane utilisation 0.22 on the reference's own profiling workload).
struct MedSearch {
  uint32_t phase;  // 0: no medium test running; 1 / 2: the first / second boundary search
  uint32_t midx;   // the medium
  uint32_t cur;    // boundary search: the reference being visited (REF_DONE: over)
  int sp;          // its stack pointer; its entries lie above the world search's (base = Trav::sp, which rests meanwhile)
  float tmin;      // lower end of its interval: -inf, then t1 + 0.0001
  float t, t1;     // its best hit so far; the first search's hit
  uint32_t ref, dfs;
  f3 inv;          // 1 / d as Ray::new makes it (the world search's copy is clamped for the 7-wide node test)
};
struct Trav {
  f3 o, d, inv;
  float tmin;
  uint32_t cur;
  int sp, sp_base;
  uint32_t g0, g1, pg, oct;  // (7-wide searches only; oct = the ray's octant, sign bits of the direction as x<<2 | y<<1 | z)
  Hit h;
  MedSearch m;               // (MEDIUM kernels only; dead otherwise)
  // (STRICT kernels only: scenes with needle triangles) a search that follows a closest hit the consistency rule refused looks only
  // for triangle hits BEHIND that one in the order of `better`: t > bt, or t == bt with an earlier dfs index
  float bt;
  uint32_t bdfs;
};

// Starts a search of `root` over [tmin, tmax]; (bxmin..bzmax) is root's own box, tested first when root is a node
// (Bvh::hit, bvh.rs:166). WIDE: root is an index into DevScene::wides, else a reference into the 2-wide tree.
template <bool WIDE>
DEV void trav_begin(Trav& t, f3 o, f3 d, float tmin, float tmax, uint32_t root, float bxmin, float bxmax, float bymin,
                    float bymax, float bzmin, float bzmax, int sp_base) {
  t.o = o; t.d = d; t.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // Ray::new (geo/mod.rs:277-285)
  t.tmin = tmin;
  t.h.t = tmax;
  t.h.ref = SOL_MAKE_REF(SOL_REF_NONE, 0);
  t.h.dfs = 0;
  t.h.u = t.h.v = 0.0f;
  t.sp = t.sp_base = sp_base;
  t.cur = WIDE ? 0u : root;
  // the root as a group of one: imask 0 makes every slot's rank 0, so the "child" picked from it is node `root` itself
  t.g0 = root | (1u << 24);
  t.g1 = 0u;
  t.pg = 0u;
  t.m.phase = 0u;
  t.bt = -__builtin_huge_valf();  // (a search behind a refused hit sets its bound after this call)
  t.bdfs = 0u;
  t.oct = (__builtin_signbitf(t.inv.x) ? 4u : 0u) | (__builtin_signbitf(t.inv.y) ? 2u : 0u) | (__builtin_signbitf(t.inv.z) ? 1u : 0u);
  // A ray with a NaN in its origin or direction cannot hit anything: every primitive test ends in a comparison with NaN,
  // which is false (the reference returns None the same way, after visiting every box - Aabb::hit ignores NaN). Such rays
  // occur a few times per 10^8 samples; without this exit one lane walks the whole tree and tests every primitive.
  if (isnan(d.x) || isnan(d.y) || isnan(d.z) || isnan(o.x) || isnan(o.y) || isnan(o.z)) { t.cur = REF_DONE; return; }
  if (WIDE || SOL_REF_KIND(root) == SOL_REF_NODE) {
    float te;
    if (!slab(bxmin, bxmax, bymin, bymax, bzmin, bzmax, o, t.inv, __builtin_signbitf(t.inv.x), __builtin_signbitf(t.inv.y),
              __builtin_signbitf(t.inv.z), te))
      t.cur = REF_DONE;
  }
  // A 7-wide search evaluates its slabs in t-space with the inverse direction clamped to +-SOL_INV_CLAMP (wide_node_test: an exactly
  // zero direction component becomes the containment test it should be instead of NaN); clamped here, once per ray, not once per node
  if (WIDE) t.inv = mk3(__builtin_amdgcn_fmed3f(t.inv.x, -SOL_INV_CLAMP, SOL_INV_CLAMP), __builtin_amdgcn_fmed3f(t.inv.y, -SOL_INV_CLAMP, SOL_INV_CLAMP),
                        __builtin_amdgcn_fmed3f(t.inv.z, -SOL_INV_CLAMP, SOL_INV_CLAMP));
}

// Decodes child `i` (compile-time) of a wide node and tests it; sets bit i of `hits` when the child must be visited.
// The slab test runs in t-space: plane q of an axis is crossed at  t = A + q * B  with  A = (origin - o) * inv,
// B = scale * inv  (one FMA per plane), and the near / far byte arrays of each axis are chosen once per node from the sign of
// the ray direction. This is NOT the reference's (b - o) * inv sequence and need not be: the boxes are culls, and the
// builder pads them by twice the fp32 box pad, three times the worst rounding error of this evaluation.
//
// The slab parameters are evaluated in units of the cull distance, t' = t / min(best t, 1e30): the search interval is then [0, 1]
// and the CLAMP output modifier of the plane FMAs does what a maximum with 0 (near planes) and a minimum with the cull distance
// (far planes) did - two of the twelve vector instructions per child. A child must be visited iff te' < tx', STRICTLY: a box wholly
// beyond the cull distance clamps to te' = tx' = 1, one wholly behind the origin to 0 = 0. (A box holding a primitive that a ray
// really hits has te < tx by the builder's margin of three pads, 2.7 times the rounding error of this evaluation - sol_tree.h -, so
// the strict comparison loses nothing; the scale is taken a millionth short of 1 / cull so that a tie at t = cull stays inside.)
// NaN clamps to 0 under DX10_CLAMP: no constraint for a near plane. No plane parameter may OVERFLOW, or a box the ray is inside of
// would read as culled (-inf + finite: far plane 0): with the inverse direction clamped to SOL_INV_CLAMP = 1e18 (trav_begin), the cull
// scale at most 1 / RAY_MIN = 1e3 and the scene's coordinates below 2^38 (sol_scene_create refuses larger ones), the addend stays
// below 2^39 * 1e21 = 5.5e32 and the slope - a node scale of at most 2^32 times the 2^24 of the subnormal halves - below 2^56 * 1e21 =
// 7.2e37 < FLT_MAX. (Round 3 clamped to 1e30: an axis-parallel ray - direction component exactly 0 - lost real hits in
// nodes further than ~3e5 from its origin, and whenever the node scale times the cull scale passed 3e8. A smaller clamp only makes
// such a ray's test more conservative: the containment reading needs |plane - origin| * clamp * cull scale >= 1.)
#define SOL_FMA_MIX_LO(d, h2, b, a) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0] clamp" : "=v"(d) : "v"(h2), "v"(b), "v"(a))
#define SOL_FMA_MIX_HI(d, h2, b, a) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(d) : "v"(h2), "v"(b), "v"(a))
#define SOL_FMA_MIX_x SOL_FMA_MIX_LO
#define SOL_FMA_MIX_y SOL_FMA_MIX_HI
// Two plane bytes become two halves by ONE byte permute that puts a zero byte above each: the fp16 SUBNORMALS q * 2^-24, exact, which
// v_fma_mix_f32 takes as its first factor (fp16 denormals are enabled in the kernel's float mode); the node's scales carry the 2^24
// (sol_search_context: wide_emin + 24), so t = q * B + A with a single rounding - no offset of 1024 to take out of the addend again.
typedef uint32_t sol_h2;
#define SOL_H2(w, sel) __builtin_amdgcn_perm(0u, (w), (sel) | 0x0C000C00u)
#define SOL_WIDE_CHILD(i, hnx, hny, hnz, hfx, hfy, hfz, e)                                                             \
  {                                                                                                                     \
    float tnx, tfx, tny, tfy, tnz, tfz;                                                                                 \
    SOL_FMA_MIX_##e(tnx, hnx, bx, ax); SOL_FMA_MIX_##e(tfx, hfx, bx, ax);                                               \
    SOL_FMA_MIX_##e(tny, hny, by, ay); SOL_FMA_MIX_##e(tfy, hfy, by, ay);                                               \
    SOL_FMA_MIX_##e(tnz, hnz, bz, az); SOL_FMA_MIX_##e(tfz, hfz, bz, az);                                               \
    float te, tx;                                                                                                        \
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(te) : "v"(tnx), "v"(tny), "v"(tnz));                                         \
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tx) : "v"(tfx), "v"(tfy), "v"(tfz));                                         \
    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(te - tx), 31u); /* (here: the HIT bits: sign of te - tx) */   \
  }
// The slab tests of one fetched 7-wide node (h = origin + meta, qa / qb / qc = the six plane arrays) for the ray of search `t`:
// the search's new node group and primitive group.
template <bool COUNT>
DEV void wide_node_test(const Stack& st, Trav& t, uint32_t oct, float4 h, uint4 qa, uint4 qb, uint4 qc) {
  const uint32_t wide_emin = st.wide_emin;
  const bool sx = (oct & 4u) != 0u, sy = (oct & 2u) != 0u, sz = (oct & 1u) != 0u;
  const uint32_t meta = __float_as_uint(h.w);
  const float scx = __uint_as_float(((meta & 31u) + wide_emin) << 23), scy = __uint_as_float((((meta >> 5) & 31u) + wide_emin) << 23);
  const float scz = __uint_as_float((((meta >> 10) & 31u) + wide_emin) << 23);
  uint32_t miss = 0u;  // children are tested 6 .. 0, each shifting its bit in at the bottom: child i ends on bit i
  // An exactly zero direction component gives inv = inf, and A + q * B = -inf + inf = NaN for every plane: "no constraint",
  // i.e. the ray would visit every node. Clamped to +-SOL_INV_CLAMP the axis becomes the containment test it should be (origin
  // inside the slab: planes at -+huge; outside: both planes at the same huge sign -> culled).
  // (units of the cull distance, a millionth short: see SOL_WIDE_CHILD; best t is a positive float or +inf, never NaN)
  const float rc = __builtin_amdgcn_rcpf(__builtin_amdgcn_fmed3f(t.h.t, RAY_MIN_F, 1e30f) * 1.000001f);
  const float ivx = t.inv.x * rc, ivy = t.inv.y * rc, ivz = t.inv.z * rc;  // (t.inv: clamped to +-SOL_INV_CLAMP by trav_begin)
  const float ax = (h.x - t.o.x) * ivx, bx = scx * ivx;
  const float ay = (h.y - t.o.y) * ivy, by = scy * ivy;
  const float az = (h.z - t.o.z) * ivz, bz = scz * ivz;
  // q words: qa = {lo_x[0..3], lo_x[4..7], lo_y[0..3], lo_y[4..7]}, qb = {lo_z.., lo_z.., hi_x.., hi_x..},
  //          qc = {hi_y.., hi_y.., hi_z.., hi_z..}; near = the plane the ray meets first on that axis
  const uint32_t nx0 = sx ? qb.z : qa.x, nx1 = sx ? qb.w : qa.y, fx0 = sx ? qa.x : qb.z, fx1 = sx ? qa.y : qb.w;
  const uint32_t ny0 = sy ? qc.x : qa.z, ny1 = sy ? qc.y : qa.w, fy0 = sy ? qa.z : qc.x, fy1 = sy ? qa.w : qc.y;
  const uint32_t nz0 = sz ? qc.z : qb.x, nz1 = sz ? qc.w : qb.y, fz0 = sz ? qb.x : qc.z, fz1 = sz ? qb.y : qc.w;
  {
    const sol_h2 hnx01 = SOL_H2(nx0, 0x04010400u), hnx23 = SOL_H2(nx0, 0x04030402u), hnx45 = SOL_H2(nx1, 0x04010400u), hnx6 = SOL_H2(nx1, 0x04030402u);
    const sol_h2 hny01 = SOL_H2(ny0, 0x04010400u), hny23 = SOL_H2(ny0, 0x04030402u), hny45 = SOL_H2(ny1, 0x04010400u), hny6 = SOL_H2(ny1, 0x04030402u);
    const sol_h2 hnz01 = SOL_H2(nz0, 0x04010400u), hnz23 = SOL_H2(nz0, 0x04030402u), hnz45 = SOL_H2(nz1, 0x04010400u), hnz6 = SOL_H2(nz1, 0x04030402u);
    const sol_h2 hfx01 = SOL_H2(fx0, 0x04010400u), hfx23 = SOL_H2(fx0, 0x04030402u), hfx45 = SOL_H2(fx1, 0x04010400u), hfx6 = SOL_H2(fx1, 0x04030402u);
    const sol_h2 hfy01 = SOL_H2(fy0, 0x04010400u), hfy23 = SOL_H2(fy0, 0x04030402u), hfy45 = SOL_H2(fy1, 0x04010400u), hfy6 = SOL_H2(fy1, 0x04030402u);
    const sol_h2 hfz01 = SOL_H2(fz0, 0x04010400u), hfz23 = SOL_H2(fz0, 0x04030402u), hfz45 = SOL_H2(fz1, 0x04010400u), hfz6 = SOL_H2(fz1, 0x04030402u);
    SOL_WIDE_CHILD(6, hnx6, hny6, hnz6, hfx6, hfy6, hfz6, x)
    SOL_WIDE_CHILD(5, hnx45, hny45, hnz45, hfx45, hfy45, hfz45, y)
    SOL_WIDE_CHILD(4, hnx45, hny45, hnz45, hfx45, hfy45, hfz45, x)
    SOL_WIDE_CHILD(3, hnx23, hny23, hnz23, hfx23, hfy23, hfz23, y)
    SOL_WIDE_CHILD(2, hnx23, hny23, hnz23, hfx23, hfy23, hfz23, x)
    SOL_WIDE_CHILD(1, hnx01, hny01, hnz01, hfx01, hfy01, hfz01, y)
    SOL_WIDE_CHILD(0, hnx01, hny01, hnz01, hfx01, hfy01, hfz01, x)
  }
  const uint32_t hits = miss;  // (the clamped form shifts in HIT bits)
  const uint32_t imask = (meta >> 15) & 0x7Fu, lmask = (meta >> 22) & 0x7Fu;
  // inner hits into visit order: bit p <- bit p ^ octant (three conditional butterfly stages)
  uint32_t ih = hits & imask;
  if (st.oct_table_on) ih = st.oct_table[(oct << 7) | ih];
  else {
  ih = (oct & 1u) ? (((ih & 0x55u) << 1) | ((ih >> 1) & 0x55u)) : ih;
  ih = (oct & 2u) ? (((ih & 0x33u) << 2) | ((ih >> 2) & 0x33u)) : ih;
  ih = (oct & 4u) ? (((ih & 0x0Fu) << 4) | (ih >> 4)) : ih;
  }
  // base indices ride in the slot-7 bytes of the six plane arrays (lo x, y, z: inner; hi x, y, z: primitives)
  // (two byte permutes each: {top of a, top of b, 0, 0}, then {.., .., top of c, 0})
  const uint32_t base_inner = __builtin_amdgcn_perm(qb.y, __builtin_amdgcn_perm(qa.w, qa.y, 0x0C0C0703u), 0x0C070100u);
  const uint32_t base_prim = __builtin_amdgcn_perm(qc.w, __builtin_amdgcn_perm(qc.y, qb.w, 0x0C0C0703u), 0x0C070100u);
  t.g0 = base_inner | (ih << 24);
  t.g1 = meta;  // (imask, lmask and leaf kind are read from it where they are needed)
  t.pg = base_prim | ((hits & lmask) << 24);
}

// Tests primitive `idx` of kind `kind` against the search's interval [tmin, best t] and keeps it when it is the better hit
// (smaller t; among equal t the later one in depth-first leaf order, bvh.rs:172-178).
// STRICT (scenes with needle triangles): 0 off; 1 = a world search - hits not behind the search's bound (t.bt, t.bdfs) are skipped, the
// consistency of the CLOSEST hit is checked by the caller when the search is over (sol_render.hip); 2 = a boundary search of a constant
// medium, whose every candidate is checked here (delta = DevScene::tri_delta, 0: the scene has no needles).
template <bool COUNT, int STRICT = 0>
DEV void triangle_prim_test(Trav& t, const Stack& st, uint32_t idx, Counters& cnt, float delta = 0.0f) {
  const float4* tp = reinterpret_cast<const float4*>(st.tris + idx);
#if SOL_FETCH_PRIO >= 10
  __builtin_amdgcn_s_setprio(SOL_FETCH_PRIO / 10);
#endif
  const float4 p0 = ldg_f4(tp), p1 = ldg_f4(tp + 1), p2 = ldg_f4(tp + 2);
#if SOL_FETCH_PRIO >= 10
  asm volatile("s_setprio %0" ::"n"(SOL_LOOP_PRIO) : "memory");
#endif
  DTri T;
  T.v0x = p0.x; T.v0y = p0.y; T.v0z = p0.z; T.e1x = p0.w; T.e1y = p1.x; T.e1z = p1.y; T.e2x = p1.z; T.e2y = p1.w; T.e2z = p2.x;
  const uint32_t dfs = __float_as_uint(p2.y);
  if (COUNT) cnt.triangle_tests++;
  float tt, u, v;
  if (tri_test(T, t.o, t.d, t.tmin, t.h.t, tt, u, v) && better(tt, dfs, t.h)) {
    if (STRICT == 1 && (tt < t.bt || (tt == t.bt && dfs >= t.bdfs))) return;
    if (STRICT == 2 && delta > 0.0f && !tri_hit_consistent(T, t.o, t.d, tt, u, v, delta)) return;
    t.h.t = tt; t.h.ref = SOL_MAKE_REF(SOL_REF_TRIANGLE, idx); t.h.dfs = dfs; t.h.u = u; t.h.v = v;
  }
}
template <bool COUNT>
DEV void sphere_prim_test(const DevScene& S, Trav& t, uint32_t idx, Counters& cnt) {
  const DSphere Sp = ldg_rec(S.spheres + idx);
  if (COUNT) cnt.sphere_tests++;
  float tt;
  if (sphere_test(Sp, t.o, t.d, t.tmin, t.h.t, S.sphere_slack, tt) && better(tt, Sp.dfs, t.h)) {
    t.h.t = tt; t.h.ref = SOL_MAKE_REF(SOL_REF_SPHERE, idx); t.h.dfs = Sp.dfs;
  }
}
template <bool COUNT>
DEV void quad_prim_test(const DevScene& S, Trav& t, uint32_t idx, Counters& cnt) {
  const DQuad Q = ldg_rec(S.quads + idx);
  if (COUNT) cnt.quad_tests++;
  float tt, u, v;
  if (quad_test(Q, t.o, t.d, t.tmin, t.h.t, tt, u, v) && better(tt, Q.dfs, t.h)) {
    t.h.t = tt; t.h.ref = SOL_MAKE_REF(SOL_REF_QUAD, idx); t.h.dfs = Q.dfs; t.h.u = u; t.h.v = v;
  }
}
// (a constant medium is not tested here: a world search starts its boundary searches as a sub-state, medium_begin)
template <bool COUNT, int STRICT = 0>
DEV void prim_test(const DevScene& S, Trav& t, const Stack& st, uint32_t kind, uint32_t idx, Counters& cnt) {
  if (kind == SOL_REF_TRIANGLE) {
    triangle_prim_test<COUNT, STRICT>(t, st, idx, cnt, S.tri_delta);
  } else if (kind == SOL_REF_SPHERE) {
    sphere_prim_test<COUNT>(S, t, idx, cnt);
  } else if (kind == SOL_REF_QUAD) {
    quad_prim_test<COUNT>(S, t, idx, cnt);
  }
}

// Part 1 of a step of a 7-wide search, for a lane without pending primitives: takes the nearest child of the lane's node group
// (popping a group first when its own is used up), pushes the rest of the group, fetches that node (64 bytes) and tests its seven
// child boxes: a new node group and a new primitive group.
// PACK: node groups go on the stack as ONE dword - base_inner in 17 bits (trees below 2^17 wide nodes: SOL_PACK_MAX_NODES), the node's
// inner mask in 7, the ordered hit bits in 8 (bit p stands for slot p ^ octant: slots 0..6, positions 0..7) - instead of (g0, meta word):
// half the LDS per level, which is what the pool kernel (sol_pool.hip) pays its path contexts with. A popped group gets its inner mask
// back in the meta word's place; the other fields of the meta word are read only between a node's test and its primitives, never after a pop.
template <bool COUNT, bool PACK = false>
DEV void wide_visit(Trav& t, const Stack& st, Counters& cnt) {
  const uint32_t oct = t.oct;
#if SOL_FETCH_PRIO
  __builtin_amdgcn_s_setprio(SOL_FETCH_PRIO % 10);  // the wave about to fetch goes first: its memory latency starts now
#endif
  uint32_t g0 = t.g0, g1 = t.g1;
  if ((g0 >> 24) == 0u) {  // (a running search without pending primitives has a group here or on the stack)
    if (PACK) {
      const uint32_t e = stack_pop(st, t.sp);
      g0 = (e & (SOL_PACK_MAX_NODES - 1u)) | (e & 0xFF000000u);
      g1 = ((e >> 17) & 0x7Fu) << 15;
    } else {
      stack_pop2(st, t.sp, g0, g1);
    }
  }
  const uint32_t p = (uint32_t)__builtin_ctz(g0 >> 24);  // nearest: lowest bit in visit order
  const uint32_t slot = p ^ oct;
  g0 &= ~(1u << (24u + p));
  const uint32_t idx = (g0 & SOL_WIDE_MAX_INDEX) + __popc(__builtin_amdgcn_ubfe(g1, 15u, slot));  // rank among the node's inner children: imask bits below `slot`
  if ((g0 >> 24) != 0u) {  // siblings left: one stack entry for all of them
    if (PACK) {
      stack_push(st, t.sp, (g0 & (0xFF000000u | (SOL_PACK_MAX_NODES - 1u))) | (((g1 >> 15) & 0x7Fu) << 17));
    } else {
      stack_push(st, t.sp, g0);
      stack_push(st, t.sp, g1);
    }
    if (COUNT) cnt.max_stack = max(cnt.max_stack, (uint32_t)t.sp);
  }
  const float4* wp = reinterpret_cast<const float4*>(st.wides + idx);
  const float4 h = ldg_f4(wp);
  const uint4 qa = ldg_u4(wp + 1), qb = ldg_u4(wp + 2), qc = ldg_u4(wp + 3);
#if SOL_FETCH_PRIO
  asm volatile("s_setprio %0" ::"n"(SOL_LOOP_PRIO) : "memory");  // (after the loads are issued; asm: the builtin may be moved across them)
#endif
  if (COUNT) cnt.node_visits++;
  wide_node_test<COUNT>(st, t, oct, h, qa, qb, qc);
}

// One step of a search of the 2-wide DNode tree (the boundary of a constant medium, whose interval includes negative t): visits
// the node or primitive t.cur, then moves to the next reference (near child, or popped from the stack).
template <bool COUNT>
DEV void boundary_step(const DevScene& S, Trav& t, const Stack& st, Counters& cnt) {
  uint32_t cur = t.cur;
  uint32_t kind = SOL_REF_KIND(cur);
  if (kind == SOL_REF_NODE) {
    const uint32_t idx = SOL_REF_INDEX(cur);
    const bool sx = __builtin_signbitf(t.inv.x), sy = __builtin_signbitf(t.inv.y), sz = __builtin_signbitf(t.inv.z);
    const float4* np = reinterpret_cast<const float4*>(S.nodes + idx);
    const float4 a = ldg_f4(np), b = ldg_f4(np + 1), c = ldg_f4(np + 2);
    const uint4 r = ldg_u4(np + 3);
    if (COUNT) cnt.node_visits++;
    // Culling: a box whose entry parameter lies beyond the best hit cannot hold a better one. The slab test clamps the
    // entry to 0 (origin inside the box), and a search over (-inf, inf) (constant-medium boundary) accepts hits at
    // negative t, so boxes entered at 0 are never culled.
    const float cull_t = fmaxf(t.h.t, 0.0f);
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
  if (cur == REF_DONE || kind == SOL_REF_NODE) { t.cur = cur; return; }
  prim_test<COUNT, 2>(S, t, st, kind, SOL_REF_INDEX(cur), cnt);  // (2: a boundary of needle triangles checks every candidate)
  t.cur = (t.sp == t.sp_base) ? REF_DONE : stack_pop(st, t.sp);
}

// ---- ConstantMedium::hit (src/hittable/constant_medium.rs:35-79) as a sub-state of the world search ----
// Its draws come from the sub-stream 0x40000000 + (depth<<20 | medium<<8) + i of the path's generator (DESIGN.md "RNG"), so the
// outcome does not depend on when the tree search reaches the medium, nor on how its steps interleave with other lanes'.
// Starts boundary search `phase` (1: over (-inf, inf); 2: over [t1 + 0.0001, inf), constant_medium.rs:39-49).
DEV void medium_search_begin(const DevScene& S, Trav& t, uint32_t phase) {
  const DMedium M = ldg_rec(S.mediums + t.m.midx);
  const float inf = __builtin_huge_valf();
  Trav b;
  trav_begin<false>(b, t.o, t.d, phase == 1u ? -inf : t.m.t1 + 0.0001f, inf, M.boundary, M.bxmin, M.bxmax, M.bymin, M.bymax, M.bzmin, M.bzmax, t.sp);
  t.m.phase = phase;
  t.m.cur = b.cur; t.m.sp = b.sp; t.m.tmin = b.tmin; t.m.t = b.h.t; t.m.ref = b.h.ref; t.m.dfs = b.h.dfs; t.m.inv = b.inv;
}
// The world search of `t` has reached medium `midx` (a primitive of its tree): the first boundary search starts.
DEV void medium_begin(const DevScene& S, Trav& t, uint32_t midx) {
  t.m.midx = midx;
  medium_search_begin(S, t, 1u);
}
// One turn of a lane inside a medium test: one step of the running boundary search; when that search is over, the next one
// starts or the medium's hit - if there is one - is offered to the world search.
template <bool COUNT>
DEV void medium_step(const DevScene& S, Trav& t, const Stack& st, const Rng& rng, uint32_t depth, Counters& cnt) {
  if (t.m.cur != REF_DONE) {
    Trav b;
    b.o = t.o; b.d = t.d; b.inv = t.m.inv; b.tmin = t.m.tmin; b.cur = t.m.cur; b.sp = t.m.sp; b.sp_base = t.sp;
    b.h.t = t.m.t; b.h.ref = t.m.ref; b.h.dfs = t.m.dfs; b.h.u = b.h.v = 0.0f;
    b.bt = 0.0f; b.bdfs = 0u;
    boundary_step<COUNT>(S, b, st, cnt);
    t.m.cur = b.cur; t.m.sp = b.sp; t.m.t = b.h.t; t.m.ref = b.h.ref; t.m.dfs = b.h.dfs;
  }
  if (t.m.cur != REF_DONE) return;
  const bool found = SOL_REF_KIND(t.m.ref) != SOL_REF_NONE;
  if (found && t.m.phase == 1u) {  // rec1 (constant_medium.rs:39-41): on to rec2
    t.m.t1 = t.m.t;
    medium_search_begin(S, t, 2u);
    return;
  }
  if (found) {  // rec2 (constant_medium.rs:43-49): the segment inside the boundary, the free path (:51-66)
    const uint32_t midx = t.m.midx;
    float t1 = fmaxf(t.m.t1, t.tmin);
    const float t2 = fminf(t.m.t, t.h.t);  // (the world search rested meanwhile: its best t is the one the test began with)
    if (t1 < t2) {
      t1 = fmaxf(t1, 0.0f);
      const float r_length = len3(t.d);
      const float distance_inside = (t2 - t1) * r_length;
      const uint32_t c = 0x40000000u + (((depth & 0x3FFu) << 20) | ((midx & 0xFFFu) << 8));
      const float nid = __uint_as_float(ldg_u32(reinterpret_cast<const uint32_t*>(&S.mediums[midx].nid)));
      const float hit_distance = nid * log_r(u32_to_unit(rng_bits(rng, c)));
      if (!(hit_distance > distance_inside)) {
        const float tt = t1 + hit_distance / r_length;
        const uint32_t dfs = ldg_u32(&S.mediums[midx].dfs);
        if (better(tt, dfs, t.h)) { t.h.t = tt; t.h.ref = SOL_MAKE_REF(SOL_REF_MEDIUM, midx); t.h.dfs = dfs; }
      }
    }
  }
  t.m.phase = 0u;  // the world search goes on
  if ((t.pg >> 24) == 0u && (t.g0 >> 24) == 0u && t.sp == t.sp_base) t.cur = REF_DONE;
}

// One step of a 7-wide world search for a whole WAVE (every lane of the wave that is still in the kernel calls it; `act`: the lane
// has a running search). Part 1 - a lane without pending primitives takes the nearest child of its node group (wide_visit): a new
// node group and a new primitive group. Part 2 - a lane with pending primitives tests ONE of them. A lane so advances by up to two
// visits per step, while a wave whose lanes are spread over nodes and primitives pays for both parts anyway. The votes on the
// step's shape (does any lane hold primitives? are they postponed?) are taken by the whole wave once, not inside the divergent
// region of the searching lanes (MI355X, 64 spp, ms: C3 69.57 -> 68.85, C2 44.0 -> 43.45, C1 10.37 -> 10.30).
template <bool COUNT, bool MEDIUM, bool STRICT = false, bool PACK = false>
DEV void trav_step_wave(const DevScene& S, Trav& t, bool act, const Stack& st, const Rng& rng, uint32_t depth, Counters& cnt) {
  // (MEDIUM) a lane inside a medium test rests its world search: no node visit, no primitive of its group, until the test is over
  const bool world = !MEDIUM || t.m.phase == 0u;
  if (act) {
    phase_tick<COUNT>(cnt, 0);
    if (world && (t.pg >> 24) == 0u) wide_visit<COUNT, PACK>(t, st, cnt);
  }
  const bool has_prim = act && world && (t.pg >> 24) != 0u;
  const bool has_inner = act && world && !has_prim && ((t.g0 >> 24) != 0u || t.sp != t.sp_base);
  if (act && world && !has_prim && !has_inner) t.cur = REF_DONE;
  const unsigned long long prim_m = sol_ballot(has_prim);
  bool prims = true;  // does the primitive part run in this turn?
  if (!MEDIUM) {
    if (prim_m == 0ull) return;
#if SOL_PRIM_MIN > 1
    // Postponed primitive tests: while fewer than SOL_PRIM_MIN lanes hold primitives and some lane has an inner node to visit
    // next turn, the holders wait (a later primitive part then runs with more lanes enabled). Results do not depend on the order
    // of the tests.
    if (sol_ballot(has_inner) != 0ull && (int)__popcll(prim_m) < SOL_PRIM_MIN) return;
#endif
  } else {  // (the same votes, but part 3 below still has to run)
    prims = prim_m != 0ull;
#if SOL_PRIM_MIN > 1
    if (prims && sol_ballot(has_inner) != 0ull && (int)__popcll(prim_m) < SOL_PRIM_MIN) prims = false;
#endif
  }
  const uint32_t lkind = t.g1 >> 29;  // (no group was popped since this node's test: a lane with pending primitives skips part 1)
  if (has_prim && prims) {
    const uint32_t slot = (uint32_t)__builtin_ctz(t.pg >> 24);
    t.pg &= ~(1u << (24u + slot));
    uint32_t idx = (t.pg & SOL_WIDE_MAX_INDEX) + __popc(__builtin_amdgcn_ubfe(t.g1, 22u, slot));  // lmask bits below `slot`
    // Triangle leaves go straight to their test: through prim_test's chain (leaf kind -> reference kind -> compare tree) every
    // test had nine more vector instructions in front of it (C3 74.8 -> 73.3 ms per 64 spp); the other kinds keep the chain.
    if (lkind == SOL_LEAF_TRIANGLES) {
      triangle_prim_test<COUNT, STRICT ? 1 : 0>(t, st, idx, cnt);
    } else {
      uint32_t kind = lkind == SOL_LEAF_SPHERES ? SOL_REF_SPHERE : SOL_REF_QUAD;
      if (lkind == SOL_LEAF_REFS) {  // mixed node: the reference is listed
        const uint32_t r = ldg_u32(S.leaf_refs + idx);
        kind = SOL_REF_KIND(r);
        idx = SOL_REF_INDEX(r);
      }
      if (MEDIUM && kind == SOL_REF_MEDIUM) medium_begin(S, t, idx);
      else prim_test<COUNT, STRICT ? 1 : 0>(S, t, st, kind, idx, cnt);
    }
    if ((t.pg >> 24) == 0u && (t.g0 >> 24) == 0u && t.sp == t.sp_base && !(MEDIUM && t.m.phase != 0u)) t.cur = REF_DONE;
  }
  if (MEDIUM) {  // part 3: one boundary step for every lane inside a medium test (those that began it in this turn included)
    const bool in_medium = act && t.m.phase != 0u;
    if (sol_ballot(in_medium) != 0ull && in_medium) medium_step<COUNT>(S, t, st, rng, depth, cnt);
  }
}

// The same step for callers outside the product kernel's loop (the diagnostic path kernel, the A/B wavefront kernels): the lanes
// that call it are the wave as far as its votes are concerned.
template <bool COUNT, bool MEDIUM, bool STRICT = false>
DEV void trav_step(const DevScene& S, Trav& t, const Stack& st, const Rng& rng, uint32_t depth, Counters& cnt) {
  trav_step_wave<COUNT, MEDIUM, STRICT>(S, t, true, st, rng, depth, cnt);
}

// Scenes with needle triangles: is the closest hit of a finished search acceptable? A triangle hit must pass the consistency rule; if
// it does not, `t` is set up to search the same ray again for what lies BEHIND the refused hit (false). Spheres, quads and media pass.
DEV bool trav_accept_or_restart(const DevScene& S, Trav& t, const Stack& st) {
  if (SOL_REF_KIND(t.h.ref) != SOL_REF_TRIANGLE) return true;
  const DTri T = ldg_rec(st.tris + SOL_REF_INDEX(t.h.ref));
  // (the shading record of the same hit is wanted next, behind this check: its line is asked for now, so that the two fetches'
  // latencies overlap instead of following each other)
  const uint32_t warm = ldg_u32(reinterpret_cast<const uint32_t*>(S.tri_shade + SOL_REF_INDEX(t.h.ref)));
  const bool ok = tri_hit_consistent(T, t.o, t.d, t.h.t, t.h.u, t.h.v, S.tri_delta);
  asm volatile("" ::"v"(warm));
  if (ok) return true;
  const float bt = t.h.t;
  const uint32_t bdfs = t.h.dfs;
  // (the lower end of the interval stays what it was: RAY_MIN, or the bound of rule 8 - sol_path.h, sol_self_hit)
  trav_begin<true>(t, t.o, t.d, t.tmin, __builtin_huge_valf(), S.wroot, S.rxmin, S.rxmax, S.rymin, S.rymax, S.rzmin, S.rzmax, 0);
  t.bt = bt;
  t.bdfs = bdfs;
  return false;
}

// Run-to-completion form of a world search (strict: the scene has needle triangles - the run-time form of the kernels' STRICT).
template <bool COUNT, bool MEDIUM>
DEV void closest_hit(const DevScene& S, f3 o, f3 d, float tmin, float tmax, Hit& h, const Stack& st, int sp_base, const Rng& rng,
                     uint32_t depth, Counters& cnt, uint32_t from = 0u) {
  Trav t;
  trav_begin<true>(t, o, d, tmin, tmax, S.wroot, S.rxmin, S.rxmax, S.rymin, S.rymax, S.rzmin, S.rzmax, sp_base);
  for (int guard = 0; guard < 64; ++guard) {
    while (t.cur != REF_DONE) trav_step<COUNT, MEDIUM, true>(S, t, st, rng, depth, cnt);
    if (from != 0u && (0x80000000u | t.h.dfs) == from && SOL_REF_KIND(t.h.ref) != SOL_REF_NONE) {  // (fp32 rule 8, sol_path.h sol_self_hit: searched again behind that hit)
      const float bt = t.bt;
      const uint32_t bdfs = t.bdfs;
      trav_begin<true>(t, o, d, __uint_as_float(__float_as_uint(t.h.t) + 1u), tmax, S.wroot, S.rxmin, S.rxmax, S.rymin, S.rymax, S.rzmin, S.rzmax, sp_base);
      t.bt = bt; t.bdfs = bdfs;  // (a bound the needle rule had set stays)
      continue;
    }
    if (!(S.tri_delta > 0.0f) || trav_accept_or_restart(S, t, st)) break;
  }
  h = t.h;
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
