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

// The world (searches with t >= 0.001) is walked through the 7-wide quantised tree; -DSOL_WORLD_BINARY=true builds the
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
  uint32_t primary_hits;  // samples whose camera ray hit something
  uint32_t path_len[6];   // samples by the number of rays of their path: 1, 2, 3-4, 5-8, 9-16, 17 and more
};

// Lane-utilisation instrumentation (COUNT builds only): every active lane counts itself, the first active lane of the
// wave counts the 64 slots of this execution.
template <bool COUNT>
DEV void phase_tick(Counters& cnt, int ph) {
#ifdef SOL_PROBE_STEP
  return;  // (probe build: the six counters describe the two parts of a search step instead, see trav_step)
#endif
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
  st.wide_emin = e + (SOL_CLAMP_SLABS ? 24u : 0u);  // (the clamped node test reads plane bytes as the halves q * 2^-24)
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
struct Trav {
  f3 o, d, inv;
  float tmin;
  uint32_t cur;
  int sp, sp_base;
  uint32_t g0, g1, pg, oct;  // (7-wide searches only; oct = the ray's octant, sign bits of the direction as x<<2 | y<<1 | z)
  Hit h;
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
  // A 7-wide search evaluates its slabs in t-space with the inverse direction clamped to +-1e30 (wide_node_test: an exactly zero
  // direction component becomes the containment test it should be instead of NaN); clamped here, once per ray, not once per node
  if (WIDE) t.inv = mk3(__builtin_amdgcn_fmed3f(t.inv.x, -1e30f, 1e30f), __builtin_amdgcn_fmed3f(t.inv.y, -1e30f, 1e30f), __builtin_amdgcn_fmed3f(t.inv.z, -1e30f, 1e30f));
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
// v_min_f32 as it is: fminf on a value the compiler cannot prove canonical (cull_t, built from bits) costs a v_max x, x first
DEV float sol_min_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#if SOL_CLAMP_SLABS
// The slab parameters are evaluated in units of the cull distance, t' = t / min(best t, 1e30): the search interval is then [0, 1]
// and the CLAMP output modifier of the plane FMAs does what a maximum with 0 (near planes) and a minimum with the cull distance
// (far planes) did - two of the twelve vector instructions per child. A child must be visited iff te' < tx', STRICTLY: a box wholly
// beyond the cull distance clamps to te' = tx' = 1, one wholly behind the origin to 0 = 0. (A box holding a primitive that a ray
// really hits has te < tx by the builder's margin of three pads, 2.7 times the rounding error of this evaluation - sol_tree.h -, so
// the strict comparison loses nothing; the scale is taken a millionth short of 1 / cull so that a tie at t = cull stays inside.)
// NaN (an overflowing plane product) clamps to 0 under DX10_CLAMP: no constraint for a near plane; planes do not overflow for
// |inv| <= 1e30 / t_min and node scales below 1e5.
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
#elif SOL_HALF_PLANES
// Plane bytes become floats two at a time: v_perm_b32 puts two bytes of a plane word under the fp16 exponent of 1024 - the halves
// (1024 + q0, 1024 + q1), exact - and v_fma_mix_f32 takes a half as its first factor: t = (1024 + q) * B + (A - 1024 * B). 24 + 42
// vector instructions per node instead of 42 conversions + 42 FMAs, and the kernel's time follows its vector instruction count
// (DESIGN.md 3). The shifted addend costs one more rounding, |A - 1024 B| * 2^-24: the builder's extra pad covers it (sol_tree.h).
typedef _Float16 sol_h2 __attribute__((ext_vector_type(2)));
#define SOL_H2(w, sel) __builtin_bit_cast(sol_h2, __builtin_amdgcn_perm(0x64646464u, (w), (sel)))
#define SOL_WIDE_CHILD(i, hnx, hny, hnz, hfx, hfy, hfz, e)                                                             \
  {                                                                                                                     \
    const float tnx = fmaf((float)hnx.e, bx, ax), tfx = fmaf((float)hfx.e, bx, ax);                                     \
    const float tny = fmaf((float)hny.e, by, ay), tfy = fmaf((float)hfy.e, by, ay);                                     \
    const float tnz = fmaf((float)hnz.e, bz, az), tfz = fmaf((float)hfz.e, bz, az);                                     \
    const float te = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));                                                         \
    const float tx = fminf(fminf(tfx, tfy), sol_min_raw(tfz, cull_t));                                                  \
    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(tx - te), 31u);                                              \
  }
#else
#define SOL_WIDE_CHILD(i, nxw, nyw, nzw, fxw, fyw, fzw, refv)                                                          \
  {                                                                                                                     \
    const float tnx = fmaf((float)(((nxw) >> (8 * ((i) & 3))) & 0xFFu), bx, ax), tfx = fmaf((float)(((fxw) >> (8 * ((i) & 3))) & 0xFFu), bx, ax); \
    const float tny = fmaf((float)(((nyw) >> (8 * ((i) & 3))) & 0xFFu), by, ay), tfy = fmaf((float)(((fyw) >> (8 * ((i) & 3))) & 0xFFu), by, ay); \
    const float tnz = fmaf((float)(((nzw) >> (8 * ((i) & 3))) & 0xFFu), bz, az), tfz = fmaf((float)(((fzw) >> (8 * ((i) & 3))) & 0xFFu), bz, az); \
    const float te = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));                                                         \
    const float tx = fminf(fminf(tfx, tfy), sol_min_raw(tfz, cull_t));                                                        \
    /* te <= tx as the sign of tx - te, shifted into the mask: a subtraction and one alignbit instead of compare, select and */ \
    /* or. te is in [0, inf], tx in [-inf, FLT_MAX] (cull_t is finite), so the difference is never NaN; tx = -0 counts as a  */ \
    /* miss, which it is for a search with tmin > 0. An empty slot has an inverted box: a miss.                             */ \
    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(tx - te), 31u);                                                \
  }
#endif
// The slab tests of one fetched 7-wide node (h = origin + meta, qa / qb / qc = the six plane arrays) for the ray of search `t`:
// the search's new node group and primitive group.
template <bool COUNT>
DEV void wide_node_test(const Stack& st, Trav& t, uint32_t oct, float4 h, uint4 qa, uint4 qb, uint4 qc) {
  const uint32_t wide_emin = st.wide_emin;
  const bool sx = (oct & 4u) != 0u, sy = (oct & 2u) != 0u, sz = (oct & 1u) != 0u;
  const uint32_t meta = __float_as_uint(h.w);
  const float scx = __uint_as_float(((meta & 31u) + wide_emin) << 23), scy = __uint_as_float((((meta >> 5) & 31u) + wide_emin) << 23);
  const float scz = __uint_as_float((((meta >> 10) & 31u) + wide_emin) << 23);
#ifdef SOL_NO_TCULL
  const float cull_t = 3.402823466e38f;
#else
  // t >= tmin > 0 in a world search; made finite (see SOL_WIDE_CHILD) by an unsigned minimum of the bit patterns: best t is a
  // positive float or +inf, never NaN, and one instruction where fminf takes two (canonicalise, then minimum)
  const float cull_t = __uint_as_float(min(__float_as_uint(t.h.t), 0x7F7FFFFFu));
#endif
  uint32_t miss = 0u;  // children are tested 6 .. 0, each shifting its bit in at the bottom: child i ends on bit i
  // An exactly zero direction component gives inv = inf, and A + q * B = -inf + inf = NaN for every plane: "no constraint",
  // i.e. the ray would visit every node. Clamped to +-1e30 the axis becomes the containment test it should be (origin
  // inside the slab: planes at -+huge; outside: both planes at the same huge sign -> culled).
#if SOL_CLAMP_SLABS
  // (units of the cull distance, a millionth short: see SOL_WIDE_CHILD; best t is a positive float or +inf, never NaN)
  const float rc = __builtin_amdgcn_rcpf(__builtin_amdgcn_fmed3f(t.h.t, RAY_MIN_F, 1e30f) * 1.000001f);
  (void)cull_t;
  const float ivx = t.inv.x * rc, ivy = t.inv.y * rc, ivz = t.inv.z * rc;  // (t.inv: clamped to +-1e30 by trav_begin)
#else
  const float ivx = t.inv.x, ivy = t.inv.y, ivz = t.inv.z;
#endif
  const float ax = (h.x - t.o.x) * ivx, bx = scx * ivx;
  const float ay = (h.y - t.o.y) * ivy, by = scy * ivy;
  const float az = (h.z - t.o.z) * ivz, bz = scz * ivz;
  // q words: qa = {lo_x[0..3], lo_x[4..7], lo_y[0..3], lo_y[4..7]}, qb = {lo_z.., lo_z.., hi_x.., hi_x..},
  //          qc = {hi_y.., hi_y.., hi_z.., hi_z..}; near = the plane the ray meets first on that axis
  const uint32_t nx0 = sx ? qb.z : qa.x, nx1 = sx ? qb.w : qa.y, fx0 = sx ? qa.x : qb.z, fx1 = sx ? qa.y : qb.w;
  const uint32_t ny0 = sy ? qc.x : qa.z, ny1 = sy ? qc.y : qa.w, fy0 = sy ? qa.z : qc.x, fy1 = sy ? qa.w : qc.y;
  const uint32_t nz0 = sz ? qc.z : qb.x, nz1 = sz ? qc.w : qb.y, fz0 = sz ? qb.x : qc.z, fz1 = sz ? qb.y : qc.w;
#if SOL_HALF_PLANES || SOL_CLAMP_SLABS
  {
#if !SOL_CLAMP_SLABS
    const float ax0 = ax, ay0 = ay, az0 = az;
    const float ax = fmaf(-1024.0f, bx, ax0), ay = fmaf(-1024.0f, by, ay0), az = fmaf(-1024.0f, bz, az0);  // (shadow the plain addends)
#endif
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
#else
  SOL_WIDE_CHILD(6, nx1, ny1, nz1, fx1, fy1, fz1, 0)
  SOL_WIDE_CHILD(5, nx1, ny1, nz1, fx1, fy1, fz1, 0)
  SOL_WIDE_CHILD(4, nx1, ny1, nz1, fx1, fy1, fz1, 0)
  SOL_WIDE_CHILD(3, nx0, ny0, nz0, fx0, fy0, fz0, 0)
  SOL_WIDE_CHILD(2, nx0, ny0, nz0, fx0, fy0, fz0, 0)
  SOL_WIDE_CHILD(1, nx0, ny0, nz0, fx0, fy0, fz0, 0)
  SOL_WIDE_CHILD(0, nx0, ny0, nz0, fx0, fy0, fz0, 0)
#endif
#if SOL_CLAMP_SLABS
  const uint32_t hits = miss;  // (the clamped form shifts in HIT bits)
#else
  const uint32_t hits = ~miss;
#endif
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
template <bool COUNT>
DEV void triangle_prim_test(Trav& t, const Stack& st, uint32_t idx, Counters& cnt) {
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
template <bool COUNT, bool MEDIUM>
DEV void prim_test(const DevScene& S, Trav& t, const Stack& st, uint32_t kind, uint32_t idx, const Rng& rng, uint32_t depth,
                   Counters& cnt) {
  const uint32_t ref = SOL_MAKE_REF(kind, idx);
  if (kind == SOL_REF_TRIANGLE) {
    triangle_prim_test<COUNT>(t, st, idx, cnt);
  } else if (kind == SOL_REF_SPHERE) {
    sphere_prim_test<COUNT>(S, t, idx, cnt);
  } else if (kind == SOL_REF_QUAD) {
    quad_prim_test<COUNT>(S, t, idx, cnt);
  } else if (MEDIUM && kind == SOL_REF_MEDIUM) {
    float tt;
    const uint32_t dfs = ldg_u32(&S.mediums[idx].dfs);
    if (medium_test<COUNT>(S, idx, t.o, t.d, t.tmin, t.h.t, tt, st, t.sp, rng, depth, cnt) && better(tt, dfs, t.h)) {
      t.h.t = tt; t.h.ref = ref; t.h.dfs = dfs;
    }
  }
}

// Part 1 of a step of a 7-wide search, for a lane without pending primitives: takes the nearest child of the lane's node group
// (popping a group first when its own is used up), pushes the rest of the group, fetches that node (64 bytes) and tests its seven
// child boxes: a new node group and a new primitive group.
template <bool COUNT>
DEV void wide_visit(Trav& t, const Stack& st, Counters& cnt) {
  const uint32_t oct = t.oct;
#if SOL_FETCH_PRIO
  __builtin_amdgcn_s_setprio(SOL_FETCH_PRIO % 10);  // the wave about to fetch goes first: its memory latency starts now
#endif
  uint32_t g0 = t.g0, g1 = t.g1;
  if ((g0 >> 24) == 0u) {  // (a running search without pending primitives has a group here or on the stack)
    stack_pop2(st, t.sp, g0, g1);
  }
  const uint32_t p = (uint32_t)__builtin_ctz(g0 >> 24);  // nearest: lowest bit in visit order
  const uint32_t slot = p ^ oct;
  g0 &= ~(1u << (24u + p));
  const uint32_t idx = (g0 & SOL_WIDE_MAX_INDEX) + __popc(__builtin_amdgcn_ubfe(g1, 15u, slot));  // rank among the node's inner children: imask bits below `slot`
  if ((g0 >> 24) != 0u) {  // siblings left: one stack entry for all of them
    stack_push(st, t.sp, g0);
    stack_push(st, t.sp, g1);
    if (COUNT) cnt.max_stack = max(cnt.max_stack, (uint32_t)t.sp);
  }
  const float4* wp = reinterpret_cast<const float4*>(st.wides + idx);
  const float4 h = ldg_f4(wp);
  const uint4 qa = ldg_u4(wp + 1), qb = ldg_u4(wp + 2), qc = ldg_u4(wp + 3);
#if SOL_FETCH_PRIO
  asm volatile("s_setprio %0" ::"n"(SOL_LOOP_PRIO) : "memory");  // (after the loads are issued; asm: the builtin may be moved across them)
#endif
  if (COUNT) cnt.node_visits++;
#if defined(SOL_EXP_VMEM) || defined(SOL_EXP_VALU) || defined(SOL_EXP_LDS)
  // Sensitivity probes (A/B builds only, tests/tools/variants.py): extra work per node visit that changes no result -
  // SOL_EXP_VMEM more 16-byte loads of this node (L1 hits), SOL_EXP_VALU more vector instructions, SOL_EXP_LDS more LDS stores -
  // to see which pipe the kernel's time follows.
  {
    uint32_t zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#ifdef SOL_EXP_VMEM
#pragma unroll
    for (int k = 0; k < SOL_EXP_VMEM; ++k) { const uint4 x = ldg_u4(wp + zero + (k % 4)); asm volatile("" ::"v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w)); }
#endif
#ifdef SOL_EXP_VALU
    float dummy = h.x;
#pragma unroll
    for (int k = 0; k < SOL_EXP_VALU; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(dummy));
    asm volatile("" ::"v"(dummy));
#endif
#ifdef SOL_EXP_LDS
#pragma unroll
    for (int k = 0; k < SOL_EXP_LDS; ++k) st.lds[(SOL_LDS_STACK - 1) * SOL_WG] = zero;  // (top level: scratch unless the stack is full)
#endif
  }
#endif
  wide_node_test<COUNT>(st, t, oct, h, qa, qb, qc);
}

// One step of a search. BINARY selects which tree this search walks: the 2-wide DNode tree (constant-medium boundaries,
// whose search interval includes negative t) or the 7-wide DWide tree (the world).
//   2-wide: visits the node or primitive t.cur, then moves to the next reference (near child, or popped from the stack).
//   7-wide: part 1 - a lane without pending primitives takes the nearest child of its node group (popping a group first when
//   its own is used up), pushes the rest of the group, fetches that node (64 bytes) and tests its seven child boxes: a new node
//   group and a new primitive group. Part 2 - a lane with pending primitives tests ONE of them. A lane so advances by up to
//   two visits per step, while a wave whose lanes are spread over nodes and primitives pays for both parts anyway.
template <bool COUNT, bool MEDIUM, bool BINARY>
DEV void trav_step(const DevScene& S, Trav& t, const Stack& st, const Rng& rng, uint32_t depth, Counters& cnt) {
  phase_tick<COUNT>(cnt, 0);
  if (!BINARY) {
#ifdef SOL_PROBE_STEP
    // probe build (tests/tools/variants.py, perf_quick --phases): [0] lanes in node parts, [1] 64 per node part executed by the
    // wave, [2] / [3] the same for primitive parts, [4] lanes holding primitives while a node part runs, [5] of those, the
    // lanes whose primitive part is then postponed
    if (COUNT) {
      const bool node_lane = (t.pg >> 24) == 0u;
      const unsigned long long nm = sol_ballot(node_lane);
      if (nm != 0ull) {
        if (node_lane) cnt.phase[0]++; else cnt.phase[4]++;
        if ((int)__lane_id() == __ffsll((long long)sol_ballot(true)) - 1) cnt.phase[1] += 64u;
      }
    }
#endif
    if ((t.pg >> 24) == 0u) wide_visit<COUNT>(t, st, cnt);
    // status for the callers' loops, and for the postponing rule below
    const bool has_prim = (t.pg >> 24) != 0u;
    const bool has_inner = !has_prim && ((t.g0 >> 24) != 0u || t.sp != t.sp_base);
#if SOL_PRIM_MIN > 1
    // Postponed primitive tests: the lanes holding primitives wait while fewer than SOL_PRIM_MIN of the wave's lanes do and
    // some lane still has an inner node to visit next turn; the primitive part then runs with more lanes enabled. Results do
    // not depend on the order of the tests.
    const unsigned long long inner_m = sol_ballot(has_inner), prim_m = sol_ballot(has_prim);
    if (!has_prim && !has_inner) { t.cur = REF_DONE; return; }
    if (!has_prim) return;
#ifdef SOL_PROBE_STEP
    if (COUNT && inner_m != 0ull && (int)__popcll(prim_m) < SOL_PRIM_MIN) cnt.phase[5]++;
#endif
    if (inner_m != 0ull && (int)__popcll(prim_m) < SOL_PRIM_MIN) return;
#else
    if (!has_prim && !has_inner) { t.cur = REF_DONE; return; }
    if (!has_prim) return;
#endif
#ifdef SOL_PROBE_STEP
    if (COUNT) {
      cnt.phase[2]++;
      if ((int)__lane_id() == __ffsll((long long)sol_ballot(true)) - 1) cnt.phase[3] += 64u;
    }
#endif
    const uint32_t slot = (uint32_t)__builtin_ctz(t.pg >> 24);
    t.pg &= ~(1u << (24u + slot));
    const uint32_t lkind = t.g1 >> 29;  // (no group was popped since this node's test: a lane with pending primitives skips part 1)
    uint32_t idx = (t.pg & SOL_WIDE_MAX_INDEX) + __popc(__builtin_amdgcn_ubfe(t.g1, 22u, slot));  // lmask bits below `slot`
    // Triangle leaves go straight to their test: through prim_test's chain (leaf kind -> reference kind -> compare tree) every
    // test had nine more vector instructions in front of it. MI355X, 64 spp, ms: C3 74.8 -> 73.3; a direct path for EVERY leaf
    // kind was no better (C3 73.8, C1 10.75 against 10.57 / 10.67): the other kinds keep the chain.
#if SOL_LEAF_KIND_DISPATCH == 1
    if (lkind == SOL_LEAF_TRIANGLES) {
      triangle_prim_test<COUNT>(t, st, idx, cnt);
    } else
#endif
    {
      uint32_t kind = lkind == SOL_LEAF_TRIANGLES ? SOL_REF_TRIANGLE : lkind == SOL_LEAF_SPHERES ? SOL_REF_SPHERE : SOL_REF_QUAD;
      if (lkind == SOL_LEAF_REFS) {  // mixed node: the reference is listed
        const uint32_t r = ldg_u32(S.leaf_refs + idx);
        kind = SOL_REF_KIND(r);
        idx = SOL_REF_INDEX(r);
      }
      prim_test<COUNT, MEDIUM>(S, t, st, kind, idx, rng, depth, cnt);
    }
    if ((t.pg >> 24) == 0u && (t.g0 >> 24) == 0u && t.sp == t.sp_base) t.cur = REF_DONE;
    return;
  }
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
  if (cur == REF_DONE || kind == SOL_REF_NODE) { t.cur = cur; return; }
  prim_test<COUNT, MEDIUM>(S, t, st, kind, SOL_REF_INDEX(cur), rng, depth, cnt);
  t.cur = (t.sp == t.sp_base) ? REF_DONE : stack_pop(st, t.sp);
}

// ---- Intra-wave donation of pending node groups (-DSOL_DONATE=1) -------------------------------------------------------------
// In a node part of the search loop 33 of 64 lanes test a node and 24 have FINISHED their search and wait for the wave to leave
// for the service block (MI355X, C3). With donation such a lane adopts the top stack entry - one node group: the
// untested siblings of a node - of a lane that is still searching, walks that sub-tree with the donor's ray (gathered through
// ds_bpermute) and the donor's best t of that moment as its cull distance, and hands its closest hit back; the donor merges it by
// the (t, dfs) rule of `better`. The closest hit does not depend on who searches which sub-tree or with which cull distance (every
// candidate is either found or beaten by something found: DESIGN.md 4, "tree independence"), so frames stay bit-identical.
//   status of a 7-wide search in t.cur:  0 running | CUR_LENT own search with a group out (running or exhausted: then the lane
//   waits) | CUR_HELP + donor lane: walking a stolen group | CUR_HELP_FIN + donor lane: that walk is over, result to deliver |
//   REF_DONE.
// No extra LDS: the helper's own finished search (ray, hit: 10 dwords - the service block needs them) is parked in the top ten
// entries of ITS OWN stack column, which is empty (its search is over; the stolen sub-tree needs at most bound - 2 entries); the
// mailbox (valid, t, ref, dfs, u, v) is the top six entries of the DONOR's column. The host enables donation only for trees whose
// stack bound leaves that room (RenderParams::donate), never for counted launches (the counters would depend on scheduling) and
// never for scenes with constant media (their nested boundary searches use the column above t.sp).
#define CUR_LENT 1u
#define CUR_HELP 0x100u
#define CUR_HELP_FIN 0x200u
#define SOL_PARK_AT (SOL_LDS_STACK - 10)  // first entry of a helper's parked search in its own column
#define SOL_MAIL_AT (SOL_LDS_STACK - 6)   // first entry of the mailbox in a donor's column
#define SOL_DONATE_BOUND (SOL_LDS_STACK - 8)  // largest stack bound of a tree that may donate: mailbox above the donor's own entries, and
                                              // a stolen group (level >= 1: bound - 2) below the helper's parked search
DEV void trav_out_of_work(Trav& t) {
#if SOL_DONATE
  if (t.cur & CUR_HELP) t.cur ^= (CUR_HELP | CUR_HELP_FIN);  // -> CUR_HELP_FIN + donor: delivered by trav_donate in this turn
  else if (t.cur != CUR_LENT) t.cur = REF_DONE;              // (CUR_LENT: wait for the helper)
#else
  t.cur = REF_DONE;
#endif
}
#if SOL_DONATE
#ifndef SOL_DONATE_MIN
#define SOL_DONATE_MIN 8   // lanes that must be waiting with a finished search before groups are handed out
#endif
#ifndef SOL_DONATE_EVERY
#define SOL_DONATE_EVERY 2  // the hand-out is considered every 2^n-th turn of the search loop
#endif
// Called by every lane of the wave after a step. pair: 64 dwords of LDS of this wave (donor lane of the k-th pair).
DEV void trav_donate(Trav& t, const Stack& st, volatile lds_u32* pair, uint32_t turn) {
  const uint32_t lane = __lane_id();
  volatile lds_u32* const mine = (volatile lds_u32*)st.lds;
  // ---- delivery: helpers whose stolen group is exhausted write their hit into the donor's mailbox and take their own finished
  // search back; lanes with a group out look into their mailbox
  const bool fin = t.cur != REF_DONE && (t.cur & CUR_HELP_FIN) != 0u;
  if (sol_ballot(fin) != 0ull) {
    if (fin) {
      volatile lds_u32* const box = mine - lane + (t.cur & 63u);
      box[(SOL_MAIL_AT + 1) * SOL_WG] = __float_as_uint(t.h.t);
      box[(SOL_MAIL_AT + 2) * SOL_WG] = t.h.ref;
      box[(SOL_MAIL_AT + 3) * SOL_WG] = t.h.dfs;
      box[(SOL_MAIL_AT + 4) * SOL_WG] = __float_as_uint(t.h.u);
      box[(SOL_MAIL_AT + 5) * SOL_WG] = __float_as_uint(t.h.v);
      box[(SOL_MAIL_AT + 0) * SOL_WG] = 1u;
      t.o.x = __uint_as_float(mine[(SOL_PARK_AT + 0) * SOL_WG]); t.o.y = __uint_as_float(mine[(SOL_PARK_AT + 1) * SOL_WG]);
      t.o.z = __uint_as_float(mine[(SOL_PARK_AT + 2) * SOL_WG]); t.d.x = __uint_as_float(mine[(SOL_PARK_AT + 3) * SOL_WG]);
      t.d.y = __uint_as_float(mine[(SOL_PARK_AT + 4) * SOL_WG]); t.d.z = __uint_as_float(mine[(SOL_PARK_AT + 5) * SOL_WG]);
      t.h.t = __uint_as_float(mine[(SOL_PARK_AT + 6) * SOL_WG]); t.h.ref = mine[(SOL_PARK_AT + 7) * SOL_WG];
      t.h.u = __uint_as_float(mine[(SOL_PARK_AT + 8) * SOL_WG]); t.h.v = __uint_as_float(mine[(SOL_PARK_AT + 9) * SOL_WG]);
      t.g0 = 0u; t.pg = 0u; t.sp = t.sp_base = 0;
      t.cur = REF_DONE;
    }
    if (t.cur == CUR_LENT && mine[SOL_MAIL_AT * SOL_WG] != 0u) {
      const float tt = __uint_as_float(mine[(SOL_MAIL_AT + 1) * SOL_WG]);
      const uint32_t ref = mine[(SOL_MAIL_AT + 2) * SOL_WG], dfs = mine[(SOL_MAIL_AT + 3) * SOL_WG];
      const float u = __uint_as_float(mine[(SOL_MAIL_AT + 4) * SOL_WG]), v = __uint_as_float(mine[(SOL_MAIL_AT + 5) * SOL_WG]);
      if (SOL_REF_KIND(ref) != SOL_REF_NONE && better(tt, dfs, t.h)) { t.h.t = tt; t.h.ref = ref; t.h.dfs = dfs; t.h.u = u; t.h.v = v; }
      t.cur = 0u;
      if ((t.pg >> 24) == 0u && (t.g0 >> 24) == 0u && t.sp == t.sp_base) t.cur = REF_DONE;
    }
  }
  // ---- hand-out: the k-th waiting lane takes the top stack entry of the k-th lane that has one to give
  if ((turn & ((1u << SOL_DONATE_EVERY) - 1u)) != 0u) return;
  const bool idle = t.cur == REF_DONE;
  const unsigned long long idle_m = sol_ballot(idle);
  if ((int)__popcll(idle_m) < SOL_DONATE_MIN) return;
  const bool giver = t.cur == 0u && t.sp - t.sp_base >= 2;
  const unsigned long long giver_m = sol_ballot(giver);
  if (giver_m == 0ull) return;
  const uint32_t n_pairs = min((uint32_t)__popcll(idle_m), (uint32_t)__popcll(giver_m));
  const uint32_t rank_i = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
  const uint32_t rank_g = __builtin_amdgcn_mbcnt_hi((uint32_t)(giver_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)giver_m, 0u));
  const bool gives = giver && rank_g < n_pairs, takes = idle && rank_i < n_pairs;
  if (gives) {
    mine[SOL_MAIL_AT * SOL_WG] = 0u;
#if SOL_DONATE_BOTTOM  // the OLDEST entry: the far siblings of a shallow node - a large sub-tree, but the one a near hit would have culled
    pair[rank_g] = lane | ((uint32_t)(t.sp_base + 2) << 8);
    t.sp_base += 2;
#else                  // the newest entry: what the donor would have searched next
    pair[rank_g] = lane | ((uint32_t)t.sp << 8);
    t.sp -= 2;  // (the entry stays where it is until the helper has read it: nothing is pushed before the next step)
#endif
    t.cur = CUR_LENT;
  }
  const uint32_t e = takes ? pair[rank_i] : lane;
  const uint32_t from = e & 63u, from4 = from << 2;
  // (every lane executes the gathers: a bpermute reads registers of enabled lanes only)
  const uint32_t ox = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from4, (int)__float_as_uint(t.o.x)), oy = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from4, (int)__float_as_uint(t.o.y));
  const uint32_t oz = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from4, (int)__float_as_uint(t.o.z)), dx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from4, (int)__float_as_uint(t.d.x));
  const uint32_t dy = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from4, (int)__float_as_uint(t.d.y)), dz = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from4, (int)__float_as_uint(t.d.z));
  const uint32_t ix = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from4, (int)__float_as_uint(t.inv.x)), iy = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from4, (int)__float_as_uint(t.inv.y));
  const uint32_t iz = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from4, (int)__float_as_uint(t.inv.z)), bt = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from4, (int)__float_as_uint(t.h.t));
  const uint32_t oc = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from4, (int)t.oct);
  if (takes) {
    mine[(SOL_PARK_AT + 0) * SOL_WG] = __float_as_uint(t.o.x); mine[(SOL_PARK_AT + 1) * SOL_WG] = __float_as_uint(t.o.y);
    mine[(SOL_PARK_AT + 2) * SOL_WG] = __float_as_uint(t.o.z); mine[(SOL_PARK_AT + 3) * SOL_WG] = __float_as_uint(t.d.x);
    mine[(SOL_PARK_AT + 4) * SOL_WG] = __float_as_uint(t.d.y); mine[(SOL_PARK_AT + 5) * SOL_WG] = __float_as_uint(t.d.z);
    mine[(SOL_PARK_AT + 6) * SOL_WG] = __float_as_uint(t.h.t); mine[(SOL_PARK_AT + 7) * SOL_WG] = t.h.ref;
    mine[(SOL_PARK_AT + 8) * SOL_WG] = __float_as_uint(t.h.u); mine[(SOL_PARK_AT + 9) * SOL_WG] = __float_as_uint(t.h.v);
    volatile lds_u32* const col = mine - lane + from;
    const uint32_t at = (e >> 8) - 2u;
    t.g0 = col[at * SOL_WG];
    t.g1 = col[(at + 1u) * SOL_WG];
    t.pg = 0u;
    t.sp = t.sp_base = 0;
    t.o = mk3(__uint_as_float(ox), __uint_as_float(oy), __uint_as_float(oz));
    t.d = mk3(__uint_as_float(dx), __uint_as_float(dy), __uint_as_float(dz));
    t.inv = mk3(__uint_as_float(ix), __uint_as_float(iy), __uint_as_float(iz));
    t.oct = oc;
    t.h.t = __uint_as_float(bt);
    t.h.ref = SOL_MAKE_REF(SOL_REF_NONE, 0);
    t.h.dfs = 0u;
    t.cur = CUR_HELP | from;
  }
}
#endif  // SOL_DONATE

// One step of a 7-wide world search for a whole WAVE (every lane of the wave calls it; `act`: the lane has a running search) - the
// product kernel's form of trav_step: the same two parts, but the votes on the step's shape (does any lane hold primitives? are
// they postponed?) are taken by the whole wave once instead of inside the divergent region of the searching lanes
// (MI355X, 64 spp, ms: C3 69.57 -> 68.85, C2 44.0 -> 43.45, C1 10.37 -> 10.30; profiles/r03_coop_triangles_ab.txt).
//
// -DSOL_COOP_TRIANGLES=1 (an A/B build; MEASURED SLOWER, kept as the record of the experiment): the primitive part COOPERATIVELY.
// A lane with pending primitives tests one of them per step and the primitive part runs with a quarter of the wave's lanes
// (C3: 16.0 of 64), a fifth of all vector instructions of a launch at that occupancy. In the cooperative form all
// pending TRIANGLE tests of the wave - every hit leaf slot of every lane's primitive group - are dealt out over all 64 lanes,
// finished and node-only lanes included: the owners list their tests in a 64-entry LDS queue (exclusive prefix of their counts by
// three ballots), the r-th enabled lane takes entry r, gathers the owner's ray and best t (ds_bpermute: the LDS crossbar, no
// memory), tests the triangle, and the owners collect the results by the (t, dfs) rule of `better` - a total order, so neither
// who tests nor in which order changes the hit (frames bit-identical). It loses: listing and collecting are loops over the
// largest primitive group of the wave (~12 + ~25 vector instructions per turn) around a 60-instruction test, and a primitive part
// deals out 20 - 30 tests, not 64: C3 69.6 -> 74.1 ms with the postponing rule at 8 lanes, 72.0 at 16, 75.0 at 24, 80.5 at 4.
template <bool COUNT, bool MEDIUM>
DEV void trav_step_wave(const DevScene& S, Trav& t, bool act, const Stack& st, volatile lds_u32* queue, const Rng& rng, uint32_t depth,
                        Counters& cnt) {
  if (act) {
    phase_tick<COUNT>(cnt, 0);
#if SOL_DONATE
    // (a lane whose own search is exhausted while a group of it is out with a helper stays in the loop without work)
    if ((t.pg >> 24) == 0u && ((t.g0 >> 24) != 0u || t.sp != t.sp_base)) wide_visit<COUNT>(t, st, cnt);
#else
    if ((t.pg >> 24) == 0u) wide_visit<COUNT>(t, st, cnt);
#endif
  }
  const bool has_prim = act && (t.pg >> 24) != 0u;
  const bool has_inner = act && !has_prim && ((t.g0 >> 24) != 0u || t.sp != t.sp_base);
  if (act && !has_prim && !has_inner) trav_out_of_work(t);
  const unsigned long long prim_m = sol_ballot(has_prim);
  if (prim_m == 0ull) return;
#if SOL_PRIM_MIN > 1
  // Postponed primitive tests: while fewer than SOL_PRIM_MIN lanes hold primitives and some lane has an inner node to visit next
  // turn, the holders wait (a later primitive part then has more tests to deal out)
  if (sol_ballot(has_inner) != 0ull && (int)__popcll(prim_m) < SOL_PRIM_MIN) return;
#endif
  const uint32_t lkind = t.g1 >> 29;
  if (has_prim && (SOL_COOP_TRIANGLES == 0 || lkind != SOL_LEAF_TRIANGLES)) {  // part 2 of trav_step: ONE of the lane's own primitives
    const uint32_t slot = (uint32_t)__builtin_ctz(t.pg >> 24);
    t.pg &= ~(1u << (24u + slot));
    uint32_t idx = (t.pg & SOL_WIDE_MAX_INDEX) + __popc(__builtin_amdgcn_ubfe(t.g1, 22u, slot));  // lmask bits below `slot`
    if (lkind == SOL_LEAF_TRIANGLES) {  // (straight to their test, not through prim_test's chain: nine vector instructions less)
      triangle_prim_test<COUNT>(t, st, idx, cnt);
    } else {
      uint32_t kind = lkind == SOL_LEAF_SPHERES ? SOL_REF_SPHERE : SOL_REF_QUAD;
      if (lkind == SOL_LEAF_REFS) {  // mixed node: the reference is listed
        const uint32_t r = ldg_u32(S.leaf_refs + idx);
        kind = SOL_REF_KIND(r);
        idx = SOL_REF_INDEX(r);
      }
      prim_test<COUNT, MEDIUM>(S, t, st, kind, idx, rng, depth, cnt);
    }
    if ((t.pg >> 24) == 0u && (t.g0 >> 24) == 0u && t.sp == t.sp_base) trav_out_of_work(t);
  }
#if SOL_COOP_TRIANGLES
  const bool tri_owner = has_prim && lkind == SOL_LEAF_TRIANGLES;
  if (sol_ballot(tri_owner) == 0ull) return;
  // ---- the wave's pending triangle tests, dealt out over all its lanes ----
  const uint32_t lane = __lane_id();
  const uint32_t pend = tri_owner ? (t.pg >> 24) : 0u;
  const uint32_t n = (uint32_t)__popc(pend);
  const unsigned long long m0 = sol_ballot((n & 1u) != 0u), m1 = sol_ballot((n & 2u) != 0u), m2 = sol_ballot((n & 4u) != 0u);
  const uint32_t first = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u)) +
                         2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u)) +
                         4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
  const uint32_t total = (uint32_t)__popcll(m0) + 2u * (uint32_t)__popcll(m1) + 4u * (uint32_t)__popcll(m2);
  // Entry r of the queue goes to the r-th ENABLED lane of the wave (at the end of a launch lanes without work have left the kernel:
  // nothing may be dealt to them, and a bpermute reads registers of enabled lanes only)
  const unsigned long long live = sol_ballot(true);
  const uint32_t n_live = (uint32_t)__popcll(live);
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
  // owners list their tests (triangle index | owner lane << 24) at queue[first ..]; what does not fit stays pending
  uint32_t listed = 0u;  // the slots this lane got into the queue
  {
    uint32_t rest = pend, pos = first;
    while (rest != 0u && pos < n_live) {
      const uint32_t slot = (uint32_t)__builtin_ctz(rest);
      rest &= rest - 1u;
      listed |= 1u << slot;
      queue[pos++] = ((t.pg & SOL_WIDE_MAX_INDEX) + __popc(__builtin_amdgcn_ubfe(t.g1, 22u, slot))) | (lane << 24);
    }
  }
  const uint32_t n_tests = min(total, n_live);
  const bool tester = rank < n_tests;
  const uint32_t e = tester ? queue[rank] : (lane << 24);
  if (tester) queue[rank] = lane;  // (its own entry, read above: where the owner finds the result)
  const uint32_t owner4 = (e >> 24) << 2;  // byte address of the owner lane for ds_bpermute (all lanes execute the gathers: a
                                           // bpermute reads registers of ENABLED lanes only)
  f3 o, d;
  o.x = __uint_as_float(__builtin_amdgcn_ds_bpermute((int)owner4, (int)__float_as_uint(t.o.x)));
  o.y = __uint_as_float(__builtin_amdgcn_ds_bpermute((int)owner4, (int)__float_as_uint(t.o.y)));
  o.z = __uint_as_float(__builtin_amdgcn_ds_bpermute((int)owner4, (int)__float_as_uint(t.o.z)));
  d.x = __uint_as_float(__builtin_amdgcn_ds_bpermute((int)owner4, (int)__float_as_uint(t.d.x)));
  d.y = __uint_as_float(__builtin_amdgcn_ds_bpermute((int)owner4, (int)__float_as_uint(t.d.y)));
  d.z = __uint_as_float(__builtin_amdgcn_ds_bpermute((int)owner4, (int)__float_as_uint(t.d.z)));
  const float tmax = __uint_as_float(__builtin_amdgcn_ds_bpermute((int)owner4, (int)__float_as_uint(t.h.t)));
  float rt = __builtin_huge_valf(), ru = 0.0f, rv = 0.0f;  // this lane's result: t = +inf when the triangle is not hit
  uint32_t rdfs = 0u;
  const uint32_t ridx = e & SOL_WIDE_MAX_INDEX;
  if (tester) {
    const float4* tp = reinterpret_cast<const float4*>(st.tris + ridx);
#if SOL_FETCH_PRIO >= 10
    __builtin_amdgcn_s_setprio(SOL_FETCH_PRIO / 10);
#endif
    const float4 p0 = ldg_f4(tp), p1 = ldg_f4(tp + 1), p2 = ldg_f4(tp + 2);
#if SOL_FETCH_PRIO >= 10
    asm volatile("s_setprio %0" ::"n"(SOL_LOOP_PRIO) : "memory");
#endif
    DTri T;
    T.v0x = p0.x; T.v0y = p0.y; T.v0z = p0.z; T.e1x = p0.w; T.e1y = p1.x; T.e1z = p1.y; T.e2x = p1.z; T.e2y = p1.w; T.e2z = p2.x;
    rdfs = __float_as_uint(p2.y);
    if (COUNT) cnt.triangle_tests++;
    float tt, u, v;
    if (tri_test(T, o, d, RAY_MIN_F, tmax, tt, u, v)) { rt = tt; ru = u; rv = v; }  // (a world search starts at RAY_MIN_F: trav_begin)
  }
  // owners collect their results, in queue order. The loop's trip count is wave-uniform and every lane executes the gathers.
  {
    uint32_t rest = listed, pos = first;
    while (sol_ballot(rest != 0u) != 0ull) {
      const uint32_t src4 = (rest != 0u ? queue[pos] : lane) << 2;
      const float tt = __uint_as_float(__builtin_amdgcn_ds_bpermute((int)src4, (int)__float_as_uint(rt)));
      const float u = __uint_as_float(__builtin_amdgcn_ds_bpermute((int)src4, (int)__float_as_uint(ru)));
      const float v = __uint_as_float(__builtin_amdgcn_ds_bpermute((int)src4, (int)__float_as_uint(rv)));
      const uint32_t dfs = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src4, (int)rdfs);
      const uint32_t idx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src4, (int)ridx);
      if (rest != 0u) {
        rest &= rest - 1u;
        pos++;
        if (tt < __builtin_huge_valf() && better(tt, dfs, t.h)) {
          t.h.t = tt; t.h.ref = SOL_MAKE_REF(SOL_REF_TRIANGLE, idx); t.h.dfs = dfs; t.h.u = u; t.h.v = v;
        }
      }
    }
  }
  if (tri_owner) {
    t.pg &= ~(listed << 24);
    if ((t.pg >> 24) == 0u && (t.g0 >> 24) == 0u && t.sp == t.sp_base) t.cur = REF_DONE;
  }
#else
  (void)queue;
#endif
}

// Run-to-completion form.
template <bool COUNT, bool MEDIUM, bool BINARY>
DEV void closest_hit(const DevScene& S, f3 o, f3 d, float tmin, float tmax, uint32_t root, float bxmin, float bxmax,
                     float bymin, float bymax, float bzmin, float bzmax, Hit& h, const Stack& st, int sp_base, const Rng& rng,
                     uint32_t depth, Counters& cnt) {
  Trav t;
  trav_begin<!BINARY>(t, o, d, tmin, tmax, root, bxmin, bxmax, bymin, bymax, bzmin, bzmax, sp_base);
  while (t.cur != REF_DONE) trav_step<COUNT, MEDIUM, BINARY>(S, t, st, rng, depth, cnt);
  h = t.h;
}

// ConstantMedium::hit (src/hittable/constant_medium.rs:35-79). Its draws come from the sub-stream
// 0x40000000 + (depth<<20 | medium<<8) + i of the path's generator (DESIGN.md "RNG"), so the outcome does not
// depend on when the tree search reaches the medium.
template <bool COUNT>
DEV bool medium_test(const DevScene& S, uint32_t midx, f3 o, f3 d, float tmin, float tmax, float& t_out, const Stack& st,
                     int sp, const Rng& rng, uint32_t depth, Counters& cnt) {
  const DMedium M = ldg_rec(S.mediums + midx);
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
