// sol_build.hip -- the world's tree built ON THE GPU (SolCreateOptions.world_tree = SOL_TREE_DEVICE).
//
// Replaces, for the world search, the host builders of sol_tree.h (themselves a results-neutral replacement of the reference's
// centroid-median build, src/hittable/bvh.rs:84-162): the closest hit does not depend on the tree, so any
// conservative tree over the same primitives may be walked. Input: the world's primitives as (reference, padded fp32 box) in
// any order. Output: the device form of the 7-wide tree (DWide, sol_types.h) + the permutations of the primitive arrays, in
// the same WideLayout record the host path produces, so everything downstream (upload, checks, kernels) is shared.
//
//   0. Triangle PRE-SPLITTING (after Karras & Aila 2013, "Fast parallel construction of high-quality bounding volume hierarchies",
//      sec. 4.3): a triangle whose box straddles an important spatial-median plane of the scene - a plane of the Morton grid that
//      separates groups of many primitives - is cut there, recursively, into several REFERENCES with tight boxes (the triangle
//      clipped against the plane, Stich et al. 2009). Large wall triangles beside small ornament, long thin rails, diagonal rods:
//      their one big, mostly empty box would otherwise overlap everything near it at every level of the tree. All references of
//      a triangle name the same primitive (t, dfs): `better` keeps the closest hit whichever reference finds it, and a hit point lies
//      inside the reference box of the part it belongs to, so results do not change (sol_trace.h). The priority rule
//      (2^-level * (box area - ideal area))^(1/3) hands out a budget of extra references; parts stop splitting when no plane above
//      the level of few-primitive cells crosses them.
//   1. Morton codes of the box centres (63 bits) and a radix sort (rocPRIM).
//   2. Binary tree by PLOC - parallel locally-ordered clustering (Meister & Bittner 2018): every cluster looks R neighbours
//      to each side in the sorted order for the partner with the smallest joint surface, mutual choices merge; repeated until
//      one cluster is left (~25 rounds for 262 k primitives). Bottom-up and surface-area driven like a SAH build, with no
//      sequential sweep: rounds of three small kernels and a prefix sum.
//   3. The surface-area-optimal collapse into 7-wide nodes of sol_tree.h (Ylitie, Karras & Laine 2017, sec. 4.1), bottom-up
//      over the binary tree (a node is computed by whichever of its two children's threads arrives second).
//   4. Emission level by level, one thread per wide node: gather the (up to 7) children the collapse chose, assign octant
//      slots (Kuhn-Munkres on the 7x7 problem, as on the host), quantise the child boxes with the device's own decode
//      arithmetic, and reserve consecutive node / primitive indices for the children - which is exactly the implicit-address
//      layout DWide needs.
// None of this is on the per-sample path. The only library call is rocPRIM's sort / scan (plain primitives).
#include <hip/hip_runtime.h>

#include <functional>
#include <memory>
#include <chrono>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <string>
#include <vector>

#include "../../include/solstrale_hip.h"
#include "sol_build.h"
#include "sol_types.h"

namespace {

constexpr int BT = 256;          // threads per block
constexpr int PLOC_R_MAX = 64;   // neighbours searched to each side: at most (LDS window); default 16 (SOL_PLOC_R)
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr float PRIM_COST = 1.0f;  // as WideBuilder (sol_tree.h); the cost of a wide node is SolSplitOptions::node_cost (2.5)
constexpr int MAXC = SOL_WIDE_CHILDREN;

struct Dp {
  float c[8];
  uint8_t eff[8], split[8];
};

__device__ __forceinline__ float box_area(const float* b) {
  const float dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
  if (!(dx >= 0.f && dy >= 0.f && dz >= 0.f)) return 0.f;
  return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ float union_area(const float* a, const float* b) {
  const float dx = fmaxf(a[1], b[1]) - fminf(a[0], b[0]), dy = fmaxf(a[3], b[3]) - fminf(a[2], b[2]), dz = fmaxf(a[5], b[5]) - fminf(a[4], b[4]);
  return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ unsigned long long spread21(unsigned long long v) {  // 21 bits -> every third bit
  v &= 0x1FFFFFull;
  v = (v | (v << 32)) & 0x001F00000000FFFFull;
  v = (v | (v << 16)) & 0x001F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

__global__ void __launch_bounds__(BT) k_morton(const SolBuildPrim* __restrict__ p, uint32_t n, float cx, float cy, float cz, float sx, float sy, float sz,
                                                unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * BT + threadIdx.x;
  if (i >= n) return;
  const float* b = p[i].box;
  // 21 bits per axis: with 10 (a 1024^3 grid) the million small triangles of a dense mesh share cells, and their order inside
  // a cell - hence the clusters - is arbitrary (C5: 12 % more node visits than the host's SAH tree)
  const float x = (0.5f * (b[0] + b[1]) - cx) * sx, y = (0.5f * (b[2] + b[3]) - cy) * sy, z = (0.5f * (b[4] + b[5]) - cz) * sz;
  const unsigned long long xi = (unsigned long long)fminf(fmaxf(x, 0.f), 2097151.f), yi = (unsigned long long)fminf(fmaxf(y, 0.f), 2097151.f),
                           zi = (unsigned long long)fminf(fmaxf(z, 0.f), 2097151.f);
  keys[i] = (spread21(xi) << 2) | (spread21(yi) << 1) | spread21(zi);
  vals[i] = i;
}

// ---- 0. pre-splitting --------------------------------------------------------------------------------------------------------
constexpr int GRID_BITS = 21;              // the Morton grid of k_morton
constexpr int GRID_CELLS = 1 << GRID_BITS;
constexpr int SPLIT_CAP = 63;              // most splits of one triangle
struct SplitGrid {
  float lo[3], inv_cell[3], cell[3];
  int level_max;   // a plane is worth a split when its level (0 = the scene's median plane on x, 1 = y, 2 = z, 3 = the quarter planes on x, ..) is below
  float pad;
};
struct Piece {
  float b[6];  // tight box of this part of the triangle
  int q[6];    // grid cells it may still be cut in (narrowed at every split: a part never crosses the same plane twice)
  int s;       // splits it may still spend
};
__device__ __forceinline__ void cell_range(const SplitGrid& G, const float* b, int* q) {
  for (int a = 0; a < 3; ++a) {
    int lo = (int)floorf((b[2 * a] - G.lo[a]) * G.inv_cell[a]), hi = (int)ceilf((b[2 * a + 1] - G.lo[a]) * G.inv_cell[a]) - 1;
    lo = max(q[2 * a], min(GRID_CELLS - 1, max(0, lo)));
    hi = min(q[2 * a + 1], min(GRID_CELLS - 1, max(0, hi)));
    if (hi < lo) hi = lo;
    q[2 * a] = lo; q[2 * a + 1] = hi;
  }
}
// The most important plane crossing the cell range: its level, axis and position (first cell of the upper side); level 1 << 30: none.
__device__ __forceinline__ int best_plane(const int* q, int& axis, int& pos) {
  int best = 1 << 30;
  for (int a = 0; a < 3; ++a) {
    const int x = q[2 * a] ^ q[2 * a + 1];
    if (!x) continue;
    const int bit = 31 - __clz(x);
    const int level = 3 * (GRID_BITS - 1 - bit) + a;
    if (level < best) { best = level; axis = a; pos = (q[2 * a + 1] >> bit) << bit; }
  }
  return best;
}
__device__ __forceinline__ void tri_vertices(const DTri& T, float v[3][3]) {
  v[0][0] = T.v0x; v[0][1] = T.v0y; v[0][2] = T.v0z;
  v[1][0] = T.v0x + T.e1x; v[1][1] = T.v0y + T.e1y; v[1][2] = T.v0z + T.e1z;
  v[2][0] = T.v0x + T.e2x; v[2][1] = T.v0y + T.e2y; v[2][2] = T.v0z + T.e2z;
}
__device__ __forceinline__ void grow_pt(float* b, const float* p) {
  for (int a = 0; a < 3; ++a) { b[2 * a] = fminf(b[2 * a], p[a]); b[2 * a + 1] = fmaxf(b[2 * a + 1], p[a]); }
}
// Boxes of (triangle with x_axis <= s) and (triangle with x_axis >= s), each cut down to the parent part's box. False: a side is empty.
__device__ bool clip_sides(const float v[3][3], const float* parent, int axis, float s, float* bl, float* br) {
  const float inf = __builtin_huge_valf();
  for (int a = 0; a < 3; ++a) { bl[2 * a] = br[2 * a] = inf; bl[2 * a + 1] = br[2 * a + 1] = -inf; }
  for (int i = 0; i < 3; ++i) {
    const float* vi = v[i];
    const float* vj = v[(i + 1) % 3];
    if (vi[axis] <= s) grow_pt(bl, vi);
    if (vi[axis] >= s) grow_pt(br, vi);
    if ((vi[axis] < s && vj[axis] > s) || (vi[axis] > s && vj[axis] < s)) {
      const float t = (s - vi[axis]) / (vj[axis] - vi[axis]);
      float p[3] = {vi[0] + (vj[0] - vi[0]) * t, vi[1] + (vj[1] - vi[1]) * t, vi[2] + (vj[2] - vi[2]) * t};
      p[axis] = s;
      grow_pt(bl, p);
      grow_pt(br, p);
    }
  }
  bool ok = true;
  for (int a = 0; a < 3; ++a) {
    bl[2 * a] = fmaxf(bl[2 * a], parent[2 * a]); bl[2 * a + 1] = fminf(bl[2 * a + 1], parent[2 * a + 1]);
    br[2 * a] = fmaxf(br[2 * a], parent[2 * a]); br[2 * a + 1] = fminf(br[2 * a + 1], parent[2 * a + 1]);
  }
  bl[2 * axis + 1] = fminf(bl[2 * axis + 1], s);
  br[2 * axis] = fmaxf(br[2 * axis], s);
  for (int a = 0; a < 3; ++a) ok = ok && bl[2 * a] <= bl[2 * a + 1] && br[2 * a] <= br[2 * a + 1];
  return ok;
}
__device__ __forceinline__ float longest(const float* b) { return fmaxf(b[1] - b[0], fmaxf(b[3] - b[2], b[5] - b[4])); }

// Splits triangle `v` with a budget of `s` splits; calls emit(box) for every final part (tight box, unpadded). Explicit stack: the
// side with MORE splits left is pushed, the other continued, so the stack never holds more than log2(s) + 1 parts.
template <typename Emit>
__device__ void split_triangle(const SplitGrid& G, const float v[3][3], int s, Emit emit) {
  Piece stk[8];
  int sp = 0;
  Piece cur;
  const float inf = __builtin_huge_valf();
  for (int a = 0; a < 3; ++a) { cur.b[2 * a] = inf; cur.b[2 * a + 1] = -inf; cur.q[2 * a] = 0; cur.q[2 * a + 1] = GRID_CELLS - 1; }
  for (int i = 0; i < 3; ++i) grow_pt(cur.b, v[i]);
  cur.s = s;
  cell_range(G, cur.b, cur.q);
  for (;;) {
    int axis = 0, pos = 0;
    bool split = false;
    Piece l, r;
    if (cur.s > 0 && best_plane(cur.q, axis, pos) < G.level_max) {
      const float sv = G.lo[axis] + (float)pos * G.cell[axis];
      if (sv > cur.b[2 * axis] && sv < cur.b[2 * axis + 1] && clip_sides(v, cur.b, axis, sv, l.b, r.b)) {
        for (int k = 0; k < 6; ++k) { l.q[k] = cur.q[k]; r.q[k] = cur.q[k]; }
        l.q[2 * axis + 1] = pos - 1;
        r.q[2 * axis] = pos;
        cell_range(G, l.b, l.q);
        cell_range(G, r.b, r.q);
        const float wl = longest(l.b), wr = longest(r.b);
        const int rest = cur.s - 1;
        int sl = (wl + wr > 0.f) ? (int)floorf((float)rest * wl / (wl + wr) + 0.5f) : rest / 2;
        sl = min(rest, max(0, sl));
        l.s = sl;
        r.s = rest - sl;
        split = true;
      } else {
        // (the plane does not really cut this part - it lies on the part's face: the cell range was rounded outwards): take it out
        // of the range and look again, without spending a split
        if (sv <= cur.b[2 * axis]) cur.q[2 * axis] = max(cur.q[2 * axis], pos); else cur.q[2 * axis + 1] = min(cur.q[2 * axis + 1], pos - 1);
        if (cur.q[2 * axis + 1] < cur.q[2 * axis]) cur.q[2 * axis + 1] = cur.q[2 * axis];
        if (sv > cur.b[2 * axis] && sv < cur.b[2 * axis + 1]) cur.s = 0;  // (a clip that failed numerically: stop here)
        continue;
      }
    }
    if (split) {
      const bool push_left = l.s > r.s;
      if (sp < 8) { stk[sp++] = push_left ? l : r; cur = push_left ? r : l; continue; }
      // (cannot happen for s <= SPLIT_CAP; be safe: stop splitting the larger side)
      Piece big = push_left ? l : r;
      emit(big.b);
      cur = push_left ? r : l;
      continue;
    }
    emit(cur.b);
    if (sp == 0) break;
    cur = stk[--sp];
  }
}

// priority of a primitive for the split budget (0: not a triangle, or no plane worth a split crosses it)
__global__ void __launch_bounds__(BT) k_split_priority(const SolBuildPrim* __restrict__ p, uint32_t n, const DTri* __restrict__ tris, uint32_t n_tris, SplitGrid G,
                                                        float* __restrict__ prio) {
  const uint32_t i = blockIdx.x * BT + threadIdx.x;
  if (i >= n) return;
  float pr = 0.f;
  const uint32_t ref = p[i].ref;
  if (SOL_REF_KIND(ref) == SOL_REF_TRIANGLE && SOL_REF_INDEX(ref) < n_tris) {
    const DTri T = tris[SOL_REF_INDEX(ref)];
    float v[3][3];
    tri_vertices(T, v);
    const float inf = __builtin_huge_valf();
    float b[6] = {inf, -inf, inf, -inf, inf, -inf};
    for (int k = 0; k < 3; ++k) grow_pt(b, v[k]);
    int q[6] = {0, GRID_CELLS - 1, 0, GRID_CELLS - 1, 0, GRID_CELLS - 1};
    cell_range(G, b, q);
    int axis, pos;
    const int level = best_plane(q, axis, pos);
    if (level < G.level_max) {
      const float dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
      const float a_box = 2.0f * (dx * dy + dy * dz + dz * dx);
      const float nx = T.e1y * T.e2z - T.e1z * T.e2y, ny = T.e1z * T.e2x - T.e1x * T.e2z, nz = T.e1x * T.e2y - T.e1y * T.e2x;
      const float a_ideal = fabsf(nx) + fabsf(ny) + fabsf(nz);
      const float d = a_box - a_ideal;
      if (d > 0.f && d < inf) pr = cbrtf(exp2f(-(float)level) * d);
      if (!(pr > 0.f && pr < inf)) pr = 0.f;
    }
  }
  prio[i] = pr;
}
__global__ void __launch_bounds__(BT) k_split_sum(const float* __restrict__ prio, uint32_t n, float D, unsigned long long* __restrict__ total) {
  __shared__ unsigned int part[BT / 64];
  const uint32_t i = blockIdx.x * BT + threadIdx.x;
  unsigned int c = 0;
  if (i < n) c = (unsigned int)fminf((float)SPLIT_CAP, floorf(D * prio[i]));
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int t = 0;
    for (int k = 0; k < BT / 64; ++k) t += part[k];
    if (t) atomicAdd(total, (unsigned long long)t);
  }
}
// pass 1: how many parts each primitive becomes (1: not split); pass 2 (out != nullptr): the parts
__global__ void __launch_bounds__(BT) k_split(const SolBuildPrim* __restrict__ p, uint32_t n, const DTri* __restrict__ tris, uint32_t n_tris, SplitGrid G,
                                               const float* __restrict__ prio, float D, uint32_t* __restrict__ parts, const uint32_t* __restrict__ offset,
                                               SolBuildPrim* __restrict__ out, uint32_t* __restrict__ extra_of, uint32_t* __restrict__ n_split,
                                               unsigned long long* __restrict__ areas, double area_scale) {
  // areas[0] / [1]: summed box area of the primitives before / after splitting, in FIXED POINT (area x area_scale, 2^40 per root-box area): integer
  // atomics add up to the same total in any order, so whether the splits are kept (the 0.85 threshold) is the same decision in every run and on
  // every rank (round-4 advisor: double atomics rounded order-dependently; near the threshold two ranks of a job could keep different trees)
  auto fixed = [&](float a) { return (unsigned long long)((double)a * area_scale); };
  const uint32_t i = blockIdx.x * BT + threadIdx.x;
  if (i >= n) return;
  const SolBuildPrim me = p[i];
  const int s = (int)fminf((float)SPLIT_CAP, floorf(D * prio[i]));
  const float a_me = box_area(me.box);
  if (s <= 0) {
    if (out) out[i] = me; else { parts[i] = 0u; if (a_me > 0.f && a_me < 1e30f) { atomicAdd(&areas[0], fixed(a_me)); atomicAdd(&areas[1], fixed(a_me)); } }
    return;
  }
  const uint32_t tri = SOL_REF_INDEX(me.ref);
  float v[3][3];
  tri_vertices(tris[tri], v);
  uint32_t k = 0;
  float a_parts = 0.f;
  const uint32_t base = out ? n + offset[i] : 0u;  // extra parts of primitive i: out[n + offset[i] ..], references n_tris + offset[i] ..
  split_triangle(G, v, s, [&](const float* b) {
    SolBuildPrim o;
    for (int a = 0; a < 3; ++a) {  // padded like every primitive box, and never larger than the triangle's own padded box
      o.box[2 * a] = fmaxf(b[2 * a] - G.pad, me.box[2 * a]);
      o.box[2 * a + 1] = fminf(b[2 * a + 1] + G.pad, me.box[2 * a + 1]);
    }
    a_parts += box_area(o.box);
    if (out) {
      o.pad = 0;
      if (k == 0) { o.ref = me.ref; out[i] = o; }
      else {
        const uint32_t e = offset[i] + (k - 1);
        o.ref = SOL_MAKE_REF(SOL_REF_TRIANGLE, n_tris + e);
        out[base + (k - 1)] = o;
        extra_of[e] = tri;
      }
    }
    ++k;
  });
  if (!out) {
    parts[i] = k - 1u;
    if (k > 1u) atomicAdd(n_split, 1u);
    if (a_me > 0.f && a_me < 1e30f) { atomicAdd(&areas[0], fixed(a_me)); atomicAdd(&areas[1], fixed(fminf(a_parts, a_me * 64.f))); }
  }
}

// leaves = nodes 0 .. n-1 in sorted order
__global__ void __launch_bounds__(BT) k_leaves(const SolBuildPrim* __restrict__ p, const uint32_t* __restrict__ order, uint32_t n, float* __restrict__ nbox,
                                                uint32_t* __restrict__ cluster, uint32_t* __restrict__ parent) {
  const uint32_t i = blockIdx.x * BT + threadIdx.x;
  if (i >= n) return;
  const float* b = p[order[i]].box;
  for (int k = 0; k < 6; ++k) nbox[(size_t)i * 6 + k] = b[k];
  cluster[i] = i;
  parent[i] = NONE;
}

// nearest neighbour (smallest joint surface) within PLOC_R positions; ties go to the smaller position
__global__ void __launch_bounds__(BT) k_nn(const uint32_t* __restrict__ cluster, uint32_t n, const float* __restrict__ nbox, int PLOC_R, uint32_t* __restrict__ nn) {
  __shared__ float sb[(BT + 2 * PLOC_R_MAX) * 6];
  const int base = (int)(blockIdx.x * BT) - PLOC_R;
  for (int k = threadIdx.x; k < BT + 2 * PLOC_R; k += BT) {
    const int g = base + k;
    if (g >= 0 && g < (int)n) {
      const float* b = nbox + (size_t)cluster[g] * 6;
      for (int c = 0; c < 6; ++c) sb[k * 6 + c] = b[c];
    }
  }
  __syncthreads();
  const uint32_t i = blockIdx.x * BT + threadIdx.x;
  if (i >= n) return;
  const float* me = sb + (threadIdx.x + PLOC_R) * 6;
  float best = __builtin_huge_valf();
  uint32_t bj = NONE;
  const int lo = max(0, (int)i - PLOC_R), hi = min((int)n - 1, (int)i + PLOC_R);
  for (int j = lo; j <= hi; ++j) {
    if (j == (int)i) continue;
    const float a = union_area(me, sb + (j - base) * 6);
    if (a < best) { best = a; bj = (uint32_t)j; }
  }
  nn[i] = bj;
}

// mutual nearest neighbours merge into a new node (kept at the smaller position); everything else is carried over
__global__ void __launch_bounds__(BT) k_merge(const uint32_t* __restrict__ cluster, uint32_t n, const uint32_t* __restrict__ nn, float* __restrict__ nbox,
                                               uint32_t* __restrict__ left, uint32_t* __restrict__ right, uint32_t* __restrict__ parent,
                                               uint32_t* __restrict__ node_counter, uint32_t* __restrict__ out_node, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * BT + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = nn[i];
  uint32_t keep = 1u, node = cluster[i];
  if (j != NONE && nn[j] == i) {
    if (i < j) {
      const uint32_t a = cluster[i], b = cluster[j];
      node = atomicAdd(node_counter, 1u);
      left[node] = a;
      right[node] = b;
      parent[node] = NONE;
      parent[a] = node;
      parent[b] = node;
      const float *ba = nbox + (size_t)a * 6, *bb = nbox + (size_t)b * 6;
      float* bo = nbox + (size_t)node * 6;
      for (int k = 0; k < 6; k += 2) { bo[k] = fminf(ba[k], bb[k]); bo[k + 1] = fmaxf(ba[k + 1], bb[k + 1]); }
    } else {
      keep = 0u;
    }
  }
  out_node[i] = node;
  flag[i] = keep;
}

__global__ void __launch_bounds__(BT) k_compact(const uint32_t* __restrict__ out_node, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ offset,
                                                 uint32_t n, uint32_t* __restrict__ cluster_out) {
  const uint32_t i = blockIdx.x * BT + threadIdx.x;
  if (i < n && flag[i]) cluster_out[offset[i]] = out_node[i];
}

// ---- 2b. reinsertion ------------------------------------------------------------------------------------------------------------
// Parallel reinsertion (after Meister & Bittner 2018, "Parallel reinsertion for bounding volume hierarchy optimization"): every
// node x looks for the place in the tree where it - with its whole sub-tree - would sit at the smallest summed surface area, a
// branch-and-bound walk that starts at its sibling and moves up pivot by pivot; the moves with a gain are then carried out, the
// ones with the largest gain first where two of them touch the same nodes. Notation: p = parent(x), s = sibling(x); pivot P_k =
// the k-th ancestor of p, O_k its child off x's path, R_k the box of P_k once x is gone. Putting x next to y (their new parent is
// the node p, which leaves its old place to s) changes the summed area by
//   A(x u y) - A(p)  -  sum_{j=1..k-1} (A(P_j) - A(R_j))  +  sum_{c on the way from O_k down to parent(y)} (A(c u x) - A(c))
// for y below O_k (k = 0: below s); the pivot and everything above keep their boxes (same leaves below them).
// A move rewrites the links of six nodes (x, p, s, parent(p), y, parent(y)): each is stamped with (gain, x) by a 64-bit atomic
// maximum, and a move is carried out when all six stamps are its own and no node between y and the pivot is about to move with a
// larger gain (k_reins_verify: what keeps simultaneous moves from tying the tree into a cycle). Moves that share only ancestors
// commute; their gains were computed one at a time, so a round is greedy, not exact - the boxes are refitted after every round
// and the summed area is watched. Every walk is bounded (a broken tree cannot hang a kernel) and k_validate checks the result.
constexpr int REINS_MAX_UP = 96;       // pivots visited by one search
constexpr int REINS_MAX_VISITS = 768;  // nodes looked at by one search
constexpr int REINS_MAX_PATH = 256;    // nodes of one path
__device__ __forceinline__ float area_of(const float* b) { return box_area(b); }
__device__ __forceinline__ void load_box(const float* __restrict__ nbox, uint32_t i, float* b) {
  for (int k = 0; k < 6; ++k) b[k] = nbox[(size_t)i * 6 + k];
}
__device__ __forceinline__ void unite(float* a, const float* b) {
  for (int k = 0; k < 6; k += 2) { a[k] = fminf(a[k], b[k]); a[k + 1] = fmaxf(a[k + 1], b[k + 1]); }
}
__device__ __forceinline__ float union_area_n(const float* __restrict__ nbox, uint32_t i, const float* b) { return union_area(nbox + (size_t)i * 6, b); }

__global__ void __launch_bounds__(BT) k_reins_find(uint32_t n_nodes, uint32_t n_leaves, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ left,
                                                    const uint32_t* __restrict__ right, const float* __restrict__ nbox, uint32_t phase, uint32_t stride,
                                                    uint32_t* __restrict__ best_out, uint32_t* __restrict__ best_pivot, float* __restrict__ best_gain) {
  const uint32_t x = blockIdx.x * BT + threadIdx.x;
  if (x >= n_nodes) return;
  best_out[x] = NONE;
  best_gain[x] = 0.f;
  if (stride > 1u && (x % stride) != phase) return;
  const uint32_t p = parent[x];
  if (p == NONE || parent[p] == NONE) return;  // (the root and its children stay: the root node keeps its index)
  const uint32_t s = left[p] == x ? right[p] : left[p];
  float bx[6];
  load_box(nbox, x, bx);
  const float a_x = area_of(bx), a_p = area_of(nbox + (size_t)p * 6);
  if (!(a_x >= 0.f) || !(a_p < 1e30f)) return;
  float R[6];
  load_box(nbox, s, R);
  float G = 0.f, d_best = 0.f;
  uint32_t out = NONE, out_pivot = NONE;
  uint32_t pivot = p, other = s;
  int visits = 0;
  for (int up = 0; up < REINS_MAX_UP; ++up) {
    // ---- the sub-tree under `other`, depth first without a stack; I = growth of the nodes between `other` and cur's parent ----
    uint32_t cur = other;
    float I = 0.f;
    for (;;) {
      if (++visits > REINS_MAX_VISITS) break;
      const float m = union_area_n(nbox, cur, bx);
      if (cur != s) {  // (next to its own sibling is where x already is)
        const float gain = a_p - m + G - I;
        if (gain > d_best) { d_best = gain; out = cur; out_pivot = pivot; }
      }
      const float grow = m - area_of(nbox + (size_t)cur * 6);
      if (cur >= n_leaves && a_p - a_x + G - I - grow > d_best) {  // something below cur may still be better
        I += grow;
        cur = left[cur];
        continue;
      }
      // next node in depth-first order: up until a left child is found, then its right sibling
      bool done = false;
      for (;;) {
        if (cur == other) { done = true; break; }
        const uint32_t par = parent[cur];
        if (cur == left[par]) { cur = right[par]; break; }
        cur = par;
        I -= union_area_n(nbox, cur, bx) - area_of(nbox + (size_t)cur * 6);
      }
      if (done) break;
    }
    if (visits > REINS_MAX_VISITS) break;
    // ---- one pivot up ----
    const uint32_t nxt = parent[pivot];
    if (nxt == NONE) break;
    if (pivot != p) G += area_of(nbox + (size_t)pivot * 6) - area_of(R);  // P_k (k >= 1) shrinks to R_k once the pivot moves above it
    const uint32_t sib = left[nxt] == pivot ? right[nxt] : left[nxt];
    float bs[6];
    load_box(nbox, sib, bs);
    unite(R, bs);
    pivot = nxt;
    other = sib;
    // nothing at or above this pivot can gain more than A(p) - A(x) + G + what the pivots still to come may shrink by; the walk
    // simply goes on to the root (bounded by REINS_MAX_UP), the searches below prune themselves
  }
  if (out != NONE && d_best > 1e-6f * a_p) {
    best_out[x] = out;
    best_pivot[x] = out_pivot;
    best_gain[x] = d_best;
  }
}

__device__ __forceinline__ unsigned long long reins_key(float gain, uint32_t x) { return ((unsigned long long)__float_as_uint(gain) << 32) | x; }  // (gain > 0: its bits order like the value)

// The six nodes whose links a move rewrites: x, p = parent(x), s = sibling(x), g = parent(p), y and q = parent(y).
template <typename F>
__device__ bool reins_nodes(uint32_t x, uint32_t y, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, F f) {
  const uint32_t p = parent[x];
  if (p == NONE) return false;
  const uint32_t g = parent[p], q = parent[y];
  if (g == NONE || q == NONE) return false;
  f(x); f(p); f(left[p] == x ? right[p] : left[p]); f(g); f(y); f(q);
  return true;
}
__global__ void __launch_bounds__(BT) k_reins_lock(uint32_t n_nodes, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                    const uint32_t* __restrict__ best_out, const float* __restrict__ best_gain, unsigned long long* __restrict__ lock,
                                                    unsigned long long* __restrict__ in_key) {
  const uint32_t x = blockIdx.x * BT + threadIdx.x;
  if (x >= n_nodes) return;
  if (best_out[x] == NONE) { in_key[x] = 0ull; return; }
  const unsigned long long key = reins_key(best_gain[x], x);
  in_key[x] = key;
  reins_nodes(x, best_out[x], parent, left, right, [&](uint32_t c) { atomicMax(&lock[c], key); });
}
// A move is carried out when (1) the six nodes it rewrites are stamped with its own key - two moves that touch a common node: the
// larger gain wins - and (2) no node between y and the pivot is itself about to move with a larger gain: x landing inside a
// sub-tree that lands inside x's own would tie the tree into a cycle, and of every such ring of moves the one in front of the
// largest gain drops out here. (Moves that only share ancestors further up commute: each rewrites its own six nodes.)
__global__ void __launch_bounds__(BT) k_reins_verify(uint32_t n_nodes, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                      uint32_t* __restrict__ best_out, const uint32_t* __restrict__ best_pivot, const float* __restrict__ best_gain,
                                                      const unsigned long long* __restrict__ lock, const unsigned long long* __restrict__ in_key, uint8_t* __restrict__ ok) {
  const uint32_t x = blockIdx.x * BT + threadIdx.x;
  if (x >= n_nodes) return;
  ok[x] = 0;
  if (best_out[x] == NONE) return;
  const unsigned long long key = reins_key(best_gain[x], x);
  bool mine = true;
  if (!reins_nodes(x, best_out[x], parent, left, right, [&](uint32_t c) { if (lock[c] != key) mine = false; })) mine = false;
  const uint32_t pivot = best_pivot[x];
  uint32_t c = best_out[x];
  int steps = 0;
  while (mine && c != pivot) {
    if (in_key[c] > key) mine = false;
    c = parent[c];
    if (c == NONE || ++steps > REINS_MAX_PATH) mine = false;
  }
  ok[x] = mine ? 1 : 0;  // (read-only kernel as far as the tree goes: the moves are carried out by the next one)
}
__global__ void __launch_bounds__(BT) k_reins_apply(uint32_t n_nodes, uint32_t* __restrict__ parent, uint32_t* __restrict__ left, uint32_t* __restrict__ right,
                                                     const uint32_t* __restrict__ best_out, const uint8_t* __restrict__ ok, uint32_t* __restrict__ n_moved,
                                                     uint32_t* __restrict__ left_from) {
  const uint32_t x = blockIdx.x * BT + threadIdx.x;
  if (x >= n_nodes || !ok[x]) return;
  const uint32_t y = best_out[x], p = parent[x], g = parent[p];
  left_from[x] = g;  // (k_refit_mark: the boxes from x's old place upwards change too)
  const uint32_t s = left[p] == x ? right[p] : left[p];
  // s takes p's place under g
  if (left[g] == p) left[g] = s; else right[g] = s;
  parent[s] = g;
  // p moves between y and y's parent (read after the step above: y's parent may be g)
  const uint32_t q = parent[y];
  if (left[q] == y) left[q] = p; else right[q] = p;
  parent[p] = q;
  left[p] = x;
  right[p] = y;
  parent[y] = p;
  atomicAdd(n_moved, 1u);
}
// ---- refit after a reinsertion round: only what the moves touched (round 5) --------------------------------------------------------
// A move changes the boxes on two paths to the root: from p (x's parent, now between the target and the target's old parent) and from g
// (p's old parent, which lost the sub-tree). Round 4 recomputed every box of the tree after every round (bottom-up from all leaves, the second
// child to arrive computing its parent) - 3.3 ms for the 2.2 M nodes of C5, 9 times per tree, a third of scene creation's GPU time. Now: (1) every move marks its two paths DIRTY
// (a walk stops at a node another walk has marked: that walk marks the rest); (2) every dirty node counts its CLEAN children as arrived -
// a dirty node whose children are both clean is a start; (3) walks go up from the starts, the second arrival at a node computes it (the
// arrival pattern of the full refit - a release fence before the arrival, an acquire fence after it -, on the dirty set only). Boxes are unions - exact -, so the result is the full refit's, bit for bit.
__global__ void __launch_bounds__(BT) k_refit_mark(uint32_t n_nodes, const uint32_t* __restrict__ parent, const uint8_t* __restrict__ ok,
                                                    const uint32_t* __restrict__ left_from, uint32_t* __restrict__ dirty) {
  const uint32_t x = blockIdx.x * BT + threadIdx.x;
  if (x >= n_nodes || !ok[x]) return;
  for (int k = 0; k < 2; ++k) {
    uint32_t m = k == 0 ? parent[x] : left_from[x];
    int steps = 0;
    while (m != NONE && ++steps < 4096) {
      if (atomicExch(&dirty[m], 1u) != 0u) break;
      m = parent[m];
    }
  }
}
__global__ void __launch_bounds__(BT) k_refit_prepare(uint32_t n_nodes, uint32_t n_leaves, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                       uint32_t* __restrict__ dirty, uint32_t* __restrict__ arrived) {
  const uint32_t m = blockIdx.x * BT + threadIdx.x;
  if (m >= n_nodes || m < n_leaves || dirty[m] == 0u) return;
  // (dirty[] of the children is 0 or 1 here: the values 2 are written to other nodes' own entries only, and read by nobody in this launch)
  const uint32_t clean = (dirty[left[m]] == 0u ? 1u : 0u) + (dirty[right[m]] == 0u ? 1u : 0u);
  arrived[m] = clean;
}
__global__ void __launch_bounds__(BT) k_refit_starts(uint32_t n_nodes, uint32_t n_leaves, uint32_t* __restrict__ dirty, const uint32_t* __restrict__ arrived) {
  const uint32_t m = blockIdx.x * BT + threadIdx.x;
  if (m >= n_nodes || m < n_leaves || dirty[m] == 0u) return;
  if (arrived[m] == 2u) dirty[m] = 2u;  // both children clean: a walk starts here
}
__global__ void __launch_bounds__(BT) k_refit_walk(uint32_t n_nodes, uint32_t n_leaves, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ left,
                                                    const uint32_t* __restrict__ right, float* __restrict__ nbox, uint32_t* __restrict__ dirty, uint32_t* __restrict__ arrived) {
  const uint32_t i = blockIdx.x * BT + threadIdx.x;
  if (i >= n_nodes || i < n_leaves || dirty[i] != 2u) return;
  uint32_t m = i;
  int steps = 0;
  for (;;) {
    const float *ba = nbox + (size_t)left[m] * 6, *bb = nbox + (size_t)right[m] * 6;
    float* bo = nbox + (size_t)m * 6;
    for (int k = 0; k < 6; k += 2) { bo[k] = fminf(ba[k], bb[k]); bo[k + 1] = fmaxf(ba[k + 1], bb[k + 1]); }
    dirty[m] = 0u;  // (clean again for the next round; nobody reads this entry any more in this launch: its parent's count was taken in k_refit_prepare)
    m = parent[m];
    if (m == NONE || ++steps >= 4096) break;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (this thread's boxes first, then its arrival; the second arrival acquires)
    if (atomicAdd(&arrived[m], 1u) != 1u) break;        // the other dirty child's walk will do this node
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}
// Summed surface area of the inner nodes (SolSceneInfo: what the reinsertion rounds bought).
__global__ void __launch_bounds__(BT) k_area_sum(uint32_t n_nodes, uint32_t n_leaves, const float* __restrict__ nbox, double* __restrict__ cost) {
  const uint32_t m = blockIdx.x * BT + threadIdx.x;
  double sum = (m < n_nodes && m >= n_leaves) ? (double)box_area(nbox + (size_t)m * 6) : 0.;
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
  if ((threadIdx.x & 63u) == 0u && sum > 0.) atomicAdd(cost, sum);
}

// After the reinsertion rounds: every inner node's children name it as their parent, are two different nodes, and its box is the
// union of theirs; the root has no parent; every leaf reaches the root. [0] broken links, [1] wrong boxes, [2] leaves cut off from the root.
__global__ void __launch_bounds__(BT) k_validate(uint32_t n_nodes, uint32_t n_leaves, uint32_t root, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ left,
                                                  const uint32_t* __restrict__ right, const float* __restrict__ nbox, uint32_t* __restrict__ bad) {
  const uint32_t m = blockIdx.x * BT + threadIdx.x;
  if (m >= n_nodes) return;
  if ((m == root) != (parent[m] == NONE)) atomicAdd(&bad[0], 1u);
  if (m < n_leaves) {
    // every leaf reaches the root: moves that tied a sub-tree into a ring pass the local checks below (each node of the ring has a parent
    // that lists it) and would surface later as a bad permutation, with a misleading message (round-4 advisor)
    uint32_t a = m;
    int steps = 0;
    while (parent[a] != NONE && ++steps < 8192) a = parent[a];
    if (a != root) atomicAdd(&bad[2], 1u);
    return;
  }
  const uint32_t l = left[m], r = right[m];
  if (l >= n_nodes || r >= n_nodes || l == r || parent[l] != m || parent[r] != m) { atomicAdd(&bad[0], 1u); return; }
  for (int k = 0; k < 6; k += 2) {
    const float lo = fminf(nbox[(size_t)l * 6 + k], nbox[(size_t)r * 6 + k]), hi = fmaxf(nbox[(size_t)l * 6 + k + 1], nbox[(size_t)r * 6 + k + 1]);
    if (nbox[(size_t)m * 6 + k] != lo || nbox[(size_t)m * 6 + k + 1] != hi) { atomicAdd(&bad[1], 1u); break; }
  }
}

// Collapse costs, bottom-up (sol_tree.h, WideBuilder::dp_compute): C(m, i) = cheapest representation of binary sub-tree m in at
// most i child slots of its parent.
__global__ void __launch_bounds__(BT) k_collapse_cost(uint32_t n_leaves, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ left,
                                                       const uint32_t* __restrict__ right, const float* __restrict__ nbox, uint32_t* __restrict__ arrived,
                                                       Dp* __restrict__ dp, float NODE_COST) {
  const uint32_t i = blockIdx.x * BT + threadIdx.x;
  if (i >= n_leaves) return;
  {
    const float a = PRIM_COST * box_area(nbox + (size_t)i * 6);
    Dp d;
    for (int k = 0; k < 8; ++k) { d.c[k] = a; d.eff[k] = 0; d.split[k] = 0; }
    dp[i] = d;
  }
  uint32_t m = parent[i];
  while (m != NONE) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (as in k_refit_walk: this thread's table first, then its arrival)
    if (atomicAdd(&arrived[m], 1u) == 0u) return;  // the sibling's thread will do this node
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const Dp dl = dp[left[m]], dr = dp[right[m]];
    Dp e;
    float dist[8];
    e.c[0] = 0.f; e.eff[0] = 0; e.split[0] = 0; e.split[1] = 0;
    for (int j = 2; j <= MAXC; ++j) {
      dist[j] = __builtin_huge_valf();
      e.split[j] = 1;
      for (int k = 1; k < j; ++k) {
        const float v = dl.c[k] + dr.c[j - k];
        if (v < dist[j]) { dist[j] = v; e.split[j] = (uint8_t)k; }
      }
    }
    e.c[1] = NODE_COST * box_area(nbox + (size_t)m * 6) + dist[MAXC];
    e.eff[1] = 1;
    for (int k = 2; k <= MAXC; ++k) {
      if (dist[k] < e.c[k - 1]) { e.c[k] = dist[k]; e.eff[k] = (uint8_t)k; }
      else { e.c[k] = e.c[k - 1]; e.eff[k] = e.eff[k - 1]; }
    }
    dp[m] = e;
    m = parent[m];
  }
}

struct Frontier {
  uint32_t node, wide;
};

struct EmitParams {
  const SolBuildPrim* prims;
  const uint32_t* order;      // sorted position -> input primitive
  const uint32_t *left, *right;
  const float* nbox;
  const Dp* dp;
  uint32_t n_leaves;
  float pad;
  uint32_t emin;
  DWide* wides;
  uint32_t* leaf_refs;
  uint32_t* new_index[3];     // triangles / spheres / quads: caller's index -> device index
  uint32_t* counters;         // [0] wide nodes, [1] frontier out, [2] leaf refs, [3..5] primitives per array, [6] error flags
};

__device__ __forceinline__ int arr_of(uint32_t kind) { return kind == SOL_REF_TRIANGLE ? 0 : kind == SOL_REF_SPHERE ? 1 : kind == SOL_REF_QUAD ? 2 : -1; }

// One wide node per thread (WideBuilder::build + WideLayout::run of sol_tree.h in one step).
__global__ void __launch_bounds__(64) k_emit(EmitParams P, const Frontier* __restrict__ in, uint32_t n_in, Frontier* __restrict__ out) {
  const uint32_t ti = blockIdx.x * 64 + threadIdx.x;
  if (ti >= n_in) return;
  const Frontier f = in[ti];
  // ---- children chosen by the collapse ----
  uint32_t child[MAXC];
  int nc = 0;
  {
    uint32_t sn[MAXC + 1];
    uint8_t si[MAXC + 1];
    int sp = 0;
    if (f.node < P.n_leaves) {  // (a world of one primitive: the root is that leaf)
      child[nc++] = f.node;
    } else {
      const int k = P.dp[f.node].split[MAXC];
      sn[sp] = P.right[f.node]; si[sp++] = (uint8_t)(MAXC - k);
      sn[sp] = P.left[f.node]; si[sp++] = (uint8_t)k;
    }
    while (sp > 0) {
      const uint32_t m = sn[--sp];
      const int i = si[sp];
      if (m < P.n_leaves) { if (nc < MAXC) child[nc++] = m; continue; }
      const int j = P.dp[m].eff[i];
      if (j <= 1) { if (nc < MAXC) child[nc++] = m; continue; }
      const int k = P.dp[m].split[j];
      sn[sp] = P.right[m]; si[sp++] = (uint8_t)(j - k);
      sn[sp] = P.left[m]; si[sp++] = (uint8_t)k;
    }
  }
  // ---- node box, quantisation grid (as the host: WideBuilder::build) ----
  const float pad = P.pad;
  float lo[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()}, hi[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
  for (int c = 0; c < nc; ++c) {
    const float* b = P.nbox + (size_t)child[c] * 6;
    for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], b[2 * a] - 2.0f * pad); hi[a] = fmaxf(hi[a], b[2 * a + 1] + 2.0f * pad); }
  }
  uint32_t eb[3];
  float scale[3];
  for (int a = 0; a < 3; ++a) {
    if (!(hi[a] >= lo[a])) { lo[a] = 0.f; hi[a] = 0.f; }
    int e = 1;
    const float ext = hi[a] - lo[a];
    if (ext > 0.f && ext < __builtin_huge_valf()) {
      int ex;
      frexpf(ext / 255.0f, &ex);
      e = min(254, max(1, ex + 127));
    }
    if (e > (int)P.emin + 31) atomicOr(&P.counters[6], 1u);
    e = max((int)P.emin, min((int)P.emin + 31, e));
    eb[a] = (uint32_t)e;
    scale[a] = __uint_as_float(eb[a] << 23);
  }
  // ---- slots: maximise the summed projections of the child centres on their slots' octant directions (Kuhn-Munkres) ----
  int slot_of[MAXC];
  {
    const float ctr[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
    float cost[MAXC + 1][MAXC + 1];
    for (int i = 1; i <= nc; ++i) {
      const float* b = P.nbox + (size_t)child[i - 1] * 6;
      const float off[3] = {0.5f * (b[0] + b[1]) - ctr[0], 0.5f * (b[2] + b[3]) - ctr[1], 0.5f * (b[4] + b[5]) - ctr[2]};
      for (int s = 0; s < MAXC; ++s) {
        const float d = ((s & 4) ? off[0] : -off[0]) + ((s & 2) ? off[1] : -off[1]) + ((s & 1) ? off[2] : -off[2]);
        cost[i][s + 1] = (d == d && fabsf(d) < __builtin_huge_valf()) ? -d : 0.f;
      }
    }
    float u[MAXC + 1], v[MAXC + 1], minv[MAXC + 1];
    int p[MAXC + 1], way[MAXC + 1];
    bool used[MAXC + 1];
    for (int j = 0; j <= MAXC; ++j) { u[j] = 0.f; v[j] = 0.f; p[j] = 0; way[j] = 0; }
    for (int i = 1; i <= nc; ++i) {
      p[0] = i;
      int j0 = 0;
      for (int j = 0; j <= MAXC; ++j) { minv[j] = __builtin_huge_valf(); used[j] = false; }
      int guard = 0;
      do {
        used[j0] = true;
        const int i0 = p[j0];
        float delta = __builtin_huge_valf();
        int j1 = 0;
        for (int j = 1; j <= MAXC; ++j)
          if (!used[j]) {
            const float cur = cost[i0][j] - u[i0] - v[j];
            if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
            if (minv[j] < delta) { delta = minv[j]; j1 = j; }
          }
        for (int j = 0; j <= MAXC; ++j)
          if (used[j]) { u[p[j]] += delta; v[j] -= delta; } else minv[j] -= delta;
        j0 = j1;
      } while (p[j0] != 0 && ++guard < 64);
      guard = 0;
      do { const int j1 = way[j0]; p[j0] = p[j1]; j0 = j1; } while (j0 && ++guard < 64);
    }
    bool taken[MAXC];
    for (int s = 0; s < MAXC; ++s) taken[s] = false;
    for (int i = 0; i < nc; ++i) slot_of[i] = -1;
    for (int j = 1; j <= MAXC; ++j)
      if (p[j] > 0 && p[j] <= nc && slot_of[p[j] - 1] < 0) { slot_of[p[j] - 1] = j - 1; taken[j - 1] = true; }
    for (int i = 0; i < nc; ++i)  // (numerical trouble in the assignment: any free slot - the layout stays valid)
      if (slot_of[i] < 0)
        for (int s = 0; s < MAXC; ++s)
          if (!taken[s]) { slot_of[i] = s; taken[s] = true; break; }
  }
  // ---- quantised planes; empty slots: inverted box ----
  uint32_t q[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t used_mask = 0u;
  for (int c = 0; c < nc; ++c) used_mask |= 1u << slot_of[c];
  for (int s = 0; s < 8; ++s)
    if (!(used_mask & (1u << s)))
      for (int a = 0; a < 3; ++a) q[2 * a + (s >> 2)] |= 255u << (8 * (s & 3));
  uint32_t imask = 0u, lmask = 0u;
  uint32_t slot_child[MAXC];
  for (int s = 0; s < MAXC; ++s) slot_child[s] = NONE;
  for (int c = 0; c < nc; ++c) {
    const int s = slot_of[c];
    slot_child[s] = child[c];
    if (child[c] >= P.n_leaves) imask |= 1u << s; else lmask |= 1u << s;
    const float* b = P.nbox + (size_t)child[c] * 6;
    for (int a = 0; a < 3; ++a) {
      const float cl = b[2 * a] - 2.0f * pad, chh = b[2 * a + 1] + 2.0f * pad;  // two more pads: the device evaluates the planes in t-space (sol_tree.h)
      long ql = (long)floorf((cl - lo[a]) / scale[a]), qh = (long)ceilf((chh - lo[a]) / scale[a]);
      ql = min(255L, max(0L, ql));
      qh = min(255L, max(0L, qh));
      while (ql > 0 && lo[a] + (float)ql * scale[a] > cl) --ql;      // conservative under the device's own decode arithmetic
      while (qh < 255 && lo[a] + (float)qh * scale[a] < chh) ++qh;
      if (lo[a] + (float)ql * scale[a] > cl || lo[a] + (float)qh * scale[a] < chh) { ql = 0; qh = 255; atomicOr(&P.counters[6], 2u); }
      q[2 * a + (s >> 2)] |= (uint32_t)ql << (8 * (s & 3));
      q[6 + 2 * a + (s >> 2)] |= (uint32_t)qh << (8 * (s & 3));
    }
  }
  // ---- implicit addresses: consecutive node indices for the inner children, consecutive primitive indices for the leaves ----
  const uint32_t n_inner = __popc(imask), n_leaf = __popc(lmask);
  uint32_t base_inner = 0u, base_prim = 0u, leaf_kind = SOL_LEAF_REFS;
  if (n_inner) {
    base_inner = atomicAdd(&P.counters[0], n_inner);
    const uint32_t o = atomicAdd(&P.counters[1], n_inner);
    uint32_t r = 0;
    for (int s = 0; s < MAXC; ++s)
      if (imask & (1u << s)) { out[o + r] = Frontier{slot_child[s], base_inner + r}; ++r; }
  }
  if (n_leaf) {
    uint32_t kind0 = SOL_REF_NONE;
    bool direct = true;
    for (int s = 0; s < MAXC; ++s)
      if (lmask & (1u << s)) {
        const uint32_t k = SOL_REF_KIND(P.prims[P.order[slot_child[s]]].ref);
        if (arr_of(k) < 0) direct = false;
        if (kind0 == SOL_REF_NONE) kind0 = k; else if (k != kind0) direct = false;
      }
    if (direct) {
      const int a = arr_of(kind0);
      leaf_kind = kind0 == SOL_REF_TRIANGLE ? SOL_LEAF_TRIANGLES : kind0 == SOL_REF_SPHERE ? SOL_LEAF_SPHERES : SOL_LEAF_QUADS;
      base_prim = atomicAdd(&P.counters[3 + a], n_leaf);
      uint32_t r = 0;
      for (int s = 0; s < MAXC; ++s)
        if (lmask & (1u << s)) P.new_index[a][SOL_REF_INDEX(P.prims[P.order[slot_child[s]]].ref)] = base_prim + r++;
    } else {
      base_prim = atomicAdd(&P.counters[2], n_leaf);
      uint32_t r = 0;
      for (int s = 0; s < MAXC; ++s)
        if (lmask & (1u << s)) {
          uint32_t ref = P.prims[P.order[slot_child[s]]].ref;
          const int a = arr_of(SOL_REF_KIND(ref));
          if (a >= 0) {
            const uint32_t ni = atomicAdd(&P.counters[3 + a], 1u);
            P.new_index[a][SOL_REF_INDEX(ref)] = ni;
            ref = SOL_MAKE_REF(SOL_REF_KIND(ref), ni);
          }
          P.leaf_refs[base_prim + r++] = ref;
        }
    }
  }
  if (base_inner + n_inner > SOL_WIDE_MAX_INDEX || base_prim + n_leaf > SOL_WIDE_MAX_INDEX) atomicOr(&P.counters[6], 4u);
  DWide w;
  w.ox = lo[0]; w.oy = lo[1]; w.oz = lo[2];
  w.meta = (eb[0] - P.emin) | ((eb[1] - P.emin) << 5) | ((eb[2] - P.emin) << 10) | (imask << 15) | (lmask << 22) | (leaf_kind << 29);
  for (int k = 0; k < 12; ++k) w.q[k] = q[k];
  for (int k = 0; k < 3; ++k) {
    w.q[2 * k + 1] = (w.q[2 * k + 1] & 0x00FFFFFFu) | (((base_inner >> (8 * k)) & 0xFFu) << 24);
    w.q[6 + 2 * k + 1] = (w.q[6 + 2 * k + 1] & 0x00FFFFFFu) | (((base_prim >> (8 * k)) & 0xFFu) << 24);
  }
  P.wides[f.wide] = w;
}

// primitives the world tree does not hold (e.g. the quads of a medium boundary) follow behind the others
__global__ void __launch_bounds__(BT) k_rest(uint32_t* __restrict__ new_index, uint32_t n, uint32_t* __restrict__ counter) {
  const uint32_t i = blockIdx.x * BT + threadIdx.x;
  if (i < n && new_index[i] == NONE) new_index[i] = atomicAdd(counter, 1u);
}

struct Scratch {  // device allocations of one build, freed on every path
  std::vector<void*> p;
  ~Scratch() { for (void* q : p) if (q) hipFree(q); }
  template <typename T>
  hipError_t get(T** out, size_t count) {
    *out = nullptr;
    hipError_t e = hipMalloc((void**)out, (count ? count : 1) * sizeof(T));
    if (e == hipSuccess) p.push_back(*out);
    return e;
  }
};

}  // namespace

#define B_TRY(expr)                                                                        \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } \
  } while (0)

// a launch that fails (bad configuration, no code object for the device) is reported under the kernel's own name
#define B_LAUNCHED(kernel)                                                                                   \
  do {                                                                                                        \
    hipError_t e_ = hipGetLastError();                                                                        \
    if (e_ != hipSuccess) { err = std::string("launch of " #kernel ": ") + hipGetErrorString(e_); return false; } \
  } while (0)

// A build in two halves: everything up to the collapse's surface-area cost of the tree (which is what a caller with several candidates
// compares), and the emission of the wide nodes + the download of the result. The device memory of the first half lives in the handle.
struct SolDeviceBuild {
  std::shared_ptr<Scratch> scratch;
  std::function<bool(SolDeviceTree&, std::string&)> emit;
};
void sol_build_world_tree_release(SolDeviceBuild* b) { delete b; }
bool sol_build_world_tree_emit(SolDeviceBuild* b, SolDeviceTree& out, std::string& err) {
  if (!b || !b->emit) { err = "device tree build: nothing prepared"; return false; }
  const bool ok = b->emit(out, err);
  b->emit = nullptr;
  b->scratch.reset();
  return ok;
}
bool sol_build_world_tree_device(const SolBuildPrim* prims, uint32_t n_in, const float root_box[6], float pad, uint32_t emin, const uint32_t counts_in[3],
                                 const DTri* tris, const SolSplitOptions& split, int ploc_radius, hipStream_t stream, SolDeviceTree& out, std::string& err) {
  SolDeviceBuild* b = nullptr;
  if (!sol_build_world_tree_prepare(prims, n_in, root_box, pad, emin, counts_in, tris, split, ploc_radius, stream, out, &b, err)) return false;
  const bool ok = sol_build_world_tree_emit(b, out, err);
  sol_build_world_tree_release(b);
  return ok;
}
bool sol_build_world_tree_prepare(const SolBuildPrim* prims, uint32_t n_in, const float root_box[6], float pad, uint32_t emin, const uint32_t counts_in[3],
                                  const DTri* tris, const SolSplitOptions& split, int ploc_radius, hipStream_t stream, SolDeviceTree& out, SolDeviceBuild** handle,
                                  std::string& err) {
  *handle = nullptr;
  if (n_in == 0 || n_in > (SOL_WIDE_MAX_INDEX >> 1)) { err = "device tree build: primitive count out of range"; return false; }
  std::shared_ptr<Scratch> Sp = std::make_shared<Scratch>();
  Scratch& S = *Sp;
  uint32_t n = n_in;
  uint32_t counts[3] = {counts_in[0], counts_in[1], counts_in[2]};
  SolBuildPrim* d_prims;
  B_TRY(S.get(&d_prims, n_in));
  B_TRY(hipMemcpyAsync(d_prims, prims, (size_t)n_in * sizeof(SolBuildPrim), hipMemcpyHostToDevice, stream));
  float ext[3], inv[3];
  for (int a = 0; a < 3; ++a) {
    ext[a] = root_box[2 * a + 1] - root_box[2 * a];
    inv[a] = (ext[a] > 0.f && ext[a] < 1e30f) ? 2097152.0f / ext[a] : 0.f;
  }
  const auto t_dbg0 = std::chrono::steady_clock::now();
  auto dbg = [&](const char* what) { if (split.verbose) { hipStreamSynchronize(stream); std::fprintf(stderr, "[solstrale]   build: %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dbg0).count()); } };
  // ---- 0. pre-splitting: d_prims (n_in) -> d_prims (n >= n_in), the extra references behind the originals ----
  out.extra_of.clear();
  out.split_triangles = 0;
  out.split_area_ratio = 1.f;
  const uint32_t budget = (tris && counts_in[0] && split.budget > 0.f) ? (uint32_t)std::min<double>((double)split.budget * n_in, (double)((SOL_WIDE_MAX_INDEX >> 1) - n_in)) : 0u;
  if (budget > 0) {
    SplitGrid G;
    for (int a = 0; a < 3; ++a) { G.lo[a] = root_box[2 * a]; G.inv_cell[a] = inv[a]; G.cell[a] = ext[a] > 0.f && ext[a] < 1e30f ? ext[a] / 2097152.0f : 0.f; }
    G.pad = pad;
    // fixed-point scale of the area sums: 2^40 per root-box area (a primitive's box lies inside the root box: every term below 2^40 and the sum of
    // up to 2^23 of them below 2^63)
    const double root_area = (double)ext[0] * ext[1] + (double)ext[1] * ext[2] + (double)ext[2] * ext[0];
    const double area_scale = root_area > 0. && root_area < 1e60 ? 1099511627776.0 / root_area : 0.;
    int log2n = 0;
    while ((2u << log2n) <= n_in) ++log2n;  // floor(log2 n)
    G.level_max = std::max(0, std::min(3 * GRID_BITS, log2n - split.level_slack));
    const uint32_t nb_in = (n_in + BT - 1) / BT;
    DTri* d_tris;
    float* prio;
    uint32_t *parts, *poff, *n_split;
    unsigned long long* total;
    unsigned long long* areas;
    B_TRY(S.get(&d_tris, counts_in[0])); B_TRY(S.get(&prio, n_in)); B_TRY(S.get(&parts, n_in)); B_TRY(S.get(&poff, n_in)); B_TRY(S.get(&n_split, 1)); B_TRY(S.get(&total, 1));
    B_TRY(S.get(&areas, 2));
    B_TRY(hipMemcpyAsync(d_tris, tris, (size_t)counts_in[0] * sizeof(DTri), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_split_priority, dim3(nb_in), dim3(BT), 0, stream, d_prims, n_in, d_tris, counts_in[0], G, prio);
    B_LAUNCHED(k_split_priority);
    // the scale D of the priorities: the largest one whose split counts floor(D * p) stay within the budget (bisection; the
    // sum is monotone in D). D_hi gives every triangle with a non-zero priority SPLIT_CAP splits.
    float p_max = 0.f;
    {
      size_t red_bytes = 0;
      float* d_max;
      B_TRY(S.get(&d_max, 1));
      B_TRY(rocprim::reduce(nullptr, red_bytes, prio, d_max, 0.f, (size_t)n_in, rocprim::maximum<float>(), stream));
      char* red_tmp;
      B_TRY(S.get(&red_tmp, red_bytes));
      B_TRY(rocprim::reduce(red_tmp, red_bytes, prio, d_max, 0.f, (size_t)n_in, rocprim::maximum<float>(), stream));
      B_TRY(hipMemcpyAsync(&p_max, d_max, 4, hipMemcpyDeviceToHost, stream));
      B_TRY(hipStreamSynchronize(stream));
    }
    if (p_max > 0.f) {
      auto total_at = [&](float D, unsigned long long& t) -> bool {
        if (hipMemsetAsync(total, 0, 8, stream) != hipSuccess) return false;
        hipLaunchKernelGGL(k_split_sum, dim3(nb_in), dim3(BT), 0, stream, prio, n_in, D, total);
        if (hipGetLastError() != hipSuccess) return false;
        if (hipMemcpyAsync(&t, total, 8, hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
        return hipStreamSynchronize(stream) == hipSuccess;
      };
      // priorities span (p_max * 2^-21, p_max] in practice (cube roots); below D_lo nobody splits
      double lo = 0.5 / p_max, hi = (double)(SPLIT_CAP + 1) / p_max * 4194304.0;
      unsigned long long t = 0;
      if (!total_at((float)hi, t)) { err = "device tree build: split budget search failed"; return false; }
      float D = (float)hi;
      if (t > budget) {
        for (int it = 0; it < 48; ++it) {
          const double mid = std::sqrt(lo * hi);
          if (!total_at((float)mid, t)) { err = "device tree build: split budget search failed"; return false; }
          if (t > budget) hi = mid; else lo = mid;
          if (hi / lo < 1.0005) break;
        }
        D = (float)lo;
      }
      B_TRY(hipMemsetAsync(n_split, 0, 4, stream));
      B_TRY(hipMemsetAsync(areas, 0, 16, stream));
      hipLaunchKernelGGL(k_split, dim3(nb_in), dim3(BT), 0, stream, d_prims, n_in, d_tris, counts_in[0], G, prio, D, parts, (const uint32_t*)nullptr, (SolBuildPrim*)nullptr,
                         (uint32_t*)nullptr, n_split, areas, area_scale);
      B_LAUNCHED(k_split);
      size_t sb = 0;
      B_TRY(rocprim::exclusive_scan(nullptr, sb, parts, poff, 0u, (size_t)n_in, rocprim::plus<uint32_t>(), stream));
      char* stmp;
      B_TRY(S.get(&stmp, sb));
      B_TRY(rocprim::exclusive_scan(stmp, sb, parts, poff, 0u, (size_t)n_in, rocprim::plus<uint32_t>(), stream));
      uint32_t last[2];
      B_TRY(hipMemcpyAsync(&last[0], poff + (n_in - 1), 4, hipMemcpyDeviceToHost, stream));
      B_TRY(hipMemcpyAsync(&last[1], parts + (n_in - 1), 4, hipMemcpyDeviceToHost, stream));
      B_TRY(hipMemcpyAsync(&out.split_triangles, n_split, 4, hipMemcpyDeviceToHost, stream));
      unsigned long long h_areas[2] = {0ull, 0ull};
      B_TRY(hipMemcpyAsync(h_areas, areas, 16, hipMemcpyDeviceToHost, stream));
      B_TRY(hipStreamSynchronize(stream));
      const uint32_t extra = last[0] + last[1];
      out.split_area_ratio = h_areas[0] > 0ull ? (float)((double)h_areas[1] / (double)h_areas[0]) : 1.f;
      const bool keep = out.split_area_ratio <= split.max_area_ratio;
      if (!keep) out.split_triangles = 0;
      if (keep && extra > 0 && (uint64_t)n_in + extra <= (SOL_WIDE_MAX_INDEX >> 1)) {
        SolBuildPrim* d_prims2;
        uint32_t* d_extra_of;
        B_TRY(S.get(&d_prims2, n_in + extra)); B_TRY(S.get(&d_extra_of, extra));
        hipLaunchKernelGGL(k_split, dim3(nb_in), dim3(BT), 0, stream, d_prims, n_in, d_tris, counts_in[0], G, prio, D, parts, (const uint32_t*)poff, d_prims2, d_extra_of, n_split, areas, area_scale);
        B_LAUNCHED(k_split);
        out.extra_of.resize(extra);
        B_TRY(hipMemcpyAsync(out.extra_of.data(), d_extra_of, (size_t)extra * 4, hipMemcpyDeviceToHost, stream));
        B_TRY(hipStreamSynchronize(stream));
        d_prims = d_prims2;
        n = n_in + extra;
        counts[0] = counts_in[0] + extra;
      }
    }
  }
  const uint32_t n_nodes = 2 * n - 1;
  const uint32_t nb = (n + BT - 1) / BT;
  unsigned long long *keys, *keys2;
  uint32_t *vals, *order, *cl_a, *cl_b, *nn, *out_node, *flag, *offset, *left, *right, *parent, *arrived, *counters, *leaf_refs, *new_index[3];
  float* nbox;
  Dp* dp;
  DWide* wides;
  Frontier *fr_a, *fr_b;
  B_TRY(S.get(&keys, n)); B_TRY(S.get(&keys2, n)); B_TRY(S.get(&vals, n)); B_TRY(S.get(&order, n));
  B_TRY(S.get(&cl_a, n)); B_TRY(S.get(&cl_b, n)); B_TRY(S.get(&nn, n)); B_TRY(S.get(&out_node, n)); B_TRY(S.get(&flag, n)); B_TRY(S.get(&offset, n));
  B_TRY(S.get(&left, n_nodes)); B_TRY(S.get(&right, n_nodes)); B_TRY(S.get(&parent, n_nodes)); B_TRY(S.get(&arrived, n_nodes));
  B_TRY(S.get(&nbox, (size_t)n_nodes * 6)); B_TRY(S.get(&dp, n_nodes)); B_TRY(S.get(&counters, 8)); B_TRY(S.get(&leaf_refs, n));
  B_TRY(S.get(&wides, n)); B_TRY(S.get(&fr_a, n)); B_TRY(S.get(&fr_b, n));
  for (int a = 0; a < 3; ++a) { B_TRY(S.get(&new_index[a], counts[a])); B_TRY(hipMemsetAsync(new_index[a], 0xFF, (size_t)(counts[a] ? counts[a] : 1) * 4, stream)); }
  B_TRY(hipMemsetAsync(arrived, 0, (size_t)n_nodes * 4, stream));
  if (split.want_boxes) {  // (sol_world_tree_check) every triangle reference's box, by expanded index
    std::vector<SolBuildPrim> hp(n);
    B_TRY(hipMemcpyAsync(hp.data(), d_prims, (size_t)n * sizeof(SolBuildPrim), hipMemcpyDeviceToHost, stream));
    B_TRY(hipStreamSynchronize(stream));
    const float inf = __builtin_huge_valf();
    out.ref_box.assign((size_t)counts[0] * 6, 0.f);
    for (uint32_t e = 0; e < counts[0]; ++e) { float* b = &out.ref_box[(size_t)e * 6]; b[0] = b[2] = b[4] = inf; b[1] = b[3] = b[5] = -inf; }
    for (const SolBuildPrim& q : hp)
      if (SOL_REF_KIND(q.ref) == SOL_REF_TRIANGLE && SOL_REF_INDEX(q.ref) < counts[0]) std::memcpy(&out.ref_box[(size_t)SOL_REF_INDEX(q.ref) * 6], q.box, 24);
  }
  dbg("pre-splitting done");
  // ---- 1. Morton order ----
  hipLaunchKernelGGL(k_morton, dim3(nb), dim3(BT), 0, stream, d_prims, n, root_box[0], root_box[2], root_box[4], inv[0], inv[1], inv[2], keys, vals);
  B_LAUNCHED(k_morton);
  size_t tmp_bytes = 0, scan_bytes = 0;
  B_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys2, vals, order, (size_t)n, 0, 63, stream));
  B_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flag, offset, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
  char* tmp;
  B_TRY(S.get(&tmp, tmp_bytes > scan_bytes ? tmp_bytes : scan_bytes));
  B_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, order, (size_t)n, 0, 63, stream));
  hipLaunchKernelGGL(k_leaves, dim3(nb), dim3(BT), 0, stream, d_prims, order, n, nbox, cl_a, parent);
  B_LAUNCHED(k_leaves);
  dbg("morton sort done");
  // ---- 2. PLOC ----
  const uint32_t init_counters[8] = {1u, 0u, 0u, 0u, 0u, 0u, 0u, n /* next free binary node */};
  B_TRY(hipMemcpyAsync(counters, init_counters, sizeof init_counters, hipMemcpyHostToDevice, stream));
  // node visits per ray against the radius (MI355X; C2 / C3 / C5; the probed host trees: 10.96 / 12.75 / 6.75): 8: 9.62 / 13.11 / 6.92,
  // 12: 10.91 / 12.94 / 7.67, 16: 10.85 / 12.98 / 6.93, 24: 10.30 / 13.06 / 7.77, 32: 9.90 / 13.08 / 7.40, 64: 10.97 / 13.73 / 7.24 - not
  // monotonic (greedy clustering); 16 is within 3 % of the host trees on the triangle scenes
  int radius = 16;
  if (ploc_radius > 0) radius = std::min(PLOC_R_MAX, ploc_radius);  // (experiment knob: SOL_PLOC_R)
  uint32_t cur_n = n, rounds = 0;
  uint32_t *cin = cl_a, *cout = cl_b;
  while (cur_n > 1) {
    const uint32_t g = (cur_n + BT - 1) / BT;
    hipLaunchKernelGGL(k_nn, dim3(g), dim3(BT), 0, stream, cin, cur_n, nbox, radius, nn);
    B_LAUNCHED(k_nn);
    hipLaunchKernelGGL(k_merge, dim3(g), dim3(BT), 0, stream, cin, cur_n, nn, nbox, left, right, parent, counters + 7, out_node, flag);
    B_LAUNCHED(k_merge);
    B_TRY(rocprim::exclusive_scan(tmp, scan_bytes, flag, offset, 0u, (size_t)cur_n, rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(k_compact, dim3(g), dim3(BT), 0, stream, out_node, flag, offset, cur_n, cout);
    B_LAUNCHED(k_compact);
    uint32_t last[2];
    B_TRY(hipMemcpyAsync(&last[0], offset + (cur_n - 1), 4, hipMemcpyDeviceToHost, stream));
    B_TRY(hipMemcpyAsync(&last[1], flag + (cur_n - 1), 4, hipMemcpyDeviceToHost, stream));
    B_TRY(hipStreamSynchronize(stream));
    const uint32_t next_n = last[0] + last[1];
    if (next_n >= cur_n || next_n == 0 || ++rounds > 4096) { err = "device tree build: clustering made no progress"; return false; }
    cur_n = next_n;
    uint32_t* t = cin; cin = cout; cout = t;
  }
  uint32_t root_node = 0;
  B_TRY(hipMemcpyAsync(&root_node, cin, 4, hipMemcpyDeviceToHost, stream));
  dbg("ploc done");
  // ---- 2b. reinsertion rounds ----
  out.reinsertion_moves = 0;
  out.area_before = out.area_after = 0.;
  if (split.reinsertion_rounds > 0 && n >= 8) {
    unsigned long long *lock, *in_key;
    uint32_t *best_out, *best_pivot, *n_moved;
    float* best_gain;
    double* cost;
    uint8_t* ok;
    B_TRY(S.get(&in_key, n_nodes)); B_TRY(S.get(&ok, n_nodes));
    B_TRY(S.get(&lock, n_nodes)); B_TRY(S.get(&best_out, n_nodes)); B_TRY(S.get(&best_pivot, n_nodes)); B_TRY(S.get(&best_gain, n_nodes)); B_TRY(S.get(&n_moved, 1)); B_TRY(S.get(&cost, 1));
    const uint32_t gn = (n_nodes + BT - 1) / BT;
    uint32_t* dirty;
    B_TRY(S.get(&dirty, n_nodes));
    B_TRY(hipMemsetAsync(dirty, 0, (size_t)n_nodes * 4, stream));
    // (the clustering leaves every box valid: k_merge computes a new node's box from its children's)
    auto area = [&](double& total) -> bool {
      if (hipMemsetAsync(cost, 0, 8, stream) != hipSuccess) return false;
      hipLaunchKernelGGL(k_area_sum, dim3(gn), dim3(BT), 0, stream, n_nodes, n, nbox, cost);
      if (hipGetLastError() != hipSuccess) return false;
      if (hipMemcpyAsync(&total, cost, 8, hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
      return hipStreamSynchronize(stream) == hipSuccess;
    };
    auto refit_moved = [&]() -> bool {  // boxes on the paths the round's moves touched (best_pivot: where each moved node left from)
      hipLaunchKernelGGL(k_refit_mark, dim3(gn), dim3(BT), 0, stream, n_nodes, parent, ok, best_pivot, dirty);
      hipLaunchKernelGGL(k_refit_prepare, dim3(gn), dim3(BT), 0, stream, n_nodes, n, left, right, dirty, arrived);
      hipLaunchKernelGGL(k_refit_starts, dim3(gn), dim3(BT), 0, stream, n_nodes, n, dirty, arrived);
      hipLaunchKernelGGL(k_refit_walk, dim3(gn), dim3(BT), 0, stream, n_nodes, n, parent, left, right, nbox, dirty, arrived);
      return hipGetLastError() == hipSuccess;
    };
    if (!area(out.area_before)) { err = "device tree build: area sum failed"; return false; }
    out.area_after = out.area_before;
    for (int round = 0; round < split.reinsertion_rounds; ++round) {
      const uint32_t stride = (uint32_t)std::max(1, split.reinsertion_stride);
      hipLaunchKernelGGL(k_reins_find, dim3(gn), dim3(BT), 0, stream, n_nodes, n, parent, left, right, nbox, (uint32_t)round % stride, stride, best_out, best_pivot, best_gain);
      B_LAUNCHED(k_reins_find);
      B_TRY(hipMemsetAsync(lock, 0, (size_t)n_nodes * 8, stream));
      B_TRY(hipMemsetAsync(n_moved, 0, 4, stream));
      hipLaunchKernelGGL(k_reins_lock, dim3(gn), dim3(BT), 0, stream, n_nodes, parent, left, right, best_out, best_gain, lock, in_key);
      B_LAUNCHED(k_reins_lock);
      hipLaunchKernelGGL(k_reins_verify, dim3(gn), dim3(BT), 0, stream, n_nodes, parent, left, right, best_out, best_pivot, best_gain, lock, in_key, ok);
      B_LAUNCHED(k_reins_verify);
      hipLaunchKernelGGL(k_reins_apply, dim3(gn), dim3(BT), 0, stream, n_nodes, parent, left, right, best_out, ok, n_moved, best_pivot);  // (best_pivot is free once the move is verified: it takes left_from)
      B_LAUNCHED(k_reins_apply);
      uint32_t moved = 0;
      B_TRY(hipMemcpyAsync(&moved, n_moved, 4, hipMemcpyDeviceToHost, stream));
      if (!refit_moved()) { err = "device tree build: refit failed"; return false; }
      B_TRY(hipStreamSynchronize(stream));
      out.reinsertion_moves += moved;
      if (split.verbose) {
        double total = 0.;
        if (!area(total)) { err = "device tree build: area sum failed"; return false; }
        std::fprintf(stderr, "[solstrale] reinsertion round %d: %u moves, summed inner area %.6g (start %.6g)\n", round, moved, total, out.area_before);
      }
      if (moved == 0 && stride == 1) break;
    }
    if (!area(out.area_after)) { err = "device tree build: area sum failed"; return false; }
    B_TRY(hipMemsetAsync(arrived, 0, (size_t)n_nodes * 4, stream));  // (k_collapse_cost counts arrivals again)
    B_TRY(hipMemsetAsync(n_moved, 0, 4, stream));
    uint32_t* bad;
    B_TRY(S.get(&bad, 3));
    B_TRY(hipMemsetAsync(bad, 0, 12, stream));
    hipLaunchKernelGGL(k_validate, dim3(gn), dim3(BT), 0, stream, n_nodes, n, root_node, parent, left, right, nbox, bad);
    B_LAUNCHED(k_validate);
    uint32_t h_bad[3] = {0, 0, 0};
    B_TRY(hipMemcpyAsync(h_bad, bad, 12, hipMemcpyDeviceToHost, stream));
    B_TRY(hipStreamSynchronize(stream));
    if (h_bad[0] || h_bad[1] || h_bad[2]) {
      err = "device tree build: reinsertion left " + std::to_string(h_bad[0]) + " broken links, " + std::to_string(h_bad[1]) + " wrong boxes and " + std::to_string(h_bad[2]) + " leaves that do not reach the root";
      return false;
    }
  }
  dbg("reinsertion done");
  // ---- 3. collapse costs ----
  hipLaunchKernelGGL(k_collapse_cost, dim3(nb), dim3(BT), 0, stream, n, parent, left, right, nbox, arrived, dp, split.node_cost);
  B_LAUNCHED(k_collapse_cost);
  out.collapse_cost = 0.f;
  if (root_node >= n) B_TRY(hipMemcpyAsync(&out.collapse_cost, &dp[root_node].c[1], 4, hipMemcpyDeviceToHost, stream));
  B_TRY(hipStreamSynchronize(stream));
  dbg("collapse costs done");
  const bool verbose = split.verbose;
  const uint32_t counts0 = counts[0], counts1 = counts[1], counts2 = counts[2];
  uint32_t* const ni0 = new_index[0]; uint32_t* const ni1 = new_index[1]; uint32_t* const ni2 = new_index[2];
  SolDeviceBuild* b = new SolDeviceBuild;
  b->scratch = Sp;
  // (everything the second half needs, by value: device pointers into the handle's scratch memory and a few numbers)
  b->emit = [=](SolDeviceTree& out, std::string& err) -> bool {
  const uint32_t counts[3] = {counts0, counts1, counts2};
  uint32_t* const new_index[3] = {ni0, ni1, ni2};
  auto dbg = [&](const char* what) { if (verbose) { hipStreamSynchronize(stream); std::fprintf(stderr, "[solstrale]   build: %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dbg0).count()); } };
  // ---- 4. emission, level by level ----
  EmitParams P;
  P.prims = d_prims; P.order = order; P.left = left; P.right = right; P.nbox = nbox; P.dp = dp; P.n_leaves = n; P.pad = pad; P.emin = emin;
  P.wides = wides; P.leaf_refs = leaf_refs; P.counters = counters;
  for (int a = 0; a < 3; ++a) P.new_index[a] = new_index[a];
  const Frontier f0{root_node, 0u};
  B_TRY(hipMemcpyAsync(fr_a, &f0, sizeof f0, hipMemcpyHostToDevice, stream));
  uint32_t n_front = 1, depth = 0;
  Frontier *fin = fr_a, *fout = fr_b;
  while (n_front > 0) {
    if (++depth > 256) { err = "device tree build: tree deeper than 256 wide levels"; return false; }
    B_TRY(hipMemsetAsync(counters + 1, 0, 4, stream));
    hipLaunchKernelGGL(k_emit, dim3((n_front + 63) / 64), dim3(64), 0, stream, P, fin, n_front, fout);
    B_LAUNCHED(k_emit);
    B_TRY(hipMemcpyAsync(&n_front, counters + 1, 4, hipMemcpyDeviceToHost, stream));
    B_TRY(hipStreamSynchronize(stream));
    Frontier* t = fin; fin = fout; fout = t;
  }
  for (int a = 0; a < 3; ++a)
    if (counts[a]) {
      hipLaunchKernelGGL(k_rest, dim3((counts[a] + BT - 1) / BT), dim3(BT), 0, stream, new_index[a], counts[a], counters + 3 + a);
      B_LAUNCHED(k_rest);
    }
  uint32_t fin_counters[8];
  B_TRY(hipMemcpyAsync(fin_counters, counters, sizeof fin_counters, hipMemcpyDeviceToHost, stream));
  B_TRY(hipStreamSynchronize(stream));
  if (fin_counters[6]) { err = "device tree build: " + std::string((fin_counters[6] & 4u) ? "more than 2^24 nodes or primitives" : (fin_counters[6] & 1u) ? "exponent outside the 5-bit range" : "a child box could not be quantised"); return false; }
  for (int a = 0; a < 3; ++a)
    if (fin_counters[3 + a] != counts[a]) { err = "device tree build: primitive permutation is not a permutation"; return false; }
  // ---- results to the host-side layout record (small: 64 B per node; the primitive arrays are permuted by the caller) ----
  out.nodes.resize(fin_counters[0]);
  out.leaf_refs.resize(fin_counters[2]);
  out.depth = depth;
  out.rounds = rounds;
  B_TRY(hipMemcpyAsync(out.nodes.data(), wides, (size_t)fin_counters[0] * sizeof(DWide), hipMemcpyDeviceToHost, stream));
  if (fin_counters[2]) B_TRY(hipMemcpyAsync(out.leaf_refs.data(), leaf_refs, (size_t)fin_counters[2] * 4, hipMemcpyDeviceToHost, stream));
  for (int a = 0; a < 3; ++a) {
    out.new_of_old[a].resize(counts[a]);
    if (counts[a]) B_TRY(hipMemcpyAsync(out.new_of_old[a].data(), new_index[a], (size_t)counts[a] * 4, hipMemcpyDeviceToHost, stream));
  }
  B_TRY(hipStreamSynchronize(stream));
  dbg("emitted and downloaded");
  return true;
  };  // (b->emit)
  *handle = b;
  return true;
}
