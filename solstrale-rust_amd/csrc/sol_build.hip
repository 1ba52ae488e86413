// sol_build.hip -- the world's tree built ON THE GPU (SolCreateOptions.world_tree = SOL_TREE_DEVICE).
//
// Replaces, for the world search, the host builders of sol_tree.h (themselves a results-neutral replacement of the reference's
// centroid-median build, src/hittable/bvh.rs:84-162): the closest hit does not depend on the tree, so any
// conservative tree over the same primitives may be walked. Input: the world's primitives as (reference, padded fp32 box) in
// any order. Output: the device form of the 7-wide tree (DWide, sol_types.h) + the permutations of the primitive arrays, in
// the same WideLayout record the host path produces, so everything downstream (upload, checks, kernels) is shared.
//
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

#include <algorithm>
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
constexpr float NODE_COST = 2.5f, PRIM_COST = 1.0f;  // as WideBuilder (sol_tree.h)
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

// Collapse costs, bottom-up (sol_tree.h, WideBuilder::dp_compute): C(m, i) = cheapest representation of binary sub-tree m in at
// most i child slots of its parent.
__global__ void __launch_bounds__(BT) k_collapse_cost(uint32_t n_leaves, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ left,
                                                       const uint32_t* __restrict__ right, const float* __restrict__ nbox, uint32_t* __restrict__ arrived,
                                                       Dp* __restrict__ dp) {
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
    __threadfence();
    if (atomicAdd(&arrived[m], 1u) == 0u) return;  // the sibling's thread will do this node
    __threadfence();
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

bool sol_build_world_tree_device(const SolBuildPrim* prims, uint32_t n, const float root_box[6], float pad, uint32_t emin, const uint32_t counts[3],
                                 int ploc_radius, hipStream_t stream, SolDeviceTree& out, std::string& err) {
  if (n == 0 || n > (SOL_WIDE_MAX_INDEX >> 1)) { err = "device tree build: primitive count out of range"; return false; }
  Scratch S;
  const uint32_t n_nodes = 2 * n - 1;
  const uint32_t nb = (n + BT - 1) / BT;
  SolBuildPrim* d_prims;
  unsigned long long *keys, *keys2;
  uint32_t *vals, *order, *cl_a, *cl_b, *nn, *out_node, *flag, *offset, *left, *right, *parent, *arrived, *counters, *leaf_refs, *new_index[3];
  float* nbox;
  Dp* dp;
  DWide* wides;
  Frontier *fr_a, *fr_b;
  B_TRY(S.get(&d_prims, n)); B_TRY(S.get(&keys, n)); B_TRY(S.get(&keys2, n)); B_TRY(S.get(&vals, n)); B_TRY(S.get(&order, n));
  B_TRY(S.get(&cl_a, n)); B_TRY(S.get(&cl_b, n)); B_TRY(S.get(&nn, n)); B_TRY(S.get(&out_node, n)); B_TRY(S.get(&flag, n)); B_TRY(S.get(&offset, n));
  B_TRY(S.get(&left, n_nodes)); B_TRY(S.get(&right, n_nodes)); B_TRY(S.get(&parent, n_nodes)); B_TRY(S.get(&arrived, n_nodes));
  B_TRY(S.get(&nbox, (size_t)n_nodes * 6)); B_TRY(S.get(&dp, n_nodes)); B_TRY(S.get(&counters, 8)); B_TRY(S.get(&leaf_refs, n));
  B_TRY(S.get(&wides, n)); B_TRY(S.get(&fr_a, n)); B_TRY(S.get(&fr_b, n));
  for (int a = 0; a < 3; ++a) { B_TRY(S.get(&new_index[a], counts[a])); B_TRY(hipMemsetAsync(new_index[a], 0xFF, (size_t)(counts[a] ? counts[a] : 1) * 4, stream)); }
  B_TRY(hipMemcpyAsync(d_prims, prims, (size_t)n * sizeof(SolBuildPrim), hipMemcpyHostToDevice, stream));
  B_TRY(hipMemsetAsync(arrived, 0, (size_t)n_nodes * 4, stream));
  // ---- 1. Morton order ----
  float ext[3], inv[3];
  for (int a = 0; a < 3; ++a) {
    ext[a] = root_box[2 * a + 1] - root_box[2 * a];
    inv[a] = (ext[a] > 0.f && ext[a] < 1e30f) ? 2097152.0f / ext[a] : 0.f;
  }
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
  // ---- 3. collapse costs ----
  hipLaunchKernelGGL(k_collapse_cost, dim3(nb), dim3(BT), 0, stream, n, parent, left, right, nbox, arrived, dp);
  B_LAUNCHED(k_collapse_cost);
  B_TRY(hipStreamSynchronize(stream));
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
  return true;
}
