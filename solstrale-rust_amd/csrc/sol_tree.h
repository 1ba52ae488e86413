// sol_tree.h -- host-side builders of the device's trees (included by sol_api.cpp only; nothing here runs per sample).
//
//   TreeBuilder  the reference-shaped binary tree (SolBvhNode, own box per node) -> DNode (child boxes in the parent): walked by
//                the boundary searches of ConstantMedium;
//   SahBuilder   a binned surface-area-heuristic rebuild of the WORLD's binary tree over the same primitives;
//   WideBuilder  collapse of a binary tree into 7-wide nodes with 8-bit quantised child boxes, in explicit form (XWide: one
//                reference per child);
//   WideLayout   the device form of that tree (DWide: 64 bytes, implicit child addresses): children of a node consecutive,
//                primitive arrays permuted so that the primitives of a node are consecutive - the tree every
//                world.hit(ray, [0.001, inf)) walks.
// Why the device may walk another tree than the reference's: DESIGN.md 4, "tree independence". sol_world_tree_check
// (sol_api.cpp) verifies the structure of what these builders produce.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/solstrale_hip.h"
#include "sol_types.h"

namespace {

struct Box {
  float v[6];
};
const float F_INF = std::numeric_limits<float>::infinity();
Box empty_box() { return Box{{F_INF, -F_INF, F_INF, -F_INF, F_INF, -F_INF}}; }
// fp32 box contract (DESIGN.md "fp32 arithmetic contract"): cast, then pad outward by pad = S * 2^-20 (box_pad_for) where S is
// the largest finite |coordinate| of the world's box and the camera origin. PAD_DELTA (1e-4, src/geo/mod.rs:11) is sized
// for f64; in fp32 a flat box seen from a distant origin collapses ((554.99994+800) == (555.00006+800) == 1355.0f) and
// the slab test t_min < t_max fails. With the pad every fp32 slab test is conservative.
Box cast_box(const SolAabb& b, float pad) {
  Box r;
  for (int i = 0; i < 6; i += 2) {
    r.v[i] = (float)b.v[i] - pad;
    r.v[i + 1] = (float)b.v[i + 1] + pad;
  }
  return r;
}
float box_pad_for(const SolSceneDesc& d) {
  float S = 0.0f;
  auto take = [&](double v) { float a = std::fabs((float)v); if (std::isfinite(a) && a > S) S = a; };
  const uint32_t k = SOL_REF_KIND(d.root), i = SOL_REF_INDEX(d.root);
  const SolAabb* b = nullptr;
  if (k == SOL_REF_NODE && i < d.n_nodes) b = &d.nodes[i].bbox;
  else if (k == SOL_REF_SPHERE && i < d.n_spheres) b = &d.spheres[i].bbox;
  else if (k == SOL_REF_QUAD && i < d.n_quads) b = &d.quads[i].bbox;
  else if (k == SOL_REF_TRIANGLE && i < d.n_triangles) b = &d.triangles[i].bbox;
  else if (k == SOL_REF_MEDIUM && i < d.n_mediums) b = &d.mediums[i].bbox;
  if (b) for (int j = 0; j < 6; ++j) take(b->v[j]);
  for (int j = 0; j < 3; ++j) take(d.camera.origin[j]);
  // (scenes with needle triangles: SOL_NEEDLE_PAD (4) times the thin pad, so that a triangle hit the consistency rule accepts - within 0.8 pads of the
  // triangle - lies inside every box around its part of the triangle with a fifth of a pad to spare: include/solstrale_hip.h)
  return S * ((sol_scene_has_needles(&d) ? SOL_NEEDLE_PAD : 1.0f) / 1048576.0f);
}

// Converts the reference-shaped tree (own box per node) into device nodes (child boxes in the parent).
struct TreeBuilder {
  const SolSceneDesc& d;
  std::vector<DNode> nodes;
  std::vector<int32_t> dev_index;  // desc node -> device node (-1 not yet / collapsed)
  std::vector<uint8_t> on_path;
  uint32_t max_depth = 0;
  std::string error;

  const float pad;  // the fp32 box pad of this scene (box_pad_for)
  TreeBuilder(const SolSceneDesc& desc, float box_pad) : d(desc), dev_index(desc.n_nodes, -1), on_path(desc.n_nodes, 0), pad(box_pad) {}

  bool prim_box(uint32_t ref, Box& box) {
    uint32_t k = SOL_REF_KIND(ref), i = SOL_REF_INDEX(ref);
    switch (k) {
      case SOL_REF_SPHERE: if (i >= d.n_spheres) return false; box = cast_box(d.spheres[i].bbox, pad); return true;
      case SOL_REF_QUAD: if (i >= d.n_quads) return false; box = cast_box(d.quads[i].bbox, pad); return true;
      case SOL_REF_TRIANGLE: if (i >= d.n_triangles) return false; box = cast_box(d.triangles[i].bbox, pad); return true;
      case SOL_REF_MEDIUM: if (i >= d.n_mediums) return false; box = cast_box(d.mediums[i].bbox, pad); return true;
    }
    return false;
  }

  // Returns the device reference of `ref` and the box the parent must test for it.
  bool resolve(uint32_t ref, uint32_t depth, uint32_t& out_ref, Box& out_box) {
    uint32_t k = SOL_REF_KIND(ref), i = SOL_REF_INDEX(ref);
    if (k == SOL_REF_NONE) { out_ref = SOL_MAKE_REF(SOL_REF_NONE, 0); out_box = empty_box(); return true; }
    if (k != SOL_REF_NODE) {
      if (!prim_box(ref, out_box)) { error = "primitive reference out of range"; return false; }
      out_ref = ref;
      return true;
    }
    if (i >= d.n_nodes) { error = "node reference out of range"; return false; }
    if (on_path[i]) { error = "cycle in BVH"; return false; }
    if (depth > 4000) { error = "BVH nesting deeper than 4000"; return false; }
    const SolBvhNode& n = d.nodes[i];
    const uint32_t lk = SOL_REF_KIND(n.left), rk = SOL_REF_KIND(n.right);
    if (rk == SOL_REF_NONE && lk != SOL_REF_NONE && lk != SOL_REF_NODE) {
      // `new_bvh` of a single primitive (bvh.rs:85-90): Bvh{Leaf(a), None, box(a)}. The parent tests the node's box and
      // goes straight to the primitive.
      Box pb;
      if (!prim_box(n.left, pb)) { error = "primitive reference out of range"; return false; }
      out_ref = n.left;
      out_box = cast_box(n.bbox, pad);
      return true;
    }
    if (dev_index[i] >= 0) {  // shared sub-tree
      out_ref = SOL_MAKE_REF(SOL_REF_NODE, (uint32_t)dev_index[i]);
      out_box = cast_box(n.bbox, pad);
      return true;
    }
    const uint32_t di = (uint32_t)nodes.size();
    if (di >= 0x0FFFFFFFu) { error = "too many nodes"; return false; }
    dev_index[i] = (int32_t)di;
    nodes.push_back(DNode{});
    on_path[i] = 1;
    if (depth + 1 > max_depth) max_depth = depth + 1;
    uint32_t lr, rr;
    Box lb, rb;
    if (!resolve(n.left, depth + 1, lr, lb) || !resolve(n.right, depth + 1, rr, rb)) return false;
    on_path[i] = 0;
    DNode& dn = nodes[di];
    dn.lxmin = lb.v[0]; dn.lxmax = lb.v[1]; dn.lymin = lb.v[2]; dn.lymax = lb.v[3]; dn.lzmin = lb.v[4]; dn.lzmax = lb.v[5];
    dn.rxmin = rb.v[0]; dn.rxmax = rb.v[1]; dn.rymin = rb.v[2]; dn.rymax = rb.v[3]; dn.rzmin = rb.v[4]; dn.rzmax = rb.v[5];
    dn.left = lr; dn.right = rr; dn.pad1 = 0;
    // flags: bit0 / bit1 = the left / right box is a direct leaf's own box, which the reference never tests (sol_trace.h)
    dn.pad0 = ((lk != SOL_REF_NONE && lk != SOL_REF_NODE) ? 1u : 0u) | ((rk != SOL_REF_NONE && rk != SOL_REF_NODE) ? 2u : 0u);
    out_ref = SOL_MAKE_REF(SOL_REF_NODE, di);
    out_box = cast_box(n.bbox, pad);
    return true;
  }
};

// Explicit form of a wide node (host only): grid origin, biased exponents of the three scales, the quantised planes laid
// out as in DWide (slot 7 unused), one reference per slot (SOL_REF_WIDE = index into the XWide array, a primitive, or NONE).
struct XWide {
  float o[3];
  uint32_t e[3];
  uint32_t q[12];
  uint32_t ref[8];
};

// Collapses the binary device tree into 7-wide nodes with 8-bit quantised child boxes (XWide). Pure layout:
// every decoded child box CONTAINS the child's padded fp32 box (checked with the device's own decode arithmetic), so the
// wide tree culls no ray that the binary tree would not; closest hits (t, tie rule on dfs_index) are identical.
struct WideBuilder {
  static constexpr int MAXC = SOL_WIDE_CHILDREN;  // children per node (slots 0 .. MAXC-1 of the eight octant slots)
  const std::vector<DNode>& bin;
  std::vector<XWide> out;
  // Exponents are stored in 5 bits relative to `emin` (DWide::meta): every scale of the tree lies in 2^[emin, emin + 31].
  // set_exponent_range() derives emin from the root's extent (no box below is larger); smaller nodes than 2^emin * 255 use the
  // coarser grid 2^emin (still containing, only less tight: 31 binary orders below the scene's size).
  uint32_t emin = 1;
  bool range_error = false;
  static uint32_t exponent_min(const Box& root_box, float pad) {
    float ext = 0.f;
    for (int a = 0; a < 3; ++a) {
      const float e = root_box.v[2 * a + 1] - root_box.v[2 * a];
      if (std::isfinite(e) && e > ext) ext = e;
    }
    int ex = -126;
    if (ext > 0.f) std::frexp((ext + 8.f * pad) / 255.0f, &ex);
    const int emax = std::min(254, std::max(1, ex + 127 + 1));  // one order of headroom for the pads added per level
    return (uint32_t)std::max(1, emax - 31);
  }
  void set_exponent_range(const Box& root_box) { emin = exponent_min(root_box, pad); }
  uint32_t max_depth = 0;
  // surface-area estimate of a random ray's work: summed box areas of the children that are wide nodes / primitives
  // (a child is visited with probability ~ its area / the root's area)
  bool slot_by_assignment = true;  // false (SOL_SLOTS=octant): children by the octant of their centre alone
  double inner_area = 0., leaf_area = 0.;
  double cost() const { return 2.5 * inner_area + leaf_area; }  // a wide-node visit costs ~2.5 primitive tests (instructions)
  struct Child { uint32_t ref; Box box; };

  const float pad;  // the fp32 box pad of this scene
  WideBuilder(const std::vector<DNode>& b, float box_pad) : bin(b), pad(box_pad) {}

  static Box lbox(const DNode& n) { return Box{{n.lxmin, n.lxmax, n.lymin, n.lymax, n.lzmin, n.lzmax}}; }
  static Box rbox(const DNode& n) { return Box{{n.rxmin, n.rxmax, n.rymin, n.rymax, n.rzmin, n.rzmax}}; }
  static float area(const Box& b) {
    float dx = b.v[1] - b.v[0], dy = b.v[3] - b.v[2], dz = b.v[5] - b.v[4];
    if (!(dx >= 0.f && dy >= 0.f && dz >= 0.f)) return 0.f;
    // an unbounded box (a primitive whose fp32 coordinates overflow) counts as a very large one: with an infinite area every
    // comparison of the collapse's cost table is false and whole sub-trees would silently drop out of the wide tree
    dx = std::min(dx, 1e15f); dy = std::min(dy, 1e15f); dz = std::min(dz, 1e15f);
    return dx * dy + dy * dz + dz * dx;
  }
  static float decode(float origin, uint32_t q, float scale) { return origin + (float)q * scale; }  // == device decode

  // ---- which binary sub-trees become the (up to eight) children of a wide node ----
  // GREEDY: open the inner child with the largest surface until eight children (or only leaves) remain.
  // DP (Ylitie, Karras, Laine 2017, "Efficient incoherent ray traversal on GPUs through compressed wide BVHs", sec. 4.1):
  //   the collapse of minimal surface-area cost. C(m, i) = cheapest way to represent binary sub-tree m with at most i child
  //   slots of its parent: either as ONE wide node (its area x NODE_COST, its own eight slots distributed over its two
  //   children) or by handing k slots to its left and i - k to its right sub-tree. The greedy rule leaves the many small
  //   sub-trees near the leaves as wide nodes with two or three children: C3 has 92 075 nodes of 3.85 children on average
  //   under it and 51 567 of 6.09 under the DP; node visits per ray 12.9 -> 11.9 (C3), 9.8 -> 9.5 (C2), time -0.5 .. -3 %.
  bool dp_collapse = true;   // SOL_COLLAPSE=greedy selects the other rule (A/B)
  double NODE_COST = 2.5;    // (experiment knob: SOL_NODE_COST)
  static constexpr double PRIM_COST = 1.0;
  struct Dp { double c[9]; uint8_t eff[9], split[9]; bool done = false; };
  std::vector<Dp> dp;
  static Box unite(const Box& a, const Box& b) {
    Box r = a;
    for (int k = 0; k < 3; ++k) { r.v[2 * k] = std::min(r.v[2 * k], b.v[2 * k]); r.v[2 * k + 1] = std::max(r.v[2 * k + 1], b.v[2 * k + 1]); }
    return r;
  }
  double dp_t(uint32_t ref, const Box& box, int i) {
    if (SOL_REF_KIND(ref) == SOL_REF_NODE) { dp_compute(SOL_REF_INDEX(ref)); return dp[SOL_REF_INDEX(ref)].c[i]; }
    return PRIM_COST * (double)area(box);
  }
  void dp_compute(uint32_t m) {
    if (dp.size() != bin.size()) dp.assign(bin.size(), Dp{});
    Dp& d = dp[m];
    if (d.done) return;
    d.done = true;  // (a cycle cannot occur: TreeBuilder rejects them)
    const DNode& n = bin[m];
    const bool hl = SOL_REF_KIND(n.left) != SOL_REF_NONE, hr = SOL_REF_KIND(n.right) != SOL_REF_NONE;
    for (int i = 0; i <= 8; ++i) { d.c[i] = 0.; d.eff[i] = 0; d.split[i] = 0; }
    if (!hl && !hr) return;
    if (hl != hr) {  // a single child: the node vanishes (eff 0 = pass through)
      for (int i = 1; i <= MAXC; ++i) d.c[i] = hl ? dp_t(n.left, lbox(n), i) : dp_t(n.right, rbox(n), i);
      dp[m] = d;
      return;
    }
    const Box bl = lbox(n), br = rbox(n);
    double tl[9], tr[9], dist[9];
    for (int i = 1; i <= MAXC; ++i) { tl[i] = dp_t(n.left, bl, i); tr[i] = dp_t(n.right, br, i); }
    Dp& e = dp[m];  // (dp may have been re-allocated? no: sized once above; re-take the reference after the recursion anyway)
    for (int j = 2; j <= MAXC; ++j) {
      dist[j] = std::numeric_limits<double>::infinity();
      for (int k = 1; k < j; ++k)
        if (tl[k] + tr[j - k] < dist[j]) { dist[j] = tl[k] + tr[j - k]; e.split[j] = (uint8_t)k; }
    }
    e.c[1] = NODE_COST * (double)area(unite(bl, br)) + dist[MAXC];
    e.eff[1] = 1;
    for (int i = 2; i <= MAXC; ++i) {
      if (dist[i] < e.c[i - 1]) { e.c[i] = dist[i]; e.eff[i] = (uint8_t)i; }
      else { e.c[i] = e.c[i - 1]; e.eff[i] = e.eff[i - 1]; }
    }
  }
  void dp_gather(uint32_t ref, const Box& box, int i, std::vector<Child>& c) {
    if (SOL_REF_KIND(ref) == SOL_REF_NONE) return;
    if (SOL_REF_KIND(ref) != SOL_REF_NODE) { c.push_back(Child{ref, box}); return; }
    const uint32_t m = SOL_REF_INDEX(ref);
    dp_compute(m);
    const DNode& n = bin[m];
    const int j = dp[m].eff[i];
    if (j == 0) {  // pass through to the only child
      if (SOL_REF_KIND(n.left) != SOL_REF_NONE) dp_gather(n.left, lbox(n), i, c); else dp_gather(n.right, rbox(n), i, c);
    } else if (j == 1) {
      c.push_back(Child{ref, box});  // a wide node of its own
    } else {
      const int k = dp[m].split[j];
      dp_gather(n.left, lbox(n), k, c);
      dp_gather(n.right, rbox(n), j - k, c);
    }
  }

  // Returns the reference to use for binary node `ni`: a wide node, or - when the node has a single child - that child.
  uint32_t build(uint32_t ni, uint32_t depth) {
    std::vector<Child> c;
    auto add = [&](uint32_t ref, const Box& b) { if (SOL_REF_KIND(ref) != SOL_REF_NONE) c.push_back(Child{ref, b}); };
    if (dp_collapse) {
      dp_compute(ni);
      const DNode& n = bin[ni];
      const bool hl = SOL_REF_KIND(n.left) != SOL_REF_NONE, hr = SOL_REF_KIND(n.right) != SOL_REF_NONE;
      if (hl && hr) { const int k = dp[ni].split[MAXC]; dp_gather(n.left, lbox(n), k, c); dp_gather(n.right, rbox(n), MAXC - k, c); }
      else { add(n.left, lbox(n)); add(n.right, rbox(n)); }
    } else {
    add(bin[ni].left, lbox(bin[ni]));
    add(bin[ni].right, rbox(bin[ni]));
    while ((int)c.size() < MAXC) {  // open the inner child with the largest surface until MAXC children (or only leaves) remain
      int best = -1;
      float best_a = -1.f;
      for (size_t i = 0; i < c.size(); ++i)
        if (SOL_REF_KIND(c[i].ref) == SOL_REF_NODE) {
          const DNode& n = bin[SOL_REF_INDEX(c[i].ref)];
          int kids = (SOL_REF_KIND(n.left) != SOL_REF_NONE) + (SOL_REF_KIND(n.right) != SOL_REF_NONE);
          if ((int)c.size() - 1 + kids > MAXC) continue;
          float a = area(c[i].box);
          if (a > best_a) { best_a = a; best = (int)i; }
        }
      if (best < 0) break;
      const DNode n = bin[SOL_REF_INDEX(c[best].ref)];
      c.erase(c.begin() + best);
      add(n.left, lbox(n));
      add(n.right, rbox(n));
    }
    }
    const uint32_t wi = (uint32_t)out.size();
    out.push_back(XWide{});
    if (depth + 1 > max_depth) max_depth = depth + 1;
    for (auto& ch : c) (SOL_REF_KIND(ch.ref) == SOL_REF_NODE ? inner_area : leaf_area) += (double)area(ch.box);
    // node box and quantisation grid
    float lo[3] = {F_INF, F_INF, F_INF}, hi[3] = {-F_INF, -F_INF, -F_INF};
    for (auto& ch : c)
      for (int a = 0; a < 3; ++a) {
        if (std::isfinite(ch.box.v[2 * a])) lo[a] = std::min(lo[a], ch.box.v[2 * a] - 2.0f * pad);  // (the planes' two extra pads, below)
        if (std::isfinite(ch.box.v[2 * a + 1])) hi[a] = std::max(hi[a], ch.box.v[2 * a + 1] + 2.0f * pad);
      }
    uint32_t eb[3];
    float scale[3];
    for (int a = 0; a < 3; ++a) {
      if (!(hi[a] >= lo[a])) { lo[a] = 0.f; hi[a] = 0.f; }
      int e = 1;
      float ext = hi[a] - lo[a];
      if (ext > 0.f && std::isfinite(ext)) {
        int ex;
        std::frexp(ext / 255.0f, &ex);  // ext/255 = m * 2^ex, m in [0.5,1)  ->  2^ex >= ext/255
        e = std::min(254, std::max(1, ex + 127));
      }
      if (e > (int)emin + 31) range_error = true;  // (a box larger than the root's: cannot happen)
      e = std::max((int)emin, std::min((int)emin + 31, e));
      eb[a] = (uint32_t)e;
      uint32_t bits = eb[a] << 23;
      std::memcpy(&scale[a], &bits, 4);
    }
    // slots by octant of the child's centre (x << 2 | y << 1 | z), nearest free slot on conflict
    float ctr[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
    int slot_of[8];
    bool used[8] = {false, false, false, false, false, false, false, false};
    if (slot_by_assignment) {
      // The device visits the hit children in the order slot ^ ray_octant, i.e. slot s is "far" along direction
      // (+-1, +-1, +-1)_s. Give child i slot s so that the summed projections of the child centres on their slots'
      // directions is largest (an 8x8 assignment problem, solved exactly: Kuhn-Munkres with potentials).
      const int n = (int)c.size(), m = MAXC;
      double cost[9][9];
      for (int i = 1; i <= n; ++i) {
        const Box& b = c[i - 1].box;
        const double off[3] = {0.5 * ((double)b.v[0] + b.v[1]) - ctr[0], 0.5 * ((double)b.v[2] + b.v[3]) - ctr[1],
                               0.5 * ((double)b.v[4] + b.v[5]) - ctr[2]};
        for (int s = 0; s < m; ++s) {
          double d = ((s & 4) ? off[0] : -off[0]) + ((s & 2) ? off[1] : -off[1]) + ((s & 1) ? off[2] : -off[2]);
          cost[i][s + 1] = std::isfinite(d) ? -d : 0.;
        }
      }
      double u[9] = {0}, v[9] = {0};
      int p[9] = {0}, way[9] = {0};
      for (int i = 1; i <= n; ++i) {
        p[0] = i;
        int j0 = 0;
        double minv[9];
        bool usedc[9];
        for (int j = 0; j <= m; ++j) { minv[j] = std::numeric_limits<double>::infinity(); usedc[j] = false; }
        do {
          usedc[j0] = true;
          const int i0 = p[j0];
          double delta = std::numeric_limits<double>::infinity();
          int j1 = 0;
          for (int j = 1; j <= m; ++j)
            if (!usedc[j]) {
              const double cur = cost[i0][j] - u[i0] - v[j];
              if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
              if (minv[j] < delta) { delta = minv[j]; j1 = j; }
            }
          for (int j = 0; j <= m; ++j)
            if (usedc[j]) { u[p[j]] += delta; v[j] -= delta; } else minv[j] -= delta;
          j0 = j1;
        } while (p[j0] != 0);
        do { const int j1 = way[j0]; p[j0] = p[j1]; j0 = j1; } while (j0);
      }
      for (int j = 1; j <= m; ++j)
        if (p[j] > 0) { slot_of[p[j] - 1] = j - 1; used[j - 1] = true; }
    } else {
      std::vector<size_t> order(c.size());
      for (size_t i = 0; i < c.size(); ++i) order[i] = i;
      std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return area(c[a].box) > area(c[b].box); });
      for (size_t oi : order) {
        const Box& b = c[oi].box;
        int pref = ((0.5f * (b.v[0] + b.v[1]) > ctr[0]) ? 4 : 0) | ((0.5f * (b.v[2] + b.v[3]) > ctr[1]) ? 2 : 0) |
                   ((0.5f * (b.v[4] + b.v[5]) > ctr[2]) ? 1 : 0);
        int best = -1, best_d = 99;
        for (int s = 0; s < MAXC; ++s)
          if (!used[s]) {
            int d = __builtin_popcount((unsigned)(s ^ pref));
            if (d < best_d) { best_d = d; best = s; }
          }
        used[best] = true;
        slot_of[oi] = best;
      }
    }
    uint32_t q[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t refs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // empty slots: inverted box (lo = 255, hi = 0) and a NONE reference
    for (int s = 0; s < 8; ++s)
      if (!used[s])
        for (int a = 0; a < 3; ++a) q[2 * a + (s >> 2)] |= 255u << (8 * (s & 3));
    for (size_t i = 0; i < c.size(); ++i) {
      const int s = slot_of[i];
      for (int a = 0; a < 3; ++a) {
        // two more box pads on top of the padded fp32 box: the device evaluates these planes in t-space, t = (1024 + q) * B +
        // (A - 1024 * B) (sol_trace.h), whose rounding error is up to ~1.1 pad at the root (|A| 2^-23 + |t| 2^-24 + |A - 1024 B|
        // 2^-24, in units of the pad S 2^-20); with three pads between the plane and the primitive the margin is 2.7x
        float cl = c[i].box.v[2 * a] - 2.0f * pad, chh = c[i].box.v[2 * a + 1] + 2.0f * pad;
        if (!std::isfinite(cl)) cl = lo[a];
        if (!std::isfinite(chh)) chh = hi[a];
        // (clamped as floats: outside a refused range - range_error above - the quotient can be infinite or NaN, which no integer holds)
        const float fl = std::floor((cl - lo[a]) / scale[a]), fh = std::ceil((chh - lo[a]) / scale[a]);
        long ql = fl >= 255.f ? 255L : fl > 0.f ? (long)fl : 0L;
        long qh = fh >= 255.f ? 255L : fh > 0.f ? (long)fh : (fh == fh ? 0L : 255L);
        while (ql > 0 && decode(lo[a], (uint32_t)ql, scale[a]) > cl) --ql;      // conservative under the device's rounding
        while (qh < 255 && decode(lo[a], (uint32_t)qh, scale[a]) < chh) ++qh;
        if (decode(lo[a], (uint32_t)ql, scale[a]) > cl || decode(lo[a], (uint32_t)qh, scale[a]) < chh) {
          // cannot happen: 255 * scale >= extent; be safe and open the box fully on this axis
          ql = 0; qh = 255;
        }
        q[2 * a + (s >> 2)] |= (uint32_t)ql << (8 * (s & 3));
        q[6 + 2 * a + (s >> 2)] |= (uint32_t)qh << (8 * (s & 3));
      }
      uint32_t r = c[i].ref;
      if (SOL_REF_KIND(r) == SOL_REF_NODE) r = build(SOL_REF_INDEX(r), depth + 1);
      refs[s] = r;
    }
    XWide& w = out[wi];
    for (int a = 0; a < 3; ++a) { w.o[a] = lo[a]; w.e[a] = eb[a]; }
    for (int k = 0; k < 12; ++k) w.q[k] = q[k];
    for (int k = 0; k < 8; ++k) w.ref[k] = refs[k];
    return SOL_MAKE_REF(SOL_REF_WIDE, wi);
  }
  // A world that is one primitive: a root with that single child (the device always starts at a wide node).
  uint32_t build_single(uint32_t prim_ref, const Box& box) {
    std::vector<DNode> one(1);
    DNode& n = one[0];
    n.lxmin = box.v[0]; n.lxmax = box.v[1]; n.lymin = box.v[2]; n.lymax = box.v[3]; n.lzmin = box.v[4]; n.lzmax = box.v[5];
    const Box e = empty_box();
    n.rxmin = e.v[0]; n.rxmax = e.v[1]; n.rymin = e.v[2]; n.rymax = e.v[3]; n.rzmin = e.v[4]; n.rzmax = e.v[5];
    n.left = prim_ref; n.right = SOL_MAKE_REF(SOL_REF_NONE, 0); n.pad0 = n.pad1 = 0;
    single_ = one;
    WideBuilder tmp(single_, pad);
    tmp.emin = emin;
    tmp.dp_collapse = false;
    const uint32_t r = tmp.build(0, 0);
    out = tmp.out; max_depth = tmp.max_depth; inner_area = tmp.inner_area; leaf_area = tmp.leaf_area; range_error = tmp.range_error;
    return r;
  }
  std::vector<DNode> single_;
};

// Device form of the wide tree. Depth-first: a node's inner children get consecutive node indices (in slot order), the
// primitives of a node whose leaves are all triangles (or all spheres, or all quads) get consecutive NEW indices in their
// array; other nodes list full references (with new indices) in `leaf_refs`. new_of_old / old_of_new are the permutations of
// the triangle / sphere / quad arrays (index 0 / 1 / 2); primitives the world tree does not hold keep their relative order
// behind the others. remap() rewrites any primitive reference of the flattened scene (binary nodes, lights, mediums).
struct WideLayout {
  std::vector<DWide> nodes;
  std::vector<uint32_t> leaf_refs;
  std::vector<uint32_t> new_of_old[3], old_of_new[3];
  uint32_t depth = 0;
  std::string error;
  static int arr(uint32_t kind) { return kind == SOL_REF_TRIANGLE ? 0 : kind == SOL_REF_SPHERE ? 1 : kind == SOL_REF_QUAD ? 2 : -1; }
  uint32_t remap(uint32_t ref) const {
    const int a = arr(SOL_REF_KIND(ref));
    if (a < 0 || SOL_REF_INDEX(ref) >= new_of_old[a].size()) return ref;
    return SOL_MAKE_REF(SOL_REF_KIND(ref), new_of_old[a][SOL_REF_INDEX(ref)]);
  }
  uint32_t take(uint32_t ref) {  // new index of primitive `ref`, assigned on first use
    const int a = arr(SOL_REF_KIND(ref));
    uint32_t& n = new_of_old[a][SOL_REF_INDEX(ref)];
    if (n == 0xFFFFFFFFu) { n = (uint32_t)old_of_new[a].size(); old_of_new[a].push_back(SOL_REF_INDEX(ref)); }
    return n;
  }
  bool run(const std::vector<XWide>& x, uint32_t root, uint32_t emin, uint32_t n_tris, uint32_t n_spheres, uint32_t n_quads) {
    const uint32_t counts[3] = {n_tris, n_spheres, n_quads};
    for (int a = 0; a < 3; ++a) { new_of_old[a].assign(counts[a], 0xFFFFFFFFu); old_of_new[a].clear(); old_of_new[a].reserve(counts[a]); }
    nodes.assign(1, DWide{});
    leaf_refs.clear();
    struct Item { uint32_t xi, ni, depth; };
    std::vector<Item> stk{{root, 0u, 1u}};
    while (!stk.empty()) {
      const Item it = stk.back();
      stk.pop_back();
      if (it.depth > depth) depth = it.depth;
      const XWide& w = x[it.xi];
      uint32_t imask = 0, lmask = 0, n_inner = 0, n_leaf = 0, kind0 = SOL_REF_NONE;
      bool direct = true;
      for (int s = 0; s < SOL_WIDE_CHILDREN; ++s) {
        const uint32_t k = SOL_REF_KIND(w.ref[s]);
        if (k == SOL_REF_NONE) continue;
        if (k == SOL_REF_WIDE) { imask |= 1u << s; n_inner++; continue; }
        lmask |= 1u << s;
        n_leaf++;
        const int a = arr(k);
        if (a < 0) { direct = false; continue; }  // a constant medium: always listed by reference
        if (SOL_REF_INDEX(w.ref[s]) >= counts[a]) { error = "primitive reference out of range"; return false; }
        if (kind0 == SOL_REF_NONE) kind0 = k; else if (k != kind0) direct = false;
        if (new_of_old[a][SOL_REF_INDEX(w.ref[s])] != 0xFFFFFFFFu) direct = false;  // (shared sub-tree: already placed elsewhere)
      }
      if (SOL_REF_KIND(w.ref[7]) != SOL_REF_NONE) { error = "slot 7 of a wide node is in use"; return false; }
      const uint32_t base_inner = (uint32_t)nodes.size();
      uint32_t base_prim = 0, leaf_kind = SOL_LEAF_REFS;
      if (n_leaf) {
        if (direct && kind0 != SOL_REF_NONE) {
          leaf_kind = kind0 == SOL_REF_TRIANGLE ? SOL_LEAF_TRIANGLES : kind0 == SOL_REF_SPHERE ? SOL_LEAF_SPHERES : SOL_LEAF_QUADS;
          base_prim = (uint32_t)old_of_new[arr(kind0)].size();
          for (int s = 0; s < SOL_WIDE_CHILDREN; ++s)
            if (lmask & (1u << s)) take(w.ref[s]);  // consecutive: nothing else allocates in between
        } else {
          base_prim = (uint32_t)leaf_refs.size();
          for (int s = 0; s < SOL_WIDE_CHILDREN; ++s)
            if (lmask & (1u << s)) {
              const uint32_t r = w.ref[s];
              leaf_refs.push_back(arr(SOL_REF_KIND(r)) >= 0 ? SOL_MAKE_REF(SOL_REF_KIND(r), take(r)) : r);
            }
        }
      }
      if ((uint64_t)base_inner + n_inner > SOL_WIDE_MAX_INDEX || (uint64_t)base_prim + n_leaf > SOL_WIDE_MAX_INDEX) {
        error = "more than 2^24 wide nodes or primitives of one kind";
        return false;
      }
      nodes.resize(nodes.size() + n_inner);
      DWide& o = nodes[it.ni];
      o.ox = w.o[0]; o.oy = w.o[1]; o.oz = w.o[2];
      for (int a = 0; a < 3; ++a)
        if (w.e[a] < emin || w.e[a] > emin + 31) { error = "wide-node exponent outside the 5-bit range"; return false; }
      o.meta = (w.e[0] - emin) | ((w.e[1] - emin) << 5) | ((w.e[2] - emin) << 10) | (imask << 15) | (lmask << 22) | (leaf_kind << 29);
      for (int k = 0; k < 12; ++k) o.q[k] = w.q[k];
      const uint32_t bi = n_inner ? base_inner : 0u;
      for (int k = 0; k < 3; ++k) {  // slot-7 bytes: top byte of the second word of each plane array
        o.q[2 * k + 1] = (o.q[2 * k + 1] & 0x00FFFFFFu) | (((bi >> (8 * k)) & 0xFFu) << 24);
        o.q[6 + 2 * k + 1] = (o.q[6 + 2 * k + 1] & 0x00FFFFFFu) | (((base_prim >> (8 * k)) & 0xFFu) << 24);
      }
      // children in reverse slot order onto the stack: the first child's sub-tree is laid out right behind the sibling block
      uint32_t rank = n_inner;
      for (int s = SOL_WIDE_CHILDREN - 1; s >= 0; --s)
        if (imask & (1u << s)) { --rank; stk.push_back(Item{SOL_REF_INDEX(w.ref[s]), base_inner + rank, it.depth + 1}); }
    }
    for (int a = 0; a < 3; ++a)  // primitives outside the world tree (e.g. the quads of a medium boundary) follow in their old order
      for (uint32_t i = 0; i < counts[a]; ++i)
        if (new_of_old[a][i] == 0xFFFFFFFFu) { new_of_old[a][i] = (uint32_t)old_of_new[a].size(); old_of_new[a].push_back(i); }
    return true;
  }
  // the device build (sol_build.hip) delivers nodes, leaf_refs and new_of_old: derive the inverse maps and check them
  // Re-lays a device-built tree out in depth-first order (as run() does for the host-built ones): the GPU emission reserves
  // indices with atomics, so the ORDER of nodes and primitives in memory differs from run to run while the tree does not;
  // after this pass the layout is a function of the tree alone (reproducible timings), and sub-trees are contiguous.
  bool canonicalize() {
    const uint32_t counts[3] = {(uint32_t)new_of_old[0].size(), (uint32_t)new_of_old[1].size(), (uint32_t)new_of_old[2].size()};
    std::vector<uint32_t> cur2new[3];
    for (int a = 0; a < 3; ++a) cur2new[a].assign(counts[a], 0xFFFFFFFFu);
    uint32_t next_prim[3] = {0, 0, 0};
    std::vector<DWide> out(1);
    out.reserve(nodes.size());
    std::vector<uint32_t> refs_out;
    refs_out.reserve(leaf_refs.size());
    const uint32_t ref_kind_of[4] = {SOL_REF_NONE, SOL_REF_TRIANGLE, SOL_REF_SPHERE, SOL_REF_QUAD};
    struct Item { uint32_t from, to; };
    std::vector<Item> stk{{0u, 0u}};
    auto place = [&](int a, uint32_t cur) -> uint32_t {
      if (a < 0 || cur >= counts[a] || cur2new[a][cur] != 0xFFFFFFFFu) { error = "device tree: a primitive is referenced twice or out of range"; return 0xFFFFFFFFu; }
      return cur2new[a][cur] = next_prim[a]++;
    };
    while (!stk.empty()) {
      const Item it = stk.back();
      stk.pop_back();
      if (it.from >= nodes.size()) { error = "device tree: node index out of range"; return false; }
      DWide w = nodes[it.from];
      const uint32_t imask = (w.meta >> 15) & 0x7Fu, lmask = (w.meta >> 22) & 0x7Fu, kind = (w.meta >> 29) & 3u;
      const uint32_t n_inner = (uint32_t)__builtin_popcount(imask), n_leaf = (uint32_t)__builtin_popcount(lmask);
      const uint32_t bi_from = base_inner(w), bp_from = base_prim(w);
      const uint32_t bi_to = n_inner ? (uint32_t)out.size() : 0u;
      out.resize(out.size() + n_inner);
      uint32_t bp_to = 0;
      if (n_leaf) {
        if (kind == SOL_LEAF_REFS) {
          bp_to = (uint32_t)refs_out.size();
          for (uint32_t r = 0; r < n_leaf; ++r) {
            if (bp_from + r >= leaf_refs.size()) { error = "device tree: listed reference out of range"; return false; }
            uint32_t ref = leaf_refs[bp_from + r];
            const int a = arr(SOL_REF_KIND(ref));
            if (a >= 0) {
              const uint32_t n = place(a, SOL_REF_INDEX(ref));
              if (n == 0xFFFFFFFFu) return false;
              ref = SOL_MAKE_REF(SOL_REF_KIND(ref), n);
            }
            refs_out.push_back(ref);
          }
        } else {
          const int a = arr(ref_kind_of[kind]);
          bp_to = next_prim[a];
          for (uint32_t r = 0; r < n_leaf; ++r)
            if (place(a, bp_from + r) == 0xFFFFFFFFu) return false;
        }
      }
      for (int k = 0; k < 3; ++k) {
        w.q[2 * k + 1] = (w.q[2 * k + 1] & 0x00FFFFFFu) | (((bi_to >> (8 * k)) & 0xFFu) << 24);
        w.q[6 + 2 * k + 1] = (w.q[6 + 2 * k + 1] & 0x00FFFFFFu) | (((bp_to >> (8 * k)) & 0xFFu) << 24);
      }
      out[it.to] = w;
      for (uint32_t r = n_inner; r-- > 0;) stk.push_back(Item{bi_from + r, bi_to + r});  // first child's sub-tree right behind the siblings
    }
    if (out.size() != nodes.size()) { error = "device tree: unreachable nodes"; return false; }
    for (int a = 0; a < 3; ++a)  // primitives outside the world tree: behind the others, in the caller's order
      for (uint32_t old = 0; old < counts[a]; ++old) {
        const uint32_t cur = new_of_old[a][old];
        if (cur >= counts[a]) { error = "device tree: the primitive map is not a permutation"; return false; }
        if (cur2new[a][cur] == 0xFFFFFFFFu) cur2new[a][cur] = next_prim[a]++;
      }
    for (int a = 0; a < 3; ++a)
      for (uint32_t old = 0; old < counts[a]; ++old) new_of_old[a][old] = cur2new[a][new_of_old[a][old]];
    nodes.swap(out);
    leaf_refs.swap(refs_out);
    return true;
  }
  bool adopt_device(std::vector<DWide>&& n, std::vector<uint32_t>&& refs, std::vector<uint32_t> (&no)[3], uint32_t levels) {
    nodes = std::move(n);
    leaf_refs = std::move(refs);
    depth = levels;
    for (int a = 0; a < 3; ++a) new_of_old[a] = std::move(no[a]);
    if (nodes.empty() || !canonicalize()) return false;
    for (int a = 0; a < 3; ++a) {
      old_of_new[a].assign(new_of_old[a].size(), 0xFFFFFFFFu);
      for (uint32_t i = 0; i < new_of_old[a].size(); ++i) {
        const uint32_t k = new_of_old[a][i];
        if (k >= old_of_new[a].size() || old_of_new[a][k] != 0xFFFFFFFFu) { error = "device tree: the primitive map is not a permutation"; return false; }
        old_of_new[a][k] = i;
      }
    }
    return !nodes.empty();
  }
  // decode helpers shared with sol_world_tree_check
  static uint32_t base_inner(const DWide& w) { return (w.q[1] >> 24) | ((w.q[3] >> 24) << 8) | ((w.q[5] >> 24) << 16); }
  static uint32_t base_prim(const DWide& w) { return (w.q[7] >> 24) | ((w.q[9] >> 24) << 8) | ((w.q[11] >> 24) << 16); }
};

// Rebuilds the WORLD's binary tree over the same primitives with a binned surface-area heuristic. The closest hit of a
// search does not depend on the tree (every box bounds its primitives, ties are decided by the primitives' dfs_index in
// the REFERENCE tree, which stays on the records), so the device is free to walk a better tree than the reference's
// centroid-median one (bvh.rs:118-162); only the world search (t >= 0.001) uses it, constant-medium boundaries keep the
// reference-shaped tree and its negative-t rules. Output has TreeBuilder's format (child boxes in the parent).
struct SahBuilder {
  struct Prim { uint32_t ref; Box box; float c[3]; };
  std::vector<Prim> prims;
  std::vector<DNode> nodes;
  static constexpr int MAX_BINS = 64;
  int BINS = 16;

  // Leaves of the reference-shaped device tree under `root` (a shared sub-tree contributes its leaves once per use).
  bool collect(const std::vector<DNode>& bin, uint32_t root) {
    std::vector<std::pair<uint32_t, Box>> stk;
    stk.push_back({root, empty_box()});
    while (!stk.empty()) {
      auto [ref, box] = stk.back();
      stk.pop_back();
      const uint32_t k = SOL_REF_KIND(ref);
      if (k == SOL_REF_NONE) continue;
      if (k == SOL_REF_NODE) {
        const DNode& n = bin[SOL_REF_INDEX(ref)];
        stk.push_back({n.right, WideBuilder::rbox(n)});
        stk.push_back({n.left, WideBuilder::lbox(n)});
        continue;
      }
      Prim p{ref, box, {0.f, 0.f, 0.f}};
      for (int a = 0; a < 3; ++a) {
        if (!std::isfinite(box.v[2 * a]) || !std::isfinite(box.v[2 * a + 1]) || box.v[2 * a] > box.v[2 * a + 1]) return false;
        p.c[a] = 0.5f * (box.v[2 * a] + box.v[2 * a + 1]);
      }
      if (prims.size() >= (1u << 26)) return false;
      prims.push_back(p);
    }
    return prims.size() >= 2;
  }
  static void grow(Box& b, const Box& o) {
    for (int a = 0; a < 3; ++a) { b.v[2 * a] = std::min(b.v[2 * a], o.v[2 * a]); b.v[2 * a + 1] = std::max(b.v[2 * a + 1], o.v[2 * a + 1]); }
  }
  static double area(const Box& b) {
    double dx = (double)b.v[1] - b.v[0], dy = (double)b.v[3] - b.v[2], dz = (double)b.v[5] - b.v[4];
    if (!(dx >= 0. && dy >= 0. && dz >= 0.)) return 0.;
    return dx * dy + dy * dz + dz * dx;
  }
  // Builds [lo, hi) and returns its reference and box.
  uint32_t build(size_t lo, size_t hi, uint32_t depth, Box& out_box) {
    if (hi - lo == 1) { out_box = prims[lo].box; return prims[lo].ref; }
    float cmin[3] = {F_INF, F_INF, F_INF}, cmax[3] = {-F_INF, -F_INF, -F_INF};
    for (size_t i = lo; i < hi; ++i)
      for (int a = 0; a < 3; ++a) { cmin[a] = std::min(cmin[a], prims[i].c[a]); cmax[a] = std::max(cmax[a], prims[i].c[a]); }
    int best_axis = -1, best_bin = -1;
    double best_cost = std::numeric_limits<double>::infinity();
    float best_k = 0.f;
    if (depth < 48)
      for (int a = 0; a < 3; ++a) {
        const float ext = cmax[a] - cmin[a];
        if (!(ext > 0.f)) continue;
        const float k = (float)BINS / ext;
        Box bb[MAX_BINS];
        uint32_t bn[MAX_BINS];
        for (int b = 0; b < BINS; ++b) { bb[b] = empty_box(); bn[b] = 0; }
        for (size_t i = lo; i < hi; ++i) {
          int b = std::min(BINS - 1, std::max(0, (int)((prims[i].c[a] - cmin[a]) * k)));
          grow(bb[b], prims[i].box);
          bn[b]++;
        }
        double ra[MAX_BINS];
        uint32_t rn[MAX_BINS];
        Box acc = empty_box();
        uint32_t n = 0;
        for (int b = BINS - 1; b > 0; --b) { grow(acc, bb[b]); n += bn[b]; ra[b] = area(acc); rn[b] = n; }
        acc = empty_box();
        n = 0;
        for (int b = 0; b < BINS - 1; ++b) {  // split after bin b
          grow(acc, bb[b]);
          n += bn[b];
          if (n == 0 || rn[b + 1] == 0) continue;
          const double cost = area(acc) * n + ra[b + 1] * rn[b + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; best_k = k; }
        }
      }
    size_t mid;
    if (best_axis >= 0) {
      const int a = best_axis;
      const float c0 = cmin[a];
      auto it = std::partition(prims.begin() + lo, prims.begin() + hi, [&](const Prim& p) {
        return std::min(BINS - 1, std::max(0, (int)((p.c[a] - c0) * best_k))) <= best_bin;
      });
      mid = (size_t)(it - prims.begin());
    } else {
      mid = lo;
    }
    if (mid == lo || mid == hi) {  // coincident centroids (or the depth guard): median along the widest axis
      int a = 0;
      for (int k = 1; k < 3; ++k) if (cmax[k] - cmin[k] > cmax[a] - cmin[a]) a = k;
      mid = lo + (hi - lo) / 2;
      std::nth_element(prims.begin() + lo, prims.begin() + mid, prims.begin() + hi, [a](const Prim& x, const Prim& y) { return x.c[a] < y.c[a]; });
    }
    const uint32_t ni = (uint32_t)nodes.size();
    nodes.push_back(DNode{});
    Box lb, rb;
    const uint32_t lr = build(lo, mid, depth + 1, lb);
    const uint32_t rr = build(mid, hi, depth + 1, rb);
    DNode& dn = nodes[ni];
    dn.lxmin = lb.v[0]; dn.lxmax = lb.v[1]; dn.lymin = lb.v[2]; dn.lymax = lb.v[3]; dn.lzmin = lb.v[4]; dn.lzmax = lb.v[5];
    dn.rxmin = rb.v[0]; dn.rxmax = rb.v[1]; dn.rymin = rb.v[2]; dn.rymax = rb.v[3]; dn.rzmin = rb.v[4]; dn.rzmax = rb.v[5];
    dn.left = lr; dn.right = rr; dn.pad0 = dn.pad1 = 0;
    out_box = lb;
    grow(out_box, rb);
    return SOL_MAKE_REF(SOL_REF_NODE, ni);
  }
};

}  // namespace
