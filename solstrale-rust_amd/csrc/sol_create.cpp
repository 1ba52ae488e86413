// sol_create.cpp -- sol_scene_create: validation of the flattened scene, conversion to the fp32 device layout (sol_types.h),
// the world tree (built on the GPU by sol_build.hip, or host candidates + probe), upload, work-order probe; sol_world_tree_check.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "sol_build.h"
#include "sol_scene.h"
#include "sol_tree.h"

// The world tree built on the GPU (sol_build.hip): primitives of the reference-shaped tree under `root_ref` (each once - a
// shared sub-tree is the same geometry twice, one copy finds the same hits), pre-split, clustered and collapsed on the current device.
// With pre-splitting a triangle may have SEVERAL references, each a record of its own in the permuted triangle array (implicit leaf
// addresses want consecutive records): lay.old_of_new[0] then has more entries than the scene has triangles, several of them naming
// the same triangle, and lay.new_of_old[0] names one of a triangle's records (any: they are copies). `split_info`: the build's
// expanded references, for sol_world_tree_check.
struct DeviceSplitInfo {
  std::vector<uint32_t> tri_of_ref;  // expanded reference -> triangle
  std::vector<uint32_t> ref_of_dev;  // device triangle index -> expanded reference
  std::vector<float> ref_box;
  uint32_t split_triangles = 0;
  uint32_t extra_references = 0;
  float area_ratio = 1.f;
  uint32_t reinsertion_moves = 0;
  double area_before = 0., area_after = 0.;
  float collapse_cost = 0.f;  // (SolDeviceTree) the collapse's surface-area cost of the tree
};
// The world's primitives as the device builder takes them (each once, in the order of their references): collected once per scene,
// whatever the number of candidate trees.
static int collect_build_prims(const std::vector<DNode>& bin, uint32_t root_ref, const Box& root_box, std::vector<SolBuildPrim>& prims) {
  prims.clear();
  if (SOL_REF_KIND(root_ref) == SOL_REF_NODE) {
    SahBuilder col;
    if (!col.collect(bin, root_ref)) return sol_fail(SOL_EINVAL, "the world's primitives cannot be collected (non-finite box or fewer than two)");
    // in the order of their references, each once (a shared sub-tree lists its primitives twice): a counting pass per kind instead of a
    // sort - std::sort of C5's 1.09 M records was 0.08 s of sol_scene_create
    uint32_t top[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const auto& q : col.prims) top[SOL_REF_KIND(q.ref) & 7u] = std::max(top[SOL_REF_KIND(q.ref) & 7u], SOL_REF_INDEX(q.ref) + 1u);
    std::vector<int32_t> first[8];
    for (int k = 0; k < 8; ++k) first[k].assign(top[k], -1);
    for (size_t i = 0; i < col.prims.size(); ++i) {
      int32_t& f = first[SOL_REF_KIND(col.prims[i].ref) & 7u][SOL_REF_INDEX(col.prims[i].ref)];
      if (f < 0) f = (int32_t)i;
    }
    prims.reserve(col.prims.size());
    for (int k = 0; k < 8; ++k)
      for (int32_t f : first[k]) {
        if (f < 0) continue;
        SolBuildPrim p;
        for (int j = 0; j < 6; ++j) p.box[j] = col.prims[(size_t)f].box.v[j];
        p.ref = col.prims[(size_t)f].ref; p.pad = 0;
        prims.push_back(p);
      }
  } else {
    SolBuildPrim p;
    for (int k = 0; k < 6; ++k) p.box[k] = root_box.v[k];
    p.ref = root_ref; p.pad = 0;
    prims.push_back(p);
  }
  return SOL_OK;
}
// A device build in its two halves (sol_build.h): the binary tree with its collapse cost, kept on the device behind `handle`; then the
// emission of the wide nodes and their adoption as a host-side layout record.
struct DevicePrepared {
  SolDeviceBuild* handle = nullptr;
  SolDeviceTree dt;
  uint32_t emin = 1;
  DevicePrepared() = default;
  DevicePrepared(const DevicePrepared&) = delete;
  DevicePrepared& operator=(const DevicePrepared&) = delete;
  ~DevicePrepared() { if (handle) sol_build_world_tree_release(handle); }
};
static int device_world_tree_prepare(const std::vector<SolBuildPrim>& prims, const Box& root_box, float box_pad, const uint32_t counts[3], const std::vector<DTri>& tris,
                                     const SolSplitOptions& split, int ploc_radius, hipStream_t stream, DevicePrepared& pr) {
  pr.emin = WideBuilder::exponent_min(root_box, box_pad);
  std::string err;
  const bool have_tris = tris.size() == counts[0] && counts[0] > 0;
  if (!sol_build_world_tree_prepare(prims.data(), (uint32_t)prims.size(), root_box.v, box_pad, pr.emin, counts, have_tris ? tris.data() : nullptr, split, ploc_radius, stream,
                                    pr.dt, &pr.handle, err))
    return sol_fail(SOL_EDEVICE, "%s", err.c_str());
  return SOL_OK;
}
static int device_world_tree_finish(DevicePrepared& pr, const uint32_t counts[3], const SolSplitOptions& split, WideLayout& lay, uint32_t& emin, DeviceSplitInfo* split_info) {
  const auto t_dbg0 = std::chrono::steady_clock::now();
  auto dbg = [&](const char* what) { if (split.verbose) std::fprintf(stderr, "[solstrale] device_world_tree: %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dbg0).count()); };
  emin = pr.emin;
  SolDeviceTree& dt = pr.dt;
  std::string err;
  const bool emitted = sol_build_world_tree_emit(pr.handle, dt, err);
  sol_build_world_tree_release(pr.handle);
  pr.handle = nullptr;
  if (!emitted) return sol_fail(SOL_EDEVICE, "%s", err.c_str());
  dbg("device build emitted");
  const std::vector<uint32_t> extra_of = std::move(dt.extra_of);
  if (!lay.adopt_device(std::move(dt.nodes), std::move(dt.leaf_refs), dt.new_of_old, dt.depth)) return sol_fail(SOL_EDEVICE, "%s", lay.error.c_str());
  if (split_info) {
    split_info->extra_references = (uint32_t)extra_of.size();
    split_info->area_ratio = dt.split_area_ratio;
    split_info->split_triangles = dt.split_triangles;
    split_info->reinsertion_moves = dt.reinsertion_moves; split_info->area_before = dt.area_before; split_info->area_after = dt.area_after;
    split_info->collapse_cost = dt.collapse_cost;
  }
  if (split_info && split.want_boxes) {
    split_info->ref_of_dev = lay.old_of_new[0];
    split_info->tri_of_ref.resize(counts[0] + extra_of.size());
    for (uint32_t e = 0; e < split_info->tri_of_ref.size(); ++e) split_info->tri_of_ref[e] = e < counts[0] ? e : extra_of[e - counts[0]];
    split_info->ref_box = std::move(dt.ref_box);
  }
  if (!extra_of.empty()) {  // expanded references -> triangles
    if (lay.old_of_new[0].size() != (size_t)counts[0] + extra_of.size()) return sol_fail(SOL_EDEVICE, "device tree: split references lost");
    for (uint32_t& o : lay.old_of_new[0])
      if (o >= counts[0]) {
        if (o - counts[0] >= extra_of.size() || extra_of[o - counts[0]] >= counts[0]) return sol_fail(SOL_EDEVICE, "device tree: bad split reference");
        o = extra_of[o - counts[0]];
      }
    lay.new_of_old[0].resize(counts[0]);  // (a triangle's first reference keeps the triangle's own index)
  }
  dbg("layout adopted");
  return SOL_OK;
}
static int device_world_tree(const std::vector<SolBuildPrim>& prims, const Box& root_box, float box_pad, const uint32_t counts[3],
                             const std::vector<DTri>& tris, const SolSplitOptions& split, int ploc_radius, hipStream_t stream, WideLayout& lay, uint32_t& emin,
                             DeviceSplitInfo* split_info) {
  DevicePrepared pr;
  if (int rc = device_world_tree_prepare(prims, root_box, box_pad, counts, tris, split, ploc_radius, stream, pr)) return rc;
  return device_world_tree_finish(pr, counts, split, lay, emin, split_info);
}

// A triangle's fp32 intersect record: starts at the vertex opposite the longest edge (fp32 arithmetic contract, solstrale_hip.h
// sol_triangle_rotation; the oracle's float instantiation makes the same choice); `reference_order`: as the reference lists the vertices -
// the frame a triangle LIGHT is sampled in (DevScene::light_tri). uv_of = which of {uv0, uv1, uv2} belongs to the record's three vertices.
static void cast_triangle(const SolTriangle& t, bool reference_order, DTri& o, int uv_of[3]) {
  double v0[3], e1[3], e2[3];
  sol_triangle_rotated(&t, reference_order ? 0 : sol_triangle_rotation(&t), v0, e1, e2, uv_of);
  o.v0x = (float)v0[0]; o.v0y = (float)v0[1]; o.v0z = (float)v0[2];
  o.e1x = (float)e1[0]; o.e1y = (float)e1[1]; o.e1z = (float)e1[2];
  o.e2x = (float)e2[0]; o.e2y = (float)e2[1]; o.e2z = (float)e2[2];
  o.dfs = t.dfs_index; o.mat = t.material; o.area = (float)t.area;
}
// the sampling frames of the triangle lights, by light index (entries of other lights stay zero)
static std::vector<DTri> light_triangle_frames(const SolSceneDesc& d) {
  std::vector<DTri> frames(std::max<uint32_t>(1u, d.n_lights), DTri{});
  int uv_of[3];
  for (uint32_t i = 0; i < d.n_lights; ++i)
    if (SOL_REF_KIND(d.lights[i]) == SOL_REF_TRIANGLE && SOL_REF_INDEX(d.lights[i]) < d.n_triangles)
      cast_triangle(d.triangles[SOL_REF_INDEX(d.lights[i])], true, frames[i], uv_of);
  return frames;
}
// the triangles' vertices as the device will hold them (what pre-splitting clips)
static std::vector<DTri> cast_triangles(const SolSceneDesc& d) {
  std::vector<DTri> tris(d.n_triangles);
  int uv_of[3];
  for (uint32_t i = 0; i < d.n_triangles; ++i) cast_triangle(d.triangles[i], false, tris[i], uv_of);
  return tris;
}
// Background blocks (include/solstrale_hip.h, SolSceneInfo::background_blocks): the 8x8 pixel blocks of which it can be PROVED that
// every camera ray of every pixel, whatever the jitter and the lens sample, sees nothing - so that every sample is the background
// colour and none has to be generated. A ray of the block (generate_path; Camera::get_ray, src/camera.rs:77-89) leaves a point L of
// the lens - the eye, or eye + lens_radius * (x u + y w) with (x, y) in the unit disc - towards a point T of the focal plane's
// rectangle of the block's pixels. Both sets are bounded by quadrilaterals (the lens disc's square; the rectangle widened by a whole
// pixel on every side plus a bound on the fp32 rounding of generate_path, ordinarily 10^-4 of a pixel). For a plane normal n all those rays lie in the half
// space n . x <= a with a = max n . L as soon as b = max n . (T - L) <= 0, both maxima taken over the corners (n . (T - L) is linear in
// T and in L): candidate normals come from the rectangle's edges and the lens corners (and the viewing direction, for what lies behind
// the camera), built from slightly LARGER quadrilaterals so that the check b <= 0 on the real ones holds with room to spare, and a
// candidate that fails the check is simply not used. The ray set so bounded walks the DEVICE tree as the kernel decodes it, every
// child box inflated by `margin` (64 box pads: the kernel's and the oracle's fp32 slab tests err by about one); a box is passed
// only when a valid plane has the whole box on its outer side. A block whose rays reach no primitive's (leaf) box is a background
// block: for each of its rays the kernel would find every leaf box missed - the quantised leaf boxes contain the primitives' own
// padded boxes, which the reference tree of the oracle tests -, so no primitive test would run on either side.
// Conservative in every step (a block near a silhouette is traced like any other); images never depend on it.
static void find_background_blocks(const WideLayout& L, uint32_t emin, const DCamera& cam, uint32_t width, uint32_t height, double margin,
                                   std::vector<uint8_t>& flags, uint32_t& n_found, uint32_t& n_pixels) {
  const uint32_t bx_n = (width + SOL_TILE - 1) / SOL_TILE, by_n = (height + SOL_TILE - 1) / SOL_TILE;
  flags.assign((size_t)bx_n * by_n, 0);
  n_found = 0; n_pixels = 0;
  if (L.nodes.empty() || width < 2 || height < 2 || !(cam.lens_radius >= 0.0f) || !std::isfinite(cam.lens_radius)) return;
  struct V { double x, y, z; };
  auto dot = [](const V& a, const V& b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
  auto cross = [](const V& a, const V& b) { return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
  auto sub = [](const V& a, const V& b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; };
  const V org{cam.ox, cam.oy, cam.oz}, ll{cam.llx, cam.lly, cam.llz}, hh{cam.hx, cam.hy, cam.hz}, vv{cam.vx, cam.vy, cam.vz};
  const V lu{cam.ux, cam.uy, cam.uz}, lw{cam.wx, cam.wy, cam.wz};
  // the lens: corners of the square around the disc (one point for a pinhole), and of a larger one for the candidate planes
  const int n_lens = cam.lens_radius > 0.0f ? 4 : 1;
  V lens[4], lens_wide[4];
  for (int k = 0; k < 4; ++k) {
    const double sx = (k == 0 || k == 3) ? -1. : 1., sy = k < 2 ? -1. : 1., r = (double)cam.lens_radius * 1.0001, rw = (double)cam.lens_radius * 1.05;
    lens[k] = V{org.x + (lu.x * sx + lw.x * sy) * r, org.y + (lu.y * sx + lw.y * sy) * r, org.z + (lu.z * sx + lw.z * sy) * r};
    lens_wide[k] = V{org.x + (lu.x * sx + lw.x * sy) * rw, org.y + (lu.y * sx + lw.y * sy) * rw, org.z + (lu.z * sx + lw.z * sy) * rw};
  }
  // generate_path forms T and the direction T - L in fp32: each component errs by a few ulps of the largest term. In pixels of the
  // focal plane that is 10^-4 for an ordinary camera; a camera a million units from the origin with a narrow field of view is another
  // matter - the margin grows with it, and beyond three pixels the proof is not attempted.
  const double norm_max = std::max(std::max(std::fabs(ll.x), std::max(std::fabs(ll.y), std::fabs(ll.z))) + std::max(std::fabs(hh.x), std::max(std::fabs(hh.y), std::fabs(hh.z))) +
                                   std::max(std::fabs(vv.x), std::max(std::fabs(vv.y), std::fabs(vv.z))), std::max(std::fabs(org.x), std::max(std::fabs(org.y), std::fabs(org.z)))) + (double)cam.lens_radius * 2.;
  const double pixel = std::min(std::sqrt(dot(hh, hh)) / (double)(width - 1), std::sqrt(dot(vv, vv)) / (double)(height - 1));
  const double rounding_px = pixel > 0. ? 8.0 * 1.1920929e-7 * norm_max / pixel : 1e300;
  if (!(rounding_px < 3.0)) return;
  const double grow = 1.0 + rounding_px;
  struct Plane { V n; double a; };
  std::vector<uint32_t> stack;
  for (uint32_t by = 0; by < by_n; ++by)
    for (uint32_t bx = 0; bx < bx_n; ++bx) {
      const uint32_t x0 = bx * SOL_TILE, x1 = std::min(x0 + SOL_TILE, width), y0 = by * SOL_TILE, y1 = std::min(y0 + SOL_TILE, height);
      // generate_path: u = (px + r) / (W - 1), v = ((H - 1 - py) + r) / (H - 1), r in [0, 1); `grow` pixels of margin on every side
      auto corners = [&](double grow, V t[4]) {
        const double u0 = ((double)x0 - grow) / (double)(width - 1), u1 = ((double)x1 + grow) / (double)(width - 1);
        const double v0 = ((double)height - (double)y1 - grow) / (double)(height - 1), v1 = ((double)height - (double)y0 + grow) / (double)(height - 1);
        const double cu[4] = {u0, u1, u1, u0}, cv[4] = {v0, v0, v1, v1};
        for (int k = 0; k < 4; ++k) t[k] = V{ll.x + hh.x * cu[k] + vv.x * cv[k], ll.y + hh.y * cu[k] + vv.y * cv[k], ll.z + hh.z * cu[k] + vv.z * cv[k]};
      };
      V T[4], Tw[4];
      corners(grow, T);
      corners(grow + 1.0, Tw);
      Plane plane[17];
      int n_planes = 0;
      // keeps the candidate n (pointing AWAY from the rays) if every ray of the block provably stays in n . x <= a
      auto offer = [&](V n) {
        const double len = std::sqrt(dot(n, n));
        if (!(len > 0.) || !std::isfinite(len)) return;
        double a = -1e300, b = -1e300;
        for (int j = 0; j < n_lens; ++j) {
          a = std::max(a, dot(n, lens[j]));
          for (int k = 0; k < 4; ++k) b = std::max(b, dot(n, sub(T[k], lens[j])));
        }
        if (b <= 0.) plane[n_planes++] = Plane{n, a};
      };
      for (int k = 0; k < 4; ++k)
        for (int j = 0; j < n_lens; ++j) {
          const V& Lj = n_lens == 1 ? org : lens_wide[j];
          V n = cross(sub(Tw[(k + 1) & 3], Tw[k]), sub(Tw[k], Lj));
          if (dot(n, sub(Tw[(k + 2) & 3], Lj)) > 0.) n = V{-n.x, -n.y, -n.z};  // the rectangle's far side is inside
          offer(n);
        }
      {
        const V c{T[0].x + T[1].x + T[2].x + T[3].x - 4. * org.x, T[0].y + T[1].y + T[2].y + T[3].y - 4. * org.y, T[0].z + T[1].z + T[2].z + T[3].z - 4. * org.z};
        offer(V{-c.x, -c.y, -c.z});  // what lies behind the camera
      }
      if (n_planes == 0) continue;
      // the least value of n . p over a box: > a = the whole box on the outer side
      auto outside = [&](const double lo[3], const double hi[3]) {
        for (int k = 0; k < n_planes; ++k) {
          const V& n = plane[k].n;
          const double m = std::min(n.x * lo[0], n.x * hi[0]) + std::min(n.y * lo[1], n.y * hi[1]) + std::min(n.z * lo[2], n.z * hi[2]);
          if (m > plane[k].a) return true;
        }
        return false;
      };
      bool reached = false;
      uint32_t visits = 0;
      stack.assign(1, 0u);
      while (!stack.empty() && !reached) {
        const uint32_t ni = stack.back();
        stack.pop_back();
        if (ni >= L.nodes.size() || ++visits > 4096u) { reached = true; break; }
        const DWide& w = L.nodes[ni];
        const float origin[3] = {w.ox, w.oy, w.oz};
        float scale[3];
        for (int a = 0; a < 3; ++a) { const uint32_t bits = (((w.meta >> (5 * a)) & 31u) + emin) << 23; std::memcpy(&scale[a], &bits, 4); }
        const uint32_t imask = (w.meta >> 15) & 0x7Fu, lmask = (w.meta >> 22) & 0x7Fu;
        for (int sl = 0; sl < SOL_WIDE_CHILDREN; ++sl) {
          const uint32_t bit = 1u << sl;
          if (!((imask | lmask) & bit)) continue;
          double lo[3], hi[3];
          for (int a = 0; a < 3; ++a) {
            const uint32_t ql = (w.q[2 * a + (sl >> 2)] >> (8 * (sl & 3))) & 0xFFu, qh = (w.q[6 + 2 * a + (sl >> 2)] >> (8 * (sl & 3))) & 0xFFu;
            lo[a] = (double)WideBuilder::decode(origin[a], ql, scale[a]) - margin;
            hi[a] = (double)WideBuilder::decode(origin[a], qh, scale[a]) + margin;
          }
          if (!(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2])) { reached = true; break; }  // (not a box: trace)
          if (outside(lo, hi)) continue;
          if (lmask & bit) { reached = true; break; }
          stack.push_back(WideLayout::base_inner(w) + (uint32_t)__builtin_popcount(imask & (bit - 1u)));
        }
      }
      if (!reached) {
        flags[(size_t)by * bx_n + bx] = 1;
        n_found++;
        n_pixels += (x1 - x0) * (y1 - y0);
      }
    }
}

static DCamera cast_camera(const SolCamera& c) {
  return DCamera{(float)c.origin[0], (float)c.origin[1], (float)c.origin[2],
                 (float)c.lower_left_corner[0], (float)c.lower_left_corner[1], (float)c.lower_left_corner[2],
                 (float)c.horizontal[0], (float)c.horizontal[1], (float)c.horizontal[2],
                 (float)c.vertical[0], (float)c.vertical[1], (float)c.vertical[2],
                 (float)c.u[0], (float)c.u[1], (float)c.u[2], (float)c.v[0], (float)c.v[1], (float)c.v[2],
                 (float)c.lens_radius};
}
static SolSplitOptions split_options(const SolDevOverrides& ovr, const SolCreateOptions* opt) {
  SolSplitOptions sp;  // (the default: a budget of 30 %, kept when the splits shrink the primitives' summed box area below 85 %)
  if (opt && opt->split_percent < 0) sp.budget = 0.f;
  else if (opt && opt->split_percent > 0) { sp.budget = (float)opt->split_percent / 100.0f; sp.max_area_ratio = 1.f; }  // an explicit budget is kept
  if (ovr.split_percent >= 0) { sp.budget = (float)ovr.split_percent / 100.0f; sp.max_area_ratio = 1.f; }  // SOL_SPLIT (percent; 0 = off)
  if (ovr.split_slack >= 0) sp.level_slack = ovr.split_slack;                                               // SOL_SPLIT_SLACK
  if (ovr.split_keep >= 0) sp.max_area_ratio = (float)ovr.split_keep / 100.0f;                             // SOL_SPLIT_KEEP (percent)
  if (opt && opt->reinsertion_rounds != 0) sp.reinsertion_rounds = std::max(0, opt->reinsertion_rounds);  // (0: the default, 8 rounds)
  if (ovr.reinsert_rounds >= 0) sp.reinsertion_rounds = ovr.reinsert_rounds;                               // SOL_REINSERT (rounds; 0 = off)
  if (ovr.reinsert_stride > 0) sp.reinsertion_stride = ovr.reinsert_stride;                                // SOL_REINSERT_STRIDE
  sp.node_cost = (float)ovr.node_cost;  // (SOL_NODE_COST; 2.5)
  sp.verbose = ovr.verbose;
  return sp;
}

static int world_tree_check(const SolSceneDesc* d, int use_sah, SolTreeCheck* out) {
  if (!d || !out) return sol_fail(SOL_EINVAL, "null argument");
  std::memset(out, 0, sizeof *out);
  const SolDevOverrides ovr = sol_dev_overrides();
  const float box_pad = box_pad_for(*d);
  TreeBuilder tb(*d, box_pad);
  uint32_t root_ref;
  Box root_box;
  if (!tb.resolve(d->root, 0, root_ref, root_box)) return sol_fail(SOL_EINVAL, "world: %s", tb.error.c_str());
  if (SOL_REF_KIND(root_ref) != SOL_REF_NODE) return sol_fail(SOL_EINVAL, "the world is a single primitive: no tree");
  SahBuilder sah;
  if (!sah.collect(tb.nodes, root_ref)) return sol_fail(SOL_EINVAL, "the world's primitives cannot be collected (non-finite box or fewer than two)");
  std::map<uint32_t, int> expected;  // primitive reference -> multiplicity
  std::map<uint32_t, Box> prim_box;
  for (const auto& p : sah.prims) { expected[p.ref]++; prim_box[p.ref] = p.box; }
  if (use_sah < 0)
    for (auto& e : expected) e.second = 1;  // (the device build keeps one copy of a shared sub-tree's primitives)
  out->n_primitives = (uint32_t)sah.prims.size();
  if (use_sah < 0) out->n_primitives = (uint32_t)expected.size();
  WideLayout lay;
  uint32_t emin_used = 1;
  double inner_area = 0., leaf_area = 0.;
  DeviceSplitInfo split;
  std::vector<DTri> dev_tris;
  if (use_sah < 0) {  // the tree sol_build.hip builds on the GPU, checked like the host-built ones
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sol_fail(SOL_EDEVICE, "no HIP device available");
    HIP_TRY(hipSetDevice(0));
    const uint32_t counts[3] = {d->n_triangles, d->n_spheres, d->n_quads};
    dev_tris = cast_triangles(*d);
    SolSplitOptions sp = split_options(ovr, nullptr);
    sp.want_boxes = true;
    std::vector<SolBuildPrim> build_prims;
    int rc = collect_build_prims(tb.nodes, root_ref, root_box, build_prims);
    if (!rc) rc = device_world_tree(build_prims, root_box, box_pad, counts, dev_tris, sp, ovr.ploc_radius, nullptr, lay, emin_used, &split);
    if (rc) return rc;
    // a split triangle has several references; each is expected once
    for (uint32_t e = d->n_triangles; e < split.tri_of_ref.size(); ++e) {
      auto it = expected.find(SOL_MAKE_REF(SOL_REF_TRIANGLE, split.tri_of_ref[e]));
      if (it != expected.end()) it->second++; else out->leaf_mismatches++;
    }
    out->n_extra_references = (uint32_t)(split.tri_of_ref.size() - d->n_triangles);
    out->n_split_triangles = split.split_triangles;
  } else {
    uint32_t bin_root = root_ref;
    if (use_sah) { Box b; sah.BINS = use_sah > 1 ? std::min((int)SahBuilder::MAX_BINS, use_sah) : 16; bin_root = sah.build(0, sah.prims.size(), 0, b); }
    WideBuilder wb(use_sah ? sah.nodes : tb.nodes, box_pad);
    wb.dp_collapse = !ovr.greedy_collapse;
    wb.slot_by_assignment = !ovr.octant_slots;
    wb.NODE_COST = ovr.node_cost;
    wb.set_exponent_range(root_box);
    const uint32_t xroot = wb.build(SOL_REF_INDEX(bin_root), 0);
    if (wb.range_error || !lay.run(wb.out, SOL_REF_INDEX(xroot), wb.emin, d->n_triangles, d->n_spheres, d->n_quads))
      return sol_fail(SOL_EINVAL, "wide tree layout: %s", wb.range_error ? "exponent range" : lay.error.c_str());
    emin_used = wb.emin;
    inner_area = wb.inner_area; leaf_area = wb.leaf_area;
  }
  out->n_wide = (uint32_t)lay.nodes.size();
  out->depth = lay.depth;
  out->inner_area = inner_area; out->leaf_area = leaf_area;
  std::map<uint32_t, int> found;
  // The DEVICE form is what gets checked, decoded exactly as the kernel decodes it (sol_trace.h): 5-bit exponents over emin,
  // implicit child addresses, permuted primitive arrays (mapped back to the caller's indices for the comparison).
  const uint32_t ref_kind_of[4] = {SOL_REF_NONE, SOL_REF_TRIANGLE, SOL_REF_SPHERE, SOL_REF_QUAD};
  // returns the union of the padded primitive boxes below node `ni`
  std::function<Box(uint32_t, uint32_t)> walk = [&](uint32_t ni, uint32_t depth) -> Box {
    Box all = empty_box();
    if (depth > 4096 || ni >= lay.nodes.size()) { out->leaf_mismatches++; return all; }
    const DWide& w = lay.nodes[ni];
    const float origin[3] = {w.ox, w.oy, w.oz};
    float scale[3];
    for (int a = 0; a < 3; ++a) { uint32_t bits = (((w.meta >> (5 * a)) & 31u) + emin_used) << 23; std::memcpy(&scale[a], &bits, 4); }
    const uint32_t imask = (w.meta >> 15) & 0x7Fu, lmask = (w.meta >> 22) & 0x7Fu, kind = (w.meta >> 29) & 3u;
    if (imask & lmask) out->bad_empty_slots++;
    uint32_t n_children = 0;
    for (int s = 0; s < SOL_WIDE_CHILDREN; ++s) {
      uint32_t ql[3], qh[3];
      for (int a = 0; a < 3; ++a) {
        ql[a] = (w.q[2 * a + (s >> 2)] >> (8 * (s & 3))) & 0xFFu;
        qh[a] = (w.q[6 + 2 * a + (s >> 2)] >> (8 * (s & 3))) & 0xFFu;
      }
      const uint32_t bit = 1u << s, below_mask = bit - 1u;
      if (!((imask | lmask) & bit)) {  // an empty slot must have an inverted box (never hit)
        if (!(ql[0] == 255u && qh[0] == 0u && ql[1] == 255u && qh[1] == 0u && ql[2] == 255u && qh[2] == 0u)) out->bad_empty_slots++;
        continue;
      }
      n_children++;
      Box below;
      if (imask & bit) {
        below = walk(WideLayout::base_inner(w) + (uint32_t)__builtin_popcount(imask & below_mask), depth + 1);
      } else {
        const uint32_t idx = WideLayout::base_prim(w) + (uint32_t)__builtin_popcount(lmask & below_mask);
        uint32_t ref = kind == SOL_LEAF_REFS ? (idx < lay.leaf_refs.size() ? lay.leaf_refs[idx] : 0u) : SOL_MAKE_REF(ref_kind_of[kind], idx);
        const int a = WideLayout::arr(SOL_REF_KIND(ref));
        const uint32_t dev_idx = SOL_REF_INDEX(ref);  // (index into the permuted device array; for a listed reference too)
        if (a >= 0) ref = SOL_REF_INDEX(ref) < lay.old_of_new[a].size() ? SOL_MAKE_REF(SOL_REF_KIND(ref), lay.old_of_new[a][SOL_REF_INDEX(ref)]) : 0u;
        found[ref]++;
        out->n_leaf_refs++;
        auto it = prim_box.find(ref);
        below = it == prim_box.end() ? empty_box() : it->second;
        if (a == 0 && !split.ref_box.empty() && dev_idx < split.ref_of_dev.size()) {
          // (device build) the box of THIS reference of the triangle: the whole triangle's, or the part a pre-split gave it
          const uint32_t e = split.ref_of_dev[dev_idx];
          if ((size_t)e * 6 + 6 <= split.ref_box.size()) std::memcpy(below.v, &split.ref_box[(size_t)e * 6], 24);
        }
      }
      bool ok = true;
      for (int a = 0; a < 3; ++a) {
        const float lo = WideBuilder::decode(origin[a], ql[a], scale[a]), hi = WideBuilder::decode(origin[a], qh[a], scale[a]);
        if (below.v[2 * a] <= below.v[2 * a + 1] && !(lo <= below.v[2 * a] && hi >= below.v[2 * a + 1])) ok = false;
      }
      if (!ok) out->box_violations++;
      SahBuilder::grow(all, below);
    }
    if (n_children > out->max_children) out->max_children = n_children;
    return all;
  };
  walk(0, 0);
  // Pre-split triangles: the boxes of a triangle's references must cover the triangle between them (a ray that hits the triangle
  // at a point enters the reference box that holds the point). Checked on a fixed set of 67 points per split triangle: corners,
  // edge mid-points, centroid and 60 low-discrepancy interior points.
  if (split.tri_of_ref.size() > d->n_triangles) {
    std::vector<std::vector<uint32_t>> refs_of(d->n_triangles);
    for (uint32_t e = d->n_triangles; e < split.tri_of_ref.size(); ++e) {
      const uint32_t t = split.tri_of_ref[e];
      if (t >= d->n_triangles) { out->split_uncovered++; continue; }
      if (refs_of[t].empty()) refs_of[t].push_back(t);
      refs_of[t].push_back(e);
    }
    for (uint32_t t = 0; t < d->n_triangles; ++t) {
      if (refs_of[t].empty()) continue;
      const DTri& T = dev_tris[t];
      const double v0[3] = {T.v0x, T.v0y, T.v0z}, e1[3] = {T.e1x, T.e1y, T.e1z}, e2[3] = {T.e2x, T.e2y, T.e2z};
      for (int k = 0; k < 67; ++k) {
        double u, v;
        if (k < 7) { const double pts[7][2] = {{0, 0}, {1, 0}, {0, 1}, {.5, 0}, {0, .5}, {.5, .5}, {1. / 3, 1. / 3}}; u = pts[k][0]; v = pts[k][1]; }
        else { u = std::fmod((k - 6) * 0.7548776662466927, 1.0); v = std::fmod((k - 6) * 0.5698402909980532, 1.0); if (u + v > 1.0) { u = 1.0 - u; v = 1.0 - v; } }
        const double p[3] = {v0[0] + u * e1[0] + v * e2[0], v0[1] + u * e1[1] + v * e2[1], v0[2] + u * e1[2] + v * e2[2]};
        bool in_one = false;
        for (uint32_t e : refs_of[t]) {
          if ((size_t)e * 6 + 6 > split.ref_box.size()) continue;
          const float* b = &split.ref_box[(size_t)e * 6];
          if (p[0] >= b[0] && p[0] <= b[1] && p[1] >= b[2] && p[1] <= b[3] && p[2] >= b[4] && p[2] <= b[5]) { in_one = true; break; }
        }
        if (!in_one) out->split_uncovered++;
      }
    }
  }
  // the permutations must be permutations
  for (int a = 0; a < 3; ++a) {
    // (device build with pre-split triangles: the records are a permutation of the EXPANDED references, several of which are
    // copies of one triangle; every triangle must find a copy of itself at new_of_old)
    const bool expanded = a == 0 && split.ref_of_dev.size() > d->n_triangles;
    const std::vector<uint32_t>& perm = expanded ? split.ref_of_dev : lay.old_of_new[a];
    std::vector<uint8_t> seen(perm.size(), 0);
    for (uint32_t o : perm) { if (o >= seen.size() || seen[o]) out->leaf_mismatches++; else seen[o] = 1; }
    if (!expanded && lay.old_of_new[a].size() != lay.new_of_old[a].size()) out->leaf_mismatches++;
    if (expanded) {
      if (lay.new_of_old[0].size() != d->n_triangles || lay.old_of_new[0].size() != perm.size()) out->leaf_mismatches++;
      else
        for (uint32_t t = 0; t < d->n_triangles; ++t)
          if (lay.new_of_old[0][t] >= lay.old_of_new[0].size() || lay.old_of_new[0][lay.new_of_old[0][t]] != t) out->leaf_mismatches++;
    }
  }
  for (const auto& e : expected) {
    auto it = found.find(e.first);
    const int f = it == found.end() ? 0 : it->second;
    if (f != e.second) out->leaf_mismatches += (uint32_t)std::abs(f - e.second);
  }
  for (const auto& f : found)
    if (!expected.count(f.first)) out->leaf_mismatches += (uint32_t)f.second;
  return SOL_OK;
}

extern "C" {

// The struct has no size field of its own and grew in round 4 (the three split counters): the plain entry point keeps writing the FIRST
// LAYOUT (through leaf_area, SOL_TREE_CHECK_V1_BYTES) so that a binding compiled against that header is not overrun; callers of this
// header pass their struct's size to sol_world_tree_check_ex and get every field that fits.
int sol_world_tree_check_ex(const SolSceneDesc* d, int use_sah, void* out, size_t out_size) {
  if (!out || out_size < SOL_TREE_CHECK_V1_BYTES) return sol_fail(SOL_EINVAL, "sol_world_tree_check_ex: out is null or smaller than the first layout (%d bytes)", (int)SOL_TREE_CHECK_V1_BYTES);
  SolTreeCheck full;
  const int rc = world_tree_check(d, use_sah, &full);
  std::memset(out, 0, out_size);
  std::memcpy(out, &full, std::min(out_size, sizeof full));
  return rc;
}
int sol_world_tree_check(const SolSceneDesc* d, int use_sah, SolTreeCheck* out) { return sol_world_tree_check_ex(d, use_sah, out, SOL_TREE_CHECK_V1_BYTES); }

// Diagnostic, host only: the background blocks sol_scene_create would find with the host-built tree `use_sah` names (as in
// sol_world_tree_check; the proof does not depend on which tree carries it, the count may).
int sol_background_blocks(const SolSceneDesc* d, int use_sah, uint8_t* flags, size_t n_flags, uint32_t* n_found) {
  if (!d || !n_found || use_sah < 0) return sol_fail(SOL_EINVAL, "bad argument");
  *n_found = 0;
  if (d->width < 2 || d->height < 2 || (uint64_t)d->width * d->height > 0x3FFFFFFFull) return sol_fail(SOL_EINVAL, "bad image size %ux%u", d->width, d->height);
  const uint32_t nb = ((d->width + SOL_TILE - 1) / SOL_TILE) * ((d->height + SOL_TILE - 1) / SOL_TILE);
  if (flags && n_flags < nb) return sol_fail(SOL_EINVAL, "%zu flags for %u blocks", n_flags, nb);
  const SolDevOverrides ovr = sol_dev_overrides();
  const float box_pad = box_pad_for(*d);
  TreeBuilder tb(*d, box_pad);
  uint32_t root_ref;
  Box root_box;
  if (!tb.resolve(d->root, 0, root_ref, root_box)) return sol_fail(SOL_EINVAL, "world: %s", tb.error.c_str());
  if (SOL_REF_KIND(root_ref) != SOL_REF_NODE) return sol_fail(SOL_EINVAL, "the world is a single primitive: no tree");
  SahBuilder sah;
  if (!sah.collect(tb.nodes, root_ref)) return sol_fail(SOL_EINVAL, "the world's primitives cannot be collected (non-finite box or fewer than two)");
  uint32_t bin_root = root_ref;
  if (use_sah) { Box b; sah.BINS = use_sah > 1 ? std::min((int)SahBuilder::MAX_BINS, use_sah) : 16; bin_root = sah.build(0, sah.prims.size(), 0, b); }
  WideBuilder wb(use_sah ? sah.nodes : tb.nodes, box_pad);
  wb.dp_collapse = !ovr.greedy_collapse;
  wb.slot_by_assignment = !ovr.octant_slots;
  wb.NODE_COST = ovr.node_cost;
  wb.set_exponent_range(root_box);
  const uint32_t xroot = wb.build(SOL_REF_INDEX(bin_root), 0);
  WideLayout lay;
  if (wb.range_error || !lay.run(wb.out, SOL_REF_INDEX(xroot), wb.emin, d->n_triangles, d->n_spheres, d->n_quads))
    return sol_fail(SOL_EINVAL, "wide tree layout: %s", wb.range_error ? "exponent range" : lay.error.c_str());
  const bool has_env = d->abi_version >= 2u && d->env_texels && d->env_width && d->env_height;
  std::vector<uint8_t> f(nb, 0);
  uint32_t pixels = 0;
  if (!has_env) find_background_blocks(lay, wb.emin, cast_camera(d->camera), d->width, d->height, 64.0 * (double)box_pad, f, *n_found, pixels);
  if (flags) std::memcpy(flags, f.data(), nb);
  return SOL_OK;
}

int sol_scene_create(const SolSceneDesc* d, int device, SolScene** out) { return sol_scene_create_ex(d, device, nullptr, out); }

int sol_scene_create_ex(const SolSceneDesc* d, int device, const SolCreateOptions* opt_in, SolScene** out) {
  if (!d || !out) return sol_fail(SOL_EINVAL, "null argument");
  *out = nullptr;
  const SolDevOverrides ovr = sol_dev_overrides();
  SolCreateOptions opt{};
  if (opt_in) {
    if (opt_in->size < 8 || opt_in->size > 4096) return sol_fail(SOL_EINVAL, "SolCreateOptions.size %u", opt_in->size);
    std::memcpy(&opt, opt_in, std::min<size_t>(opt_in->size, sizeof opt));
  }
  if (opt.world_tree < SOL_TREE_AUTO || opt.world_tree > SOL_TREE_HOST_PROBE) return sol_fail(SOL_EINVAL, "bad world_tree option %d", opt.world_tree);
  if (opt.split_percent > 1000 || opt.reinsertion_rounds > 1024)  // (a typo must not become a build of hours: ten times the references, a thousand rounds)
    return sol_fail(SOL_EINVAL, "SolCreateOptions: split_percent %d (at most 1000) / reinsertion_rounds %d (at most 1024)", opt.split_percent, opt.reinsertion_rounds);
  const auto t_begin = std::chrono::steady_clock::now();
  auto seconds_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  if (d->abi_version != SOL_ABI_VERSION && d->abi_version != 1u) return sol_fail(SOL_EINVAL, "abi_version %u, expected %u (or 1)", d->abi_version, SOL_ABI_VERSION);
  const bool has_env = d->abi_version >= 2u && d->env_texels && d->env_width && d->env_height;  // (a version-1 description ends before these fields)
  if (has_env && ((uint64_t)d->env_width * d->env_height > (1ull << 28) || !std::isfinite(d->env_scale))) return sol_fail(SOL_EINVAL, "bad environment map");
  if (d->width < 2 || d->height < 2 || (uint64_t)d->width * d->height > 0x3FFFFFFFull) return sol_fail(SOL_EINVAL, "bad image size %ux%u", d->width, d->height);
  if (d->shader_kind > SOL_SHADER_SIMPLE) return sol_fail(SOL_EINVAL, "bad shader kind %u", d->shader_kind);
  if ((d->n_nodes && !d->nodes) || (d->n_spheres && !d->spheres) || (d->n_quads && !d->quads) ||
      (d->n_triangles && !d->triangles) || (d->n_mediums && !d->mediums) || (d->n_materials && !d->materials) ||
      (d->n_textures && !d->textures) || (d->n_texel_bytes && !d->texels) || (d->n_lights && !d->lights))
    return sol_fail(SOL_EINVAL, "null array with non-zero count");
  // Renderer::new: "Scene should have at least one light" (src/renderer/mod.rs:143-147)
  if (d->n_lights == 0) return sol_fail(SOL_ENOLIGHT, "Scene should have at least one light");
  if (d->n_texel_bytes > 0xFFFFFFF0ull) return sol_fail(SOL_EINVAL, "more than 4 GiB of texels");

  // ---- materials / textures ----
  std::vector<DTex> texs(d->n_textures);
  for (uint32_t i = 0; i < d->n_textures; ++i) {
    const SolTexture& t = d->textures[i];
    DTex& o = texs[i];
    std::memset(&o, 0, sizeof o);
    o.kind = t.kind;
    if (t.kind == SOL_TEX_IMAGE) {
      // (no sum or product here may wrap: an offset of 2^64 - 1 plus a size is a small number again)
      const uint64_t px = (uint64_t)t.width * t.height;
      if (!t.width || !t.height || px > d->n_texel_bytes / 3 || t.texel_offset > d->n_texel_bytes - px * 3)
        return sol_fail(SOL_EINVAL, "texture %u: image outside texel buffer", i);
      o.w = t.width; o.h = t.height; o.offset = (uint32_t)t.texel_offset;
    } else if (t.kind == SOL_TEX_SOLID) {
      o.r = (float)t.rgb[0]; o.g = (float)t.rgb[1]; o.b = (float)t.rgb[2];
    } else {
      return sol_fail(SOL_EINVAL, "texture %u: bad kind %d", i, t.kind);
    }
  }
  auto tex_ok = [&](int32_t id, bool optional) { return (optional && id < 0) || (id >= 0 && (uint32_t)id < d->n_textures); };
  std::vector<DMat> mats(d->n_materials);
  for (uint32_t i = 0; i < d->n_materials; ++i) {
    const SolMaterial& m = d->materials[i];
    DMat& o = mats[i];
    std::memset(&o, 0, sizeof o);
    o.kind = m.kind; o.albedo = m.albedo_tex; o.normal = m.normal_tex; o.m1 = m.m1; o.m2 = m.m2;
    o.param = (float)m.param;
    if (std::isnan(m.param)) o.flags |= DMAT_PARAM_NONE;
    if (m.kind != SOL_MAT_BLEND && m.albedo_tex >= 0 && (uint32_t)m.albedo_tex < d->n_textures && texs[m.albedo_tex].kind == SOL_TEX_SOLID) {
      o.flags |= DMAT_ALBEDO_SOLID;
      o.ar = texs[m.albedo_tex].r; o.ag = texs[m.albedo_tex].g; o.ab = texs[m.albedo_tex].b;
    }
    switch (m.kind) {
      case SOL_MAT_LAMBERTIAN: case SOL_MAT_METAL: case SOL_MAT_DIELECTRIC:
        if (!tex_ok(m.albedo_tex, false) || !tex_ok(m.normal_tex, true)) return sol_fail(SOL_EINVAL, "material %u: bad texture id", i);
        break;
      case SOL_MAT_DIFFUSE_LIGHT: case SOL_MAT_ISOTROPIC:
        if (!tex_ok(m.albedo_tex, false)) return sol_fail(SOL_EINVAL, "material %u: bad texture id", i);
        o.normal = -1;
        break;
      case SOL_MAT_BLEND:
        if (m.m1 < 0 || m.m2 < 0 || (uint32_t)m.m1 >= d->n_materials || (uint32_t)m.m2 >= d->n_materials || (uint32_t)m.m1 == i || (uint32_t)m.m2 == i)
          return sol_fail(SOL_EINVAL, "material %u: bad blend children", i);
        break;
      default: return sol_fail(SOL_EINVAL, "material %u: bad kind %d", i, m.kind);
    }
  }
  // NEEDS_UV: any image texture reachable from the material (Blend children included; bounded iteration)
  for (int pass = 0; pass < 16; ++pass)
    for (uint32_t i = 0; i < d->n_materials; ++i) {
      DMat& o = mats[i];
      bool need = false;
      if (o.kind == SOL_MAT_BLEND) need = (mats[o.m1].flags | mats[o.m2].flags) & DMAT_NEEDS_UV;
      else need = (o.albedo >= 0 && texs[o.albedo].kind == SOL_TEX_IMAGE) || (o.normal >= 0 && texs[o.normal].kind == SOL_TEX_IMAGE);
      if (need) o.flags |= DMAT_NEEDS_UV;
    }
  auto mat_ok = [&](int32_t id) { return id >= 0 && (uint32_t)id < d->n_materials; };

  // ---- primitives (plain casts) ----
  std::vector<DTri> tris(d->n_triangles);
  std::vector<DTriShade> tshade(d->n_triangles);
  {
    // (a million triangles are 0.03 s of casts on one core: split over a few threads above 64 k; every triangle is independent)
    auto cast_range = [&](uint32_t i0, uint32_t i1, int64_t* bad) {
      for (uint32_t i = i0; i < i1; ++i) {
        const SolTriangle& t = d->triangles[i];
        if (!mat_ok(t.material)) { if (*bad < 0) *bad = i; continue; }
        int uo[3];
        cast_triangle(t, false, tris[i], uo);
        const float* uvs[3] = {t.uv0, t.uv1, t.uv2};
        DTriShade& s = tshade[i];
        s.nx = (float)t.normal[0]; s.ny = (float)t.normal[1]; s.nz = (float)t.normal[2]; s.mat = t.material;
        s.tx = (float)t.tangent[0]; s.ty = (float)t.tangent[1]; s.tz = (float)t.tangent[2];
        s.bx = (float)t.bi_tangent[0]; s.by = (float)t.bi_tangent[1]; s.bz = (float)t.bi_tangent[2];
        s.u0 = uvs[uo[0]][0]; s.v0 = uvs[uo[0]][1]; s.u1 = uvs[uo[1]][0]; s.v1 = uvs[uo[1]][1]; s.u2 = uvs[uo[2]][0]; s.v2 = uvs[uo[2]][1];
      }
    };
    const uint32_t nt = d->n_triangles;
    const uint32_t n_thr = nt >= 65536u ? std::min<uint32_t>(8u, std::max<uint32_t>(1u, std::thread::hardware_concurrency())) : 1u;
    std::vector<int64_t> bad(n_thr, -1);
    std::vector<std::thread> pool;
    for (uint32_t k = 1; k < n_thr; ++k) pool.emplace_back(cast_range, (uint32_t)((uint64_t)nt * k / n_thr), (uint32_t)((uint64_t)nt * (k + 1) / n_thr), &bad[k]);
    cast_range(0, (uint32_t)((uint64_t)nt / n_thr), &bad[0]);
    for (auto& th : pool) th.join();
    for (int64_t b : bad)
      if (b >= 0) return sol_fail(SOL_EINVAL, "triangle %u: bad material", (uint32_t)b);
  }
  std::vector<DQuad> quads(d->n_quads);
  for (uint32_t i = 0; i < d->n_quads; ++i) {
    const SolQuad& q = d->quads[i];
    if (!mat_ok(q.material)) return sol_fail(SOL_EINVAL, "quad %u: bad material", i);
    DQuad& o = quads[i];
    o.nx = (float)q.normal[0]; o.ny = (float)q.normal[1]; o.nz = (float)q.normal[2]; o.d = (float)q.d;
    o.qx = (float)q.q[0]; o.qy = (float)q.q[1]; o.qz = (float)q.q[2]; o.dfs = q.dfs_index;
    o.wx = (float)q.w[0]; o.wy = (float)q.w[1]; o.wz = (float)q.w[2]; o.mat = q.material;
    o.ux = (float)q.u[0]; o.uy = (float)q.u[1]; o.uz = (float)q.u[2]; o.area = (float)q.area;
    o.vx = (float)q.v[0]; o.vy = (float)q.v[1]; o.vz = (float)q.v[2]; o.pad = 0.f;
  }
  std::vector<DSphere> spheres(d->n_spheres);
  for (uint32_t i = 0; i < d->n_spheres; ++i) {
    const SolSphere& s = d->spheres[i];
    if (!mat_ok(s.material)) return sol_fail(SOL_EINVAL, "sphere %u: bad material", i);
    DSphere& o = spheres[i];
    o.cx = (float)s.center[0]; o.cy = (float)s.center[1]; o.cz = (float)s.center[2]; o.radius = std::fabs((float)s.radius);  // |r|: the reference uses r^2 and a min/max box only (sphere.rs:26-28,68), the fp32 rules of sol_trace.h / sol_shade.h use r itself
    o.dfs = s.dfs_index; o.mat = s.material; o.pad0 = o.pad1 = 0;
  }

  if (sol_dev_overrides().verbose) std::fprintf(stderr, "[solstrale] create: records cast at %.1f ms\n", 1e3 * seconds_since(t_begin));
  // ---- tree ----
  const float box_pad = box_pad_for(*d);
  const bool scene_has_needles = sol_scene_has_needles(d) != 0;  // (a pass over every triangle: asked once more here, not once per use)
  // (the 7-wide node test's plane parameters must not overflow: sol_trace.h, wide_node_test; pad = largest |coordinate| * 2^-20)
  if (!(box_pad * 1048576.0f <= 2.7487791e11f)) return sol_fail(SOL_EINVAL, "the scene's coordinates reach beyond 2^38 (%g): not supported by the fp32 search", (double)box_pad * 1048576.0);
  TreeBuilder tb(*d, box_pad);
  uint32_t root_ref;
  Box root_box;
  if (!tb.resolve(d->root, 0, root_ref, root_box)) return sol_fail(SOL_EINVAL, "world: %s", tb.error.c_str());
  if (SOL_REF_KIND(root_ref) == SOL_REF_NONE) return sol_fail(SOL_EINVAL, "world is empty");
  if (sol_dev_overrides().verbose) std::fprintf(stderr, "[solstrale] create: reference tree resolved at %.1f ms\n", 1e3 * seconds_since(t_begin));
  std::vector<DMedium> mediums(d->n_mediums);
  uint32_t medium_depth = 0;
  for (uint32_t i = 0; i < d->n_mediums; ++i) {
    const SolMedium& m = d->mediums[i];
    if (!mat_ok(m.material)) return sol_fail(SOL_EINVAL, "medium %u: bad material", i);
    if (i >= 0x1000u) return sol_fail(SOL_EINVAL, "more than 4096 constant mediums");
    DMedium& o = mediums[i];
    std::memset(&o, 0, sizeof o);
    tb.max_depth = 0;
    uint32_t bref;
    Box bb;
    if (!tb.resolve(m.boundary, 0, bref, bb)) return sol_fail(SOL_EINVAL, "medium %u boundary: %s", i, tb.error.c_str());
    if (SOL_REF_KIND(bref) == SOL_REF_MEDIUM || SOL_REF_KIND(bref) == SOL_REF_NONE) return sol_fail(SOL_EINVAL, "medium %u: unsupported boundary", i);
    medium_depth = std::max(medium_depth, tb.max_depth);
    o.boundary = bref; o.mat = m.material; o.nid = (float)m.negative_inverse_density; o.dfs = m.dfs_index;
    o.bxmin = bb.v[0]; o.bxmax = bb.v[1]; o.bymin = bb.v[2]; o.bymax = bb.v[3]; o.bzmin = bb.v[4]; o.bzmax = bb.v[5];
  }
  // a medium inside a medium boundary would recurse in the device search: reject (never built by the reference's scenes)
  for (uint32_t i = 0; i < d->n_mediums; ++i) {
    std::vector<uint32_t> stk{mediums[i].boundary};
    while (!stk.empty()) {
      uint32_t r = stk.back(); stk.pop_back();
      if (SOL_REF_KIND(r) == SOL_REF_MEDIUM) return sol_fail(SOL_EINVAL, "medium %u: nested ConstantMedium in a boundary is unsupported", i);
      if (SOL_REF_KIND(r) == SOL_REF_NODE) { stk.push_back(tb.nodes[SOL_REF_INDEX(r)].left); stk.push_back(tb.nodes[SOL_REF_INDEX(r)].right); }
    }
  }
  // 7-wide tree of the world. Candidates: the reference's topology collapsed, and binned-SAH rebuilds over the same primitives
  // with 8, 16 and 64 bins (how well the binary splits line up with the wide collapse varies with the bin count: with the
  // first, 8-wide layout C2 visited 9.8 / 12.2 / 11.4 nodes per ray at 8 / 16 / 64 bins and 11.2 on the reference's
  // topology; C3 13.3 / 13.0 / 12.9 vs 14.4). A counted probe render on the device picks one (below).
  // SolCreateOptions.world_tree (or SOL_BVH=ref | sah (16 bins) | sah8 | sah16 | sah64) forces a candidate.
  struct TreeCand {
    std::string name;
    std::unique_ptr<SahBuilder> sah;
    std::unique_ptr<WideBuilder> wb;
    WideLayout lay;
    uint32_t depth = 0, emin = 1;
    DevTree dev;
    double cost = 0.;
  };
  std::vector<TreeCand> cands;
  // stack entries (dwords): a wide level keeps at most one sibling group of two dwords
  auto depth_of = [&](const WideLayout& l) { return 2u * l.depth + medium_depth + 2; };
  const uint32_t stack_limit = SOL_LDS_STACK + SOL_SPILL_STACK;
  // AUTO = the device build: as good a tree as the probed host candidates (node visits per ray, host probe / device: C2 11.0 /
  // 10.9, C3 12.8 / 13.0, C5 6.8 / 6.9) in a sixth to an eighth of the time (sol_scene_create, C3: 0.40 s -> 0.06 s, C5 2.2 s -> 0.3 s)
  static const char* const tree_names[] = {"device", "ref", "sah8", "sah16", "sah64", "device", ""};
  std::string want = !ovr.bvh.empty() ? ovr.bvh : tree_names[opt.world_tree];  // (SOL_BVH: developer override of SolCreateOptions.world_tree)
  if (want == "host") want = "";  // all host candidates + the probe
  auto tune = [&](WideBuilder& wb) { wb.dp_collapse = !ovr.greedy_collapse; wb.slot_by_assignment = !ovr.octant_slots; wb.NODE_COST = ovr.node_cost; };
  auto finish_cand = [&](TreeCand& c, uint32_t wide_root) {  // explicit tree -> device layout
    if (c.wb->range_error || !c.lay.run(c.wb->out, SOL_REF_INDEX(wide_root), c.wb->emin, d->n_triangles, d->n_spheres, d->n_quads)) {
      c.wb.reset();
      return;
    }
    c.depth = depth_of(c.lay);
    c.emin = c.wb->emin;
  };
  // The host candidates wanted by `want` ("" = all of them, the probe decides): into `cands`, the provisional best first.
  auto host_candidates = [&](const std::string& want) -> int {
  std::string layout_error;
  if (SOL_REF_KIND(root_ref) == SOL_REF_NODE) {
    {
      TreeCand c;
      c.name = "ref";
      c.wb.reset(new WideBuilder(tb.nodes, box_pad));
      tune(*c.wb);
      c.wb->set_exponent_range(root_box);
      finish_cand(c, c.wb->build(SOL_REF_INDEX(root_ref), 0));
      if (!c.wb) layout_error = c.lay.error.empty() ? "wide tree: exponent range" : c.lay.error;
      else cands.push_back(std::move(c));
    }
    std::vector<int> bin_list = {8, 16, 64};
    if (!ovr.sah_bins.empty()) bin_list = ovr.sah_bins;  // (SOL_SAH_LIST)
    // (the rebuilds are independent of each other: one host thread each)
    std::vector<std::future<TreeCand>> jobs;
    for (int bins : bin_list) {
      const std::string name = "sah" + std::to_string(bins);
      if (want == "ref" || (!want.empty() && want != name)) continue;
      jobs.push_back(std::async(std::launch::async, [&, bins, name]() {
        TreeCand c;
        c.name = name;
        c.sah.reset(new SahBuilder());
        c.sah->BINS = bins;
        if (!c.sah->collect(tb.nodes, root_ref)) return c;  // non-finite boxes or a single primitive: no rebuild (wb stays null)
        Box bx;
        const uint32_t r = c.sah->build(0, c.sah->prims.size(), 0, bx);
        c.wb.reset(new WideBuilder(c.sah->nodes, box_pad));
        tune(*c.wb);
        c.wb->set_exponent_range(root_box);
        finish_cand(c, c.wb->build(SOL_REF_INDEX(r), 0));
        return c;
      }));
    }
    for (auto& j : jobs) {
      TreeCand c = j.get();
      if (c.wb) cands.push_back(std::move(c));
    }
    if (cands.empty()) return sol_fail(SOL_EINVAL, "world: %s", layout_error.c_str());
    // drop what cannot run; a forced choice drops the rest
    std::vector<TreeCand> keep;
    for (auto& c : cands)
      if (c.depth <= stack_limit && (want.empty() || c.name == want || (want != "ref" && c.name == "ref" && cands.size() == 1))) keep.push_back(std::move(c));
    if (keep.empty()) {
      uint32_t dmin = 0xFFFFFFFFu;
      for (auto& c : cands) if (c.wb) dmin = std::min(dmin, c.depth);
      return sol_fail(SOL_EDEPTH, "BVH depth %u exceeds the traversal stack (%d)", dmin, stack_limit);
    }
    cands = std::move(keep);
    // provisional choice by the surface-area estimate (it knows nothing of occlusion and visit order: the probe decides)
    size_t best = 0;
    for (size_t i = 1; i < cands.size(); ++i)
      if (cands[i].wb->cost() < cands[best].wb->cost()) best = i;
    std::swap(cands[0], cands[best]);
  } else {  // the world is ONE primitive: a root with a single child
    TreeCand c;
    c.name = "ref";
    c.wb.reset(new WideBuilder(tb.nodes, box_pad));
    c.wb->set_exponent_range(root_box);
    finish_cand(c, c.wb->build_single(root_ref, root_box));
    if (!c.wb) return sol_fail(SOL_EINVAL, "world: %s", c.lay.error.c_str());
    cands.push_back(std::move(c));
  }
  return SOL_OK;
  };
  // AUTO falls back to the host candidates when the device build cannot make a tree (a primitive with a non-finite or inverted box -
  // e.g. a NaN vertex of an OBJ -, more than 2^23 primitives, no memory for its scratch): scenes the host path accepts are never
  // refused by the default. An explicit SOL_TREE_DEVICE (or SOL_BVH=device) keeps the hard error.
  const bool device_explicit = opt.world_tree == SOL_TREE_DEVICE || ovr.bvh == "device";
  bool device_build = want == "device";
  std::string fallback_note;
  if (!device_build) { int rc0 = host_candidates(want); if (rc0) return rc0; }
  double t_host_trees = seconds_since(t_begin);
  if (sol_dev_overrides().verbose) std::fprintf(stderr, "[solstrale] create: host part done at %.1f ms\n", 1e3 * t_host_trees);

  // ---- lights ----
  std::vector<uint32_t> lights(d->lights, d->lights + d->n_lights);
  for (uint32_t i = 0; i < d->n_lights; ++i) {
    uint32_t k = SOL_REF_KIND(lights[i]), x = SOL_REF_INDEX(lights[i]);
    bool ok = (k == SOL_REF_SPHERE && x < d->n_spheres) || (k == SOL_REF_QUAD && x < d->n_quads) || (k == SOL_REF_TRIANGLE && x < d->n_triangles);
    if (!ok) return sol_fail(SOL_EINVAL, "light %u: not a sphere/quad/triangle reference", i);
  }

  // ---- device ----
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sol_fail(SOL_EDEVICE, "no HIP device available");
  if (device < 0 || device >= ndev) return sol_fail(SOL_EDEVICE, "device %d out of range (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  SolScene* s = new SolScene();
  s->device = device;
  s->build_times[0] = t_host_trees;
  const auto t_upload0 = std::chrono::steady_clock::now();
  struct Cleanup { SolScene* s; bool keep = false; ~Cleanup() { if (!keep) sol_scene_destroy(s); } } cleanup{s};
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  s->n_cu = prop.multiProcessorCount;
  HIP_TRY(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
  s->stream = s->own_stream;
  int rc;
  std::vector<DeviceSplitInfo> dev_info;  // (per device candidate, in the order of `cands`)
  if (device_build) {
    // The clustering radius of the device build decides little on average and a few per cent on any one scene, not monotonically (greedy
    // clustering; MI355X, 1080p x 64 spp, ms with radius 8 / 16 / 32 / 64: C3 61.0 / 63.0 / 63.8 / 65.0, C2 37.1 / 35.7 / 34.7 / 37.3, C5
    // 37.7 / 38.0 / 37.6 / 36.8, heterogeneous atrium 67.8 / 68.2 / 74.7 / 76.2 - profiles/r04_tree_ploc_radius.txt): round 4's SOL_TREE_AUTO
    // built the tree with radius 16, 8 and 32, uploaded all three and let the counted probe choose (the probe's cost - 2.5 per node visit,
    // 1 per primitive test - ranks them as the render times do): C5 paid twice the build time for a 0 % choice. An explicit
    // SOL_TREE_DEVICE builds one tree (radius 16); SOL_PLOC_R forces a radius.
    const auto t_dev0 = std::chrono::steady_clock::now();
    // Round 5: two radii (16 never won a probe on the four scene families and costs a third of the build time), each built as far as the
    // collapse's surface-area cost of the whole tree; what is emitted, uploaded and probed follows from the two costs
    // (profiles/r05_scene_creation.txt: on triangle meshes the cost ranks the candidates as the probes and the render times do - C3 1.7 %, the
    // heterogeneous atrium 2.1 % apart -; on C5 the candidates are 0.3 % apart and render within 1 % of each other; on C2's spheres they are
    // 0.6 % apart and the cost ranks them the WRONG way round - box area overstates how often a sphere is hit - by 7 % of render time):
    //   apart by less than 0.4 % or by more than 1.2 %: the cheaper tree, unprobed;   in between: both, the counted probe decides.
    std::vector<int> radii = {16};
    if (ovr.ploc_radius > 0) radii = {ovr.ploc_radius};
    else if (opt.world_tree == SOL_TREE_AUTO && ovr.bvh.empty()) radii = {8, 32};
    const uint32_t counts[3] = {d->n_triangles, d->n_spheres, d->n_quads};
    std::vector<SolBuildPrim> build_prims;
    rc = collect_build_prims(tb.nodes, root_ref, root_box, build_prims);
    const SolSplitOptions sopt = split_options(ovr, &opt);
    std::vector<std::unique_ptr<DevicePrepared>> prepared;
    std::vector<int> prepared_radius;
    for (int radius : radii) {
      if (rc && prepared.empty()) break;
      std::unique_ptr<DevicePrepared> pr(new DevicePrepared);
      rc = device_world_tree_prepare(build_prims, root_box, box_pad, counts, tris, sopt, radius, s->stream, *pr);
      if (rc) {
        if (!prepared.empty()) { rc = SOL_OK; continue; }  // (a later candidate failed: the earlier ones stand)
        break;
      }
      if (ovr.verbose) std::fprintf(stderr, "[solstrale] device tree (radius %d): collapse cost %.6g\n", radius, (double)pr->dt.collapse_cost);
      prepared.push_back(std::move(pr));
      prepared_radius.push_back(radius);
    }
    if (prepared.size() == 2) {
      const double a = prepared[0]->dt.collapse_cost, b = prepared[1]->dt.collapse_cost;
      const double apart = (a > 0. && b > 0.) ? std::fabs(a - b) / std::min(a, b) : 0.;
      const bool probe_both = ovr.probe_radii > 0 || (ovr.probe_radii < 0 && apart >= 0.004 && apart <= 0.012);  // (SOL_PROBE_RADII=1 / 0 forces)
      if (b < a) { std::swap(prepared[0], prepared[1]); std::swap(prepared_radius[0], prepared_radius[1]); }  // the cheaper one first
      if (ovr.verbose) std::fprintf(stderr, "[solstrale] device trees: collapse costs %.4f %% apart -> %s\n", apart * 100., probe_both ? "both emitted, the probe decides" : "the cheaper one, unprobed");
      if (!probe_both) { prepared.pop_back(); prepared_radius.pop_back(); }
    }
    for (size_t k = 0; k < prepared.size() && !(rc && cands.empty()); ++k) {
      const int radius = prepared_radius[k];
      TreeCand c;
      c.name = radii.size() > 1 ? "device" + std::to_string(radius) : "device";
      DeviceSplitInfo si;
      rc = device_world_tree_finish(*prepared[k], counts, sopt, c.lay, c.emin, &si);
      if (ovr.verbose) std::fprintf(stderr, "[solstrale] device tree (radius %d): pre-splitting %u triangles into %u extra references, box area ratio %.3f%s; %u reinsertion moves\n",
                                    radius, si.split_triangles, si.extra_references, si.area_ratio, si.extra_references ? "" : " (not kept)", si.reinsertion_moves);
      if (!rc) {
        c.depth = depth_of(c.lay);
        if (c.depth > stack_limit) rc = sol_fail(SOL_EDEPTH, "BVH depth %u exceeds the traversal stack (%d)", c.depth, stack_limit);
      }
      if (rc) {
        if (!cands.empty()) { rc = SOL_OK; continue; }  // (a later candidate failed: the earlier ones stand)
        break;
      }
      // Small scenes: the radii often give the SAME tree (the layout is a function of the tree alone, so equal trees are equal
      // bytes) - a duplicate is not uploaded and probed a second time (the reference's test scene: 33 identical nodes either way).
      bool duplicate = false;
      for (const TreeCand& p : cands) {
        const WideLayout &a = p.lay, &b = c.lay;
        if (p.emin != c.emin || a.nodes.size() != b.nodes.size() || a.leaf_refs != b.leaf_refs) continue;
        if (std::memcmp(a.nodes.data(), b.nodes.data(), a.nodes.size() * sizeof(DWide)) != 0) continue;
        if (a.old_of_new[0] == b.old_of_new[0] && a.old_of_new[1] == b.old_of_new[1] && a.old_of_new[2] == b.old_of_new[2]) { duplicate = true; break; }
      }
      if (duplicate) {
        if (ovr.verbose) std::fprintf(stderr, "[solstrale] device tree (radius %d): the same tree as an earlier candidate, dropped\n", radius);
        continue;
      }
      cands.push_back(std::move(c));
      dev_info.push_back(si);
    }
    prepared.clear();
    s->build_times[2] = seconds_since(t_dev0);
    if (!cands.empty()) {
      rc = SOL_OK;
    } else if (device_explicit) {
      return rc;
    } else {
      fallback_note = std::string("device build failed (") + sol_last_error() + "): host candidates";
      if (ovr.verbose) std::fprintf(stderr, "[solstrale] world tree: %s\n", fallback_note.c_str());
      device_build = false;
      const auto t_host0 = std::chrono::steady_clock::now();
      if ((rc = host_candidates(""))) return rc;
      s->build_times[0] += seconds_since(t_host0);
    }
  }
  auto adopt_info = [&](size_t k) {  // what the chosen device candidate's build did
    if (k >= dev_info.size()) return;
    const DeviceSplitInfo& si = dev_info[k];
    s->split_references = si.extra_references; s->split_triangles = si.split_triangles; s->split_area_ratio = si.area_ratio;
    s->reinsertion_moves = si.reinsertion_moves; s->reinsertion_area_ratio = si.area_before > 0. ? (float)(si.area_after / si.area_before) : 1.f;
  };
  adopt_info(0);
  const bool calibrate = cands.size() > 1;
  // everything that depends on the choice of the world tree: the tree itself, the permuted primitive arrays and every table of
  // references into them (DevTree); candidate 0 first, the others only if a probe has to decide
  const bool need_binary = d->n_mediums > 0;  // the 2-wide tree serves medium boundaries only
  auto upload_tree = [&](TreeCand& c) -> int {
    // (the node test reads plane bytes as the fp16 subnormals q * 2^-24 and keeps the 2^24 in the node scales: sol_trace.h)
    if (c.emin + 31u + 24u > 254u) return sol_fail(SOL_EINVAL, "the scene is too large for the quantised world tree (extent beyond 2^100)");
    const WideLayout& L = c.lay;
    DevTree& t = c.dev;
    std::vector<DTri> ptris(L.old_of_new[0].size());  // (more records than triangles when the device build pre-split some: copies)
    std::vector<DTriShade> pshade(L.old_of_new[0].size());
    std::vector<DQuad> pquads(quads.size());
    std::vector<DSphere> pspheres(spheres.size());
    for (size_t i = 0; i < ptris.size(); ++i) { ptris[i] = tris[L.old_of_new[0][i]]; pshade[i] = tshade[L.old_of_new[0][i]]; }
    for (size_t i = 0; i < spheres.size(); ++i) pspheres[i] = spheres[L.old_of_new[1][i]];
    for (size_t i = 0; i < quads.size(); ++i) pquads[i] = quads[L.old_of_new[2][i]];
    std::vector<DNode> pnodes;
    if (need_binary) {
      pnodes = tb.nodes;
      for (auto& n : pnodes) { n.left = L.remap(n.left); n.right = L.remap(n.right); }
    }
    std::vector<DMedium> pmed = mediums;
    for (auto& m : pmed) m.boundary = L.remap(m.boundary);
    std::vector<uint32_t> plights = lights;
    for (auto& r : plights) r = L.remap(r);
    int e;
    if ((e = sol_upload(L.nodes, &t.wides)) || (e = sol_upload(L.leaf_refs, &t.leaf_refs)) || (e = sol_upload(ptris, &t.tris)) || (e = sol_upload(pshade, &t.tri_shade)) ||
        (e = sol_upload(pquads, &t.quads)) || (e = sol_upload(pspheres, &t.spheres)) || (e = sol_upload(pnodes, &t.nodes)) || (e = sol_upload(pmed, &t.mediums)) ||
        (e = sol_upload(plights, &t.lights))) {
      t.release();
      return e;
    }
    t.emin = c.emin; t.depth = c.depth; t.root = L.remap(root_ref); t.light0 = plights.empty() ? 0u : plights[0];
    t.n_wide = (uint32_t)L.nodes.size(); t.packed_depth = L.depth + medium_depth + 2u;  // (depth_of: 2 * L.depth + medium_depth + 2 with two dwords per group)
    t.old_tri = L.old_of_new[0]; t.old_sphere = L.old_of_new[1]; t.old_quad = L.old_of_new[2];
    return SOL_OK;
  };
  auto adopt_tree = [&](DevTree& t) {  // the scene takes ownership
    s->nodes = t.nodes; s->wides = t.wides; s->leaf_refs = t.leaf_refs; s->tris = t.tris; s->tri_shade = t.tri_shade; s->quads = t.quads;
    s->spheres = t.spheres; s->mediums = t.mediums; s->lights = t.lights;
    s->old_index[0] = std::move(t.old_tri); s->old_index[1] = std::move(t.old_sphere); s->old_index[2] = std::move(t.old_quad);
    DevScene& S = s->S;
    S.nodes = t.nodes; S.wides = t.wides; S.leaf_refs = t.leaf_refs; S.tris = t.tris; S.tri_shade = t.tri_shade; S.quads = t.quads; S.spheres = t.spheres;
    S.mediums = t.mediums; S.lights = t.lights; S.light0 = t.light0; S.wroot = 0; S.wide_emin = t.emin; S.root = t.root;
    s->tree_depth = t.depth; s->n_wide = t.n_wide; s->packed_depth = t.packed_depth;
    t = DevTree{};
  };
  if ((rc = upload_tree(cands[0])) || (rc = sol_upload(mats, &s->mats)) || (rc = sol_upload(texs, &s->texs))) return rc;
  {
    std::vector<uint8_t> texels(d->texels, d->texels + d->n_texel_bytes);
    if ((rc = sol_upload(texels, &s->texels))) return rc;
  }
  HIP_TRY(hipMalloc((void**)&s->work, 64));
  HIP_TRY(hipMalloc((void**)&s->counters, sizeof(DevCounters)));
  HIP_TRY(hipMemset(s->counters, 0, sizeof(DevCounters)));
  HIP_TRY(hipMalloc((void**)&s->image, (size_t)d->width * d->height * 3 * sizeof(float)));
  HIP_TRY(hipMalloc((void**)&s->rgb8, (size_t)d->width * d->height * 3));
  s->build_times[1] = seconds_since(t_upload0) - s->build_times[2];
  const auto t_probe0 = std::chrono::steady_clock::now();

  DevScene& S = s->S;
  S.mats = s->mats; S.texs = s->texs; S.texels = s->texels;
  S.n_lights = d->n_lights;
  adopt_tree(cands[0].dev);
  S.rxmin = root_box.v[0]; S.rxmax = root_box.v[1]; S.rymin = root_box.v[2]; S.rymax = root_box.v[3];
  S.rzmin = root_box.v[4]; S.rzmax = root_box.v[5];
  S.width = d->width; S.height = d->height; S.shader = d->shader_kind; S.max_depth = d->max_depth;
  S.sphere_slack = box_pad * 0.5f;
  if ((rc = sol_upload(light_triangle_frames(*d), &s->light_tri))) return rc;
  S.light_tri = s->light_tri;
  S.tri_delta = scene_has_needles ? box_pad * 0.8f : 0.0f;
  s->strict_triangles = S.tri_delta > 0.0f;
  S.env = nullptr; S.env_w = S.env_h = 0; S.env_scale = 1.0f;
  if (has_env) {
    std::vector<float> env(d->env_texels, d->env_texels + (size_t)d->env_width * d->env_height * 3);
    if ((rc = sol_upload(env, &s->env))) return rc;
    S.env = s->env; S.env_w = d->env_width; S.env_h = d->env_height; S.env_scale = (float)d->env_scale;
  }
  S.bgx = (float)d->background[0]; S.bgy = (float)d->background[1]; S.bgz = (float)d->background[2];
  S.cam = cast_camera(d->camera);
  s->kernel_version = ovr.kernel_version;
  s->pool_swap_min = (uint32_t)ovr.pool_swap_min;
  s->order_mode = ovr.order_mode;
  // v1's search/shade switch (RenderParams::switch_below), measured on MI355X at 1080p x 128 spp (ms, C1 / C2 / C3 / test scene):
  // 0: 29.2 / 162.9 / 236.7 / 25.9, 8: 27.7 / 124.3 / 193.1 / 25.5, 16: 27.5 / 111.8 / 186.9 / 25.6, 24: 28.6 / 108.6 / 192.3 / 26.8.
  s->switch_below = 16u;
  if (ovr.switch_below >= 0) s->switch_below = (uint32_t)ovr.switch_below;
  if (ovr.max_bpc >= 0) s->max_bpc = ovr.max_bpc;  // occupancy experiments
  s->pool_slots_override = (uint32_t)ovr.pool_slots;
  if (ovr.wf_slots > 0) s->wf_slots = (uint32_t)std::max(4096, ovr.wf_slots);
  if (ovr.fine_tail >= -1) s->fine_tail = ovr.fine_tail;
  if (ovr.wf_min_items >= 0) s->wf_min_items = (uint32_t)ovr.wf_min_items;
  s->has_medium = d->n_mediums > 0;
  s->blocks_x = (d->width + SOL_TILE - 1) / SOL_TILE;
  s->blocks_y = (d->height + SOL_TILE - 1) / SOL_TILE;
  if ((rc = sol_set_partition(s, 0, 1))) return rc;
  {  // a null table would be a GPU memory fault at the first launch, not an error code: refuse here
    const void* tables[] = {S.nodes, S.wides, S.leaf_refs, S.tris, S.tri_shade, S.quads, S.spheres, S.mediums, S.mats, S.texs, S.texels, S.lights};
    for (const void* p : tables)
      if (!p) return sol_fail(SOL_EDEVICE, "internal error: a device table of the scene is missing");
  }
  s->tree_name = cands[0].name;
  s->tree_note = fallback_note;
  size_t chosen = 0;  // the candidate the scene keeps
  if (calibrate) {
    // Probe every candidate tree with a counted render of 16 samples per pixel over ~256 pixel blocks spread across the image
    // and keep the one with the least search work (a wide-node visit weighs ~2.5 primitive tests, by instruction count).
    // Images do not depend on the tree, the counters are deterministic, so is the choice.
    auto free_cands = [&]() { for (auto& c : cands) c.dev.release(); };
    for (size_t k = 1; k < cands.size(); ++k)
      if ((rc = upload_tree(cands[k]))) { free_cands(); return rc; }
    const uint32_t nb = s->blocks_x * s->blocks_y;
    rc = sol_set_partition(s, 0, (int)std::max(1u, nb / 256u));
    size_t pick = 0, current = 0;  // `current`: the candidate whose arrays the scene holds at the moment
    auto swap_in = [&](size_t k) {  // hand the scene's tree back to its candidate, adopt candidate k's
      if (k == current) return;
      DevTree& back = cands[current].dev;
      back.nodes = s->nodes; back.wides = s->wides; back.leaf_refs = s->leaf_refs; back.tris = s->tris; back.tri_shade = s->tri_shade;
      back.quads = s->quads; back.spheres = s->spheres; back.mediums = s->mediums; back.lights = s->lights;
      back.emin = S.wide_emin; back.depth = s->tree_depth; back.root = S.root; back.light0 = S.light0; back.n_wide = s->n_wide; back.packed_depth = s->packed_depth;
      back.old_tri = std::move(s->old_index[0]); back.old_sphere = std::move(s->old_index[1]); back.old_quad = std::move(s->old_index[2]);
      adopt_tree(cands[k].dev);
      current = k;
    };
    for (size_t k = 0; k < cands.size() && !rc; ++k) {
      swap_in(k);
      if (!(rc = sol_clear(s)) && !(rc = sol_render_probe(s)))
        cands[k].cost = 2.5 * (double)s->stats.node_visits + (double)(s->stats.sphere_tests + s->stats.quad_tests + s->stats.triangle_tests);
      if (!rc && cands[k].cost < cands[pick].cost) pick = k;
    }
    if (ovr.verbose) {
      std::fprintf(stderr, "[solstrale] world tree probe:");
      for (auto& c : cands) std::fprintf(stderr, " %s %.4g (%zu nodes)", c.name.c_str(), c.cost, c.lay.nodes.size());
      std::fprintf(stderr, " -> %s\n", cands[pick].name.c_str());
    }
    if (hipStreamSynchronize(s->stream) != hipSuccess && !rc) rc = SOL_EDEVICE;
    swap_in(pick);
    chosen = pick;
    s->tree_name = cands[pick].name;
    adopt_info(pick);
    free_cands();
    if (rc) return rc;
    s->stats = SolStats{};
    if ((rc = sol_set_partition(s, 0, 1)) || (rc = sol_clear(s))) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  // Background blocks: constant background only (an environment map is looked up per ray). The proof is host work (0.06 s for C5 at 1080p): it
  // runs on a thread of its own beside the cost probe's render below and is adopted after it (the probe traces every block either way).
  std::future<void> background_proof;
  std::vector<uint8_t> bg_block;
  uint32_t bg_n = 0, bg_pixels = 0;
  const bool want_background = !opt.no_background_blocks && ovr.background_blocks != 0 && !has_env;
  if (want_background) {
    const WideLayout* lay_ptr = &cands[chosen].lay;
    const uint32_t lay_emin = cands[chosen].emin;
    const DCamera cam = S.cam;
    const double inflate = 64.0 * (double)box_pad;
    const uint32_t iw = d->width, ih = d->height;
    background_proof = std::async(std::launch::async, [=, &bg_block, &bg_n, &bg_pixels]() { find_background_blocks(*lay_ptr, lay_emin, cam, iw, ih, inflate, bg_block, bg_n, bg_pixels); });
  }
  auto adopt_background = [&]() -> int {  // (waits for the proof; idempotent)
    if (!background_proof.valid()) return SOL_OK;
    background_proof.get();
    s->background_block = std::move(bg_block); s->n_background = bg_n; s->background_pixels = bg_pixels;
    if (ovr.verbose) std::fprintf(stderr, "[solstrale] background blocks: %u of %u (%u pixels)\n", s->n_background, s->blocks_x * s->blocks_y, s->background_pixels);
    if (s->n_background == 0) { s->background_block.clear(); return SOL_OK; }
    return sol_rebuild_order(s);  // (also without the cost probe below)
  };
  struct ProofJoin { std::future<void>& f; ~ProofJoin() { if (f.valid()) f.wait(); } } proof_join{background_proof};  // (an early return must not leave the thread behind)
  // Cost probe: per 8x8 block, the ray count of the longest 4-sample item in a counted render of the whole frame, for the
  // heavy-first work order
  // (rebuild_order; sol_path.h decode_item_ordered). SOL_ORDER=0 switches it off.
  if (!opt.no_work_order_probe && ovr.order_mode != 0 && s->blocks_x * s->blocks_y >= 64u) {
    const uint32_t nb = s->blocks_x * s->blocks_y;
    uint32_t* cost_dev = nullptr;
    HIP_TRY(hipMalloc((void**)&cost_dev, (size_t)nb * sizeof(uint32_t)));
    hipError_t e = hipMemset(cost_dev, 0, (size_t)nb * sizeof(uint32_t));
    uint32_t* work_dev = nullptr;
    if (e == hipSuccess) e = hipMalloc((void**)&work_dev, (size_t)nb * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(work_dev, 0, (size_t)nb * sizeof(uint32_t));
    S.block_cost = cost_dev;
    S.block_work = work_dev;
    rc = e == hipSuccess ? sol_render_impl(s, 0, 4, 0xC057ull, true) : SOL_EDEVICE;
    if (int rb = adopt_background()) { if (rc == SOL_OK) rc = rb; }
    S.block_cost = nullptr;
    S.block_work = nullptr;
    s->block_cost.assign(nb, 0u);
    s->block_work.assign(nb, 0u);
    if (rc == SOL_OK && hipMemcpy(s->block_work.data(), work_dev, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = SOL_EDEVICE;
    if (work_dev) hipFree(work_dev);
    if (rc == SOL_OK && hipMemcpy(s->block_cost.data(), cost_dev, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = SOL_EDEVICE;
    hipFree(cost_dev);
    if (rc != SOL_OK) return rc == SOL_EDEVICE ? sol_fail(SOL_EDEVICE, "cost probe failed") : rc;
    // The fine tail takes an item fetch per SAMPLE (a dependent load, three integer divisions: ~2 us): worth it where a sample
    // is long. MI355X, 1080p x 64 spp, ms with 0 / 1 / 2 whole items per lane in the tail: C3 (38 node visits per sample) 76.5 /
    // 75.5 / 74.4, C2 (22) 45.5 / 45.7 / 46.3, C1 (2) 10.5 / 11.2 / 11.8.
    // Round 5, re-measured on the round-4 trees and kernel (profiles/r05_fine_tail_sweep.txt; 64 spp, ms with a tail of 0 / 2 / 4 / 8 / 16 whole
    // items per lane): C1 (2 node visits per sample) 9.09 / 9.07 / 9.66 / 9.74 / 9.65, test scene (8) 9.75 / 8.33 / 8.39 / 8.28 / 8.16, C5 (12) 37.9 /
    // 38.2 / 38.8 / 38.2 / 38.3, C2 (19) 33.7 / 32.5 / 31.5 / 31.4 / 31.3, C3 (36) 62.5 / 60.5 / 60.5 / 60.6 / 60.6: the per-sample fetch no longer
    // costs what it did (the reservoir, the work order), a short launch of long paths gains most. By the probe's node visits per sample:
    // below 5 none, 15 .. 30 eight whole items per lane, else two.
    if (s->stats.samples > 0) {
      const double vps = (double)s->stats.node_visits / (double)s->stats.samples;
      s->fine_tail_auto = vps < 5.0 ? 0 : (vps >= 15.0 && vps < 30.0) ? 32 : 8;
    }
    s->stats = SolStats{};
    if ((rc = sol_clear(s)) || (rc = sol_rebuild_order(s))) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (ovr.verbose) std::fprintf(stderr, "[solstrale] work order: %u of %u blocks heavy (first)\n", S.n_first, s->n_local_blocks);
  }
  if ((rc = adopt_background())) return rc;  // (no cost probe ran: the proof is adopted here)
  s->build_times[3] = seconds_since(t_probe0);
  cleanup.keep = true;
  *out = s;
  return SOL_OK;
}

int sol_scene_build_times(const SolScene* s, double out[4]) {
  if (!s || !out) return sol_fail(SOL_EINVAL, "null argument");
  for (int k = 0; k < 4; ++k) out[k] = s->build_times[k];
  return SOL_OK;
}

}  // extern "C"
