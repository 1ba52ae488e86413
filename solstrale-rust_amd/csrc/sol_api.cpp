// sol_api.cpp -- implementation of the C ABI (include/solstrale_hip.h): validation of the flattened scene,
// conversion to the fp32 device layout (sol_types.h), upload, and launches of the kernels in sol_render.hip / sol_wavefront.hip / sol_aux.hip / sol_build.hip.
// There is NO CPU fallback: without a HIP device every compute entry point fails with SOL_EDEVICE.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is dlopen-ed on the first sol_comm_* call
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/solstrale_hip.h"
#include "sol_build.h"
#include "sol_launch.h"
#include "sol_tree.h"
#include "sol_types.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(SOL_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

// Box, TreeBuilder, WideBuilder, SahBuilder: sol_tree.h (included above)

template <typename T>
int upload(const std::vector<T>& host, T** dev) {
  *dev = nullptr;
  size_t bytes = std::max<size_t>(host.size() * sizeof(T), 64);  // never a null device pointer
  HIP_TRY(hipMalloc((void**)dev, bytes));
  HIP_TRY(hipMemset(*dev, 0, bytes));
  if (!host.empty()) HIP_TRY(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  return SOL_OK;
}

}  // namespace

#define SOL_MAX_ITEMS 0xFF000000ull

// Device memory that depends on the choice of the world tree (sol_scene_create probes several candidates): the 7-wide tree, the
// primitive arrays in that tree's leaf order and every table holding references into them.
struct DevTree {
  DWide* wides = nullptr; uint32_t* leaf_refs = nullptr; DTri* tris = nullptr; DTriShade* tri_shade = nullptr; DQuad* quads = nullptr;
  DSphere* spheres = nullptr; DNode* nodes = nullptr; DMedium* mediums = nullptr; uint32_t* lights = nullptr;
  uint32_t emin = 1, depth = 0, root = 0, light0 = 0;
  std::vector<uint32_t> old_tri, old_sphere, old_quad;  // device index -> index in the caller's arrays
  void release() {
    void* p[] = {wides, leaf_refs, tris, tri_shade, quads, spheres, nodes, mediums, lights};
    for (void* q : p) if (q) hipFree(q);
    *this = DevTree{};
  }
};

struct SolScene {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  DevScene S{};
  DevScene* dscene = nullptr; DevScene S_uploaded{}; bool dscene_valid = false;  // device copy of S (the v1 kernel reads it through a pointer)
  // owned device buffers
  std::vector<uint32_t> old_index[3];  // triangles / spheres / quads: device index -> index in the caller's SolSceneDesc arrays
  std::string tree_name;               // which world tree the handle walks ("ref", "sah8", .., "device")
  uint32_t* leaf_refs = nullptr;
  DNode* nodes = nullptr; DWide* wides = nullptr; DTri* tris = nullptr; DTriShade* tri_shade = nullptr; DQuad* quads = nullptr;
  DSphere* spheres = nullptr; DMedium* mediums = nullptr; DMat* mats = nullptr; DTex* texs = nullptr;
  uint8_t* texels = nullptr; uint32_t* lights = nullptr; float* env = nullptr;
  float* acc_own = nullptr; float* acc = nullptr; size_t acc_floats = 0;
  float* aux[2] = {nullptr, nullptr}; size_t aux_floats = 0;
  std::vector<uint32_t> block_cost;  // per 8x8 block (global index): rays of its longest item in the cost probe; empty: no ordering
  uint32_t* order_dev = nullptr; size_t order_cap = 0;  // DevScene::block_order of the current partition  // albedo / normal accumulators (sol_render_aux), same layout as acc
  float* partial = nullptr; size_t partial_floats = 0;
  int fine_tail = -1;                // SOL_OPT_FINE_TAIL / SOL_FINE_TAIL: quarters of a whole item per resident lane that the end of a launch hands
                                     // out sample by sample; 0: none; -1: by the creation probe's node visits per sample (fine_tail_auto)
  int fine_tail_auto = 0;
  float* image = nullptr;  // W*H*3 scratch for sol_read / sol_resolve_image
  uint8_t* rgb8 = nullptr;
  double* bloom_a = nullptr; double* bloom_b = nullptr; double* bloom_w = nullptr; size_t bloom_w_cap = 0;  // sol_bloom scratch
  uint32_t* work = nullptr; uint32_t* spill = nullptr; size_t spill_words = 0;
  DevCounters* counters = nullptr;
  SolStats stats{};
  bool has_medium = false;
  uint32_t tree_depth = 0;
  int rank = 0, world = 1;
  uint32_t blocks_x = 0, blocks_y = 0, n_local_blocks = 0;
  int n_cu = 0;
  int kernel_version = 0;          // 0 auto; SOL_KERNEL=v1|v2|v3 forces one (A/B comparisons)
  void* pool = nullptr; size_t pool_bytes = 0;  // path-slot pool of the wavefront kernels
  uint32_t pool_slots_override = 0;  // SOL_POOL_SLOTS (v2: slots per wave)
  uint32_t switch_below = 0;         // SOL_SWITCH (v1, RenderParams::switch_below)
  uint32_t* queue = nullptr; size_t queue_slots = 0;  // v3 ray queue
  void* wf_ctr = nullptr; uint32_t* wf_ctr_host = nullptr;
  uint32_t wf_slots = 4u << 20;       // SOL_WF_SLOTS: pool size of the two-kernel wavefront
  uint32_t wf_min_items = 2u << 20;   // SOL_WF_MIN_ITEMS: jobs below this use the single-launch kernel
  uint32_t last_rounds = 0; int last_version = 0;
  double build_times[4] = {0., 0., 0., 0.};  // sol_scene_build_times
  bool order_enabled = true;         // SOL_OPT_WORK_ORDER
  int max_bpc = 0;                   // SOL_OPT_MAX_BLOCKS_PER_CU (0 = what the occupancy query allows)
  // multi-GPU (sol_comm_init): RCCL communicator of the tile partition and rank 0's receive buffer
  void* comm = nullptr; float* gathered = nullptr; size_t gathered_floats = 0;
  bool timing = false;  // sol_kernel_timing: HIP events around the render kernel on its own stream
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  uint32_t timed_launches = 0, last_grid = 0;
};

// DevScene::block_order for the current partition from the probe's per-block costs: the blocks more than three times as
// costly as the average come first, costliest first; everything else keeps its order. No heavy blocks: identity (null).
static int rebuild_order(SolScene* s) {
  s->S.block_order = nullptr;
  s->S.n_first = 0;
  const uint32_t n = s->n_local_blocks;
  if (!s->order_enabled || s->block_cost.empty() || n < 2) return SOL_OK;
  std::vector<uint32_t> cost(n);
  double sum = 0.;
  for (uint32_t lb = 0; lb < n; ++lb) {
    const size_t b = (size_t)lb * s->world + s->rank;
    cost[lb] = b < s->block_cost.size() ? s->block_cost[b] : 0u;
    sum += cost[lb];
  }
  const double limit = 3.0 * sum / n;
  std::vector<uint32_t> heavy, rest;
  for (uint32_t lb = 0; lb < n; ++lb) (cost[lb] > limit ? heavy : rest).push_back(lb);
  // The rest keeps the chunk-major order, but within a chunk the blocks go from costly to cheap in eight cost classes (octiles of
  // the probe's ray counts; raster order inside a class, so neighbours stay together): the launch then ENDS on the cheapest blocks
  // (sky) of the last chunk instead of on whatever the raster order ends on (floor and walls: one 16-sample item of a long path
  // is milliseconds of one lane). A fixed ~6 ms of tail per launch otherwise - 1 % of a 1080p x 512 spp frame on one GPU, 7 % of
  // its eighth on eight.
  static const int order_mode = std::getenv("SOL_ORDER") ? std::atoi(std::getenv("SOL_ORDER")) : 2;  // 1: heavy-first only (round 1)
  if (order_mode >= 2 && rest.size() >= 64) {
    std::vector<uint32_t> sorted_cost;
    sorted_cost.reserve(rest.size());
    for (uint32_t lb : rest) sorted_cost.push_back(cost[lb]);
    std::sort(sorted_cost.begin(), sorted_cost.end());
    uint32_t edge[7];
    for (int k = 0; k < 7; ++k) edge[k] = sorted_cost[(size_t)(k + 1) * sorted_cost.size() / 8];
    auto cls = [&](uint32_t lb) { int c = 0; while (c < 7 && cost[lb] >= edge[c]) ++c; return c; };
    std::stable_sort(rest.begin(), rest.end(), [&](uint32_t a, uint32_t b) { return cls(a) > cls(b); });
  } else if (heavy.empty() || rest.empty()) {
    return SOL_OK;
  }
  std::stable_sort(heavy.begin(), heavy.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
  heavy.insert(heavy.end(), rest.begin(), rest.end());
  if (n > s->order_cap) {
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->order_dev) hipFree(s->order_dev);
    s->order_dev = nullptr; s->order_cap = 0;
    HIP_TRY(hipMalloc((void**)&s->order_dev, (size_t)n * sizeof(uint32_t)));
    s->order_cap = n;
  }
  HIP_TRY(hipStreamSynchronize(s->stream));  // a launch in flight may still read the old table
  HIP_TRY(hipMemcpy(s->order_dev, heavy.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
  s->S.block_order = s->order_dev;
  s->S.n_first = (uint32_t)(heavy.size() - rest.size());
  return SOL_OK;
}

static int set_partition(SolScene* s, int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world) return fail(SOL_EINVAL, "bad partition %d/%d", rank, world);
  s->rank = rank; s->world = world;
  const uint32_t nb = s->blocks_x * s->blocks_y;
  s->n_local_blocks = (nb + (uint32_t)world - 1u - (uint32_t)rank) / (uint32_t)world;  // blocks b with b % world == rank
  // every rank's compact buffer has the size of rank 0's (the largest) so that a gather has equal counts
  const uint32_t max_blocks = (nb + (uint32_t)world - 1u) / (uint32_t)world;
  size_t floats = (size_t)max_blocks * 64u * 3u;
  if (floats != s->acc_floats || !s->acc_own) {
    if (s->acc_own) { hipFree(s->acc_own); s->acc_own = nullptr; }
    HIP_TRY(hipMalloc((void**)&s->acc_own, std::max<size_t>(floats * sizeof(float), 64)));
    HIP_TRY(hipMemset(s->acc_own, 0, std::max<size_t>(floats * sizeof(float), 64)));
    s->acc = s->acc_own;
    s->acc_floats = floats;
  }
  return rebuild_order(s);
}


// The world tree built on the GPU (sol_build.hip): primitives of the reference-shaped tree under `root_ref` (each once - a
// shared sub-tree is the same geometry twice, one copy finds the same hits), clustered and collapsed on the current device.
static int device_world_tree(const std::vector<DNode>& bin, uint32_t root_ref, const Box& root_box, float box_pad, const uint32_t counts[3],
                             hipStream_t stream, WideLayout& lay, uint32_t& emin) {
  std::vector<SolBuildPrim> prims;
  if (SOL_REF_KIND(root_ref) == SOL_REF_NODE) {
    SahBuilder col;
    if (!col.collect(bin, root_ref)) return fail(SOL_EINVAL, "the world's primitives cannot be collected (non-finite box or fewer than two)");
    std::sort(col.prims.begin(), col.prims.end(), [](const SahBuilder::Prim& a, const SahBuilder::Prim& b) { return a.ref < b.ref; });
    prims.reserve(col.prims.size());
    for (size_t i = 0; i < col.prims.size(); ++i) {
      if (i && col.prims[i].ref == col.prims[i - 1].ref) continue;
      SolBuildPrim p;
      for (int k = 0; k < 6; ++k) p.box[k] = col.prims[i].box.v[k];
      p.ref = col.prims[i].ref; p.pad = 0;
      prims.push_back(p);
    }
  } else {
    SolBuildPrim p;
    for (int k = 0; k < 6; ++k) p.box[k] = root_box.v[k];
    p.ref = root_ref; p.pad = 0;
    prims.push_back(p);
  }
  emin = WideBuilder::exponent_min(root_box, box_pad);
  SolDeviceTree dt;
  std::string err;
  if (!sol_build_world_tree_device(prims.data(), (uint32_t)prims.size(), root_box.v, box_pad, emin, counts, stream, dt, err)) return fail(SOL_EDEVICE, "%s", err.c_str());
  if (!lay.adopt_device(std::move(dt.nodes), std::move(dt.leaf_refs), dt.new_of_old, dt.depth)) return fail(SOL_EDEVICE, "%s", lay.error.c_str());
  return SOL_OK;
}

static int render_probe(SolScene* s);
static int render_impl(SolScene* s, uint32_t first, uint32_t n, uint64_t seed, bool count);

extern "C" {

const char* sol_last_error(void) { return g_err.c_str(); }

int sol_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sol_world_tree_check(const SolSceneDesc* d, int use_sah, SolTreeCheck* out) {
  if (!d || !out) return fail(SOL_EINVAL, "null argument");
  std::memset(out, 0, sizeof *out);
  const float box_pad = box_pad_for(*d);
  TreeBuilder tb(*d, box_pad);
  uint32_t root_ref;
  Box root_box;
  if (!tb.resolve(d->root, 0, root_ref, root_box)) return fail(SOL_EINVAL, "world: %s", tb.error.c_str());
  if (SOL_REF_KIND(root_ref) != SOL_REF_NODE) return fail(SOL_EINVAL, "the world is a single primitive: no tree");
  SahBuilder sah;
  if (!sah.collect(tb.nodes, root_ref)) return fail(SOL_EINVAL, "the world's primitives cannot be collected (non-finite box or fewer than two)");
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
  if (use_sah < 0) {  // the tree sol_build.hip builds on the GPU, checked like the host-built ones
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SOL_EDEVICE, "no HIP device available");
    HIP_TRY(hipSetDevice(0));
    const uint32_t counts[3] = {d->n_triangles, d->n_spheres, d->n_quads};
    int rc = device_world_tree(tb.nodes, root_ref, root_box, box_pad, counts, nullptr, lay, emin_used);
    if (rc) return rc;
  } else {
    uint32_t bin_root = root_ref;
    if (use_sah) { Box b; sah.BINS = use_sah > 1 ? std::min((int)SahBuilder::MAX_BINS, use_sah) : 16; bin_root = sah.build(0, sah.prims.size(), 0, b); }
    WideBuilder wb(use_sah ? sah.nodes : tb.nodes, box_pad);
    wb.dp_collapse = !(std::getenv("SOL_COLLAPSE") && std::strcmp(std::getenv("SOL_COLLAPSE"), "greedy") == 0);
    wb.set_exponent_range(root_box);
    const uint32_t xroot = wb.build(SOL_REF_INDEX(bin_root), 0);
    if (wb.range_error || !lay.run(wb.out, SOL_REF_INDEX(xroot), wb.emin, d->n_triangles, d->n_spheres, d->n_quads))
      return fail(SOL_EINVAL, "wide tree layout: %s", wb.range_error ? "exponent range" : lay.error.c_str());
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
        if (a >= 0) ref = SOL_REF_INDEX(ref) < lay.old_of_new[a].size() ? SOL_MAKE_REF(SOL_REF_KIND(ref), lay.old_of_new[a][SOL_REF_INDEX(ref)]) : 0u;
        found[ref]++;
        out->n_leaf_refs++;
        auto it = prim_box.find(ref);
        below = it == prim_box.end() ? empty_box() : it->second;
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
  // the permutations must be permutations
  for (int a = 0; a < 3; ++a) {
    std::vector<uint8_t> seen(lay.old_of_new[a].size(), 0);
    for (uint32_t o : lay.old_of_new[a]) { if (o >= seen.size() || seen[o]) out->leaf_mismatches++; else seen[o] = 1; }
    if (lay.old_of_new[a].size() != lay.new_of_old[a].size()) out->leaf_mismatches++;
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

int sol_record_sizes(uint32_t out[6]) {
  out[0] = SOL_WORLD_BINARY ? sizeof(DNode) : sizeof(DWide); out[1] = sizeof(DSphere); out[2] = sizeof(DQuad); out[3] = sizeof(DTri);
  out[4] = sizeof(DTriShade); out[5] = sizeof(DMat);
  return SOL_OK;
}

void sol_scene_destroy(SolScene* s) {
  if (!s) return;
  hipSetDevice(s->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  sol_comm_destroy(s);
  void* ptrs[] = {s->leaf_refs, s->nodes, s->wides, s->tris, s->tri_shade, s->quads, s->spheres, s->mediums, s->mats, s->texs, s->texels, s->env, s->lights,
                  s->acc_own, s->partial, s->image, s->rgb8, s->work, s->spill, s->counters, s->pool, s->queue, s->wf_ctr,
                  s->bloom_a, s->bloom_b, s->bloom_w, s->aux[0], s->aux[1], s->dscene, s->order_dev};
  if (s->wf_ctr_host) hipHostFree(s->wf_ctr_host);
  for (void* p : ptrs)
    if (p) hipFree(p);
  if (s->ev_start) hipEventDestroy(s->ev_start);
  if (s->ev_stop) hipEventDestroy(s->ev_stop);
  if (s->own_stream) hipStreamDestroy(s->own_stream);
  delete s;
}

int sol_scene_create(const SolSceneDesc* d, int device, SolScene** out) { return sol_scene_create_ex(d, device, nullptr, out); }

int sol_scene_create_ex(const SolSceneDesc* d, int device, const SolCreateOptions* opt_in, SolScene** out) {
  if (!d || !out) return fail(SOL_EINVAL, "null argument");
  *out = nullptr;
  SolCreateOptions opt{};
  if (opt_in) {
    if (opt_in->size < 8 || opt_in->size > 4096) return fail(SOL_EINVAL, "SolCreateOptions.size %u", opt_in->size);
    std::memcpy(&opt, opt_in, std::min<size_t>(opt_in->size, sizeof opt));
  }
  if (opt.world_tree < SOL_TREE_AUTO || opt.world_tree > SOL_TREE_HOST_PROBE) return fail(SOL_EINVAL, "bad world_tree option %d", opt.world_tree);
  const auto t_begin = std::chrono::steady_clock::now();
  auto seconds_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  if (d->abi_version != SOL_ABI_VERSION && d->abi_version != 1u) return fail(SOL_EINVAL, "abi_version %u, expected %u (or 1)", d->abi_version, SOL_ABI_VERSION);
  const bool has_env = d->abi_version >= 2u && d->env_texels && d->env_width && d->env_height;  // (a version-1 description ends before these fields)
  if (has_env && ((uint64_t)d->env_width * d->env_height > (1ull << 28) || !std::isfinite(d->env_scale))) return fail(SOL_EINVAL, "bad environment map");
  if (d->width < 2 || d->height < 2 || (uint64_t)d->width * d->height > 0x3FFFFFFFull) return fail(SOL_EINVAL, "bad image size %ux%u", d->width, d->height);
  if (d->shader_kind > SOL_SHADER_SIMPLE) return fail(SOL_EINVAL, "bad shader kind %u", d->shader_kind);
  if ((d->n_nodes && !d->nodes) || (d->n_spheres && !d->spheres) || (d->n_quads && !d->quads) ||
      (d->n_triangles && !d->triangles) || (d->n_mediums && !d->mediums) || (d->n_materials && !d->materials) ||
      (d->n_textures && !d->textures) || (d->n_texel_bytes && !d->texels) || (d->n_lights && !d->lights))
    return fail(SOL_EINVAL, "null array with non-zero count");
  // Renderer::new: "Scene should have at least one light" (src/renderer/mod.rs:143-147)
  if (d->n_lights == 0) return fail(SOL_ENOLIGHT, "Scene should have at least one light");
  if (d->n_texel_bytes > 0xFFFFFFF0ull) return fail(SOL_EINVAL, "more than 4 GiB of texels");

  // ---- materials / textures ----
  std::vector<DTex> texs(d->n_textures);
  for (uint32_t i = 0; i < d->n_textures; ++i) {
    const SolTexture& t = d->textures[i];
    DTex& o = texs[i];
    std::memset(&o, 0, sizeof o);
    o.kind = t.kind;
    if (t.kind == SOL_TEX_IMAGE) {
      if (!t.width || !t.height || t.texel_offset + (uint64_t)t.width * t.height * 3 > d->n_texel_bytes)
        return fail(SOL_EINVAL, "texture %u: image outside texel buffer", i);
      o.w = t.width; o.h = t.height; o.offset = (uint32_t)t.texel_offset;
    } else if (t.kind == SOL_TEX_SOLID) {
      o.r = (float)t.rgb[0]; o.g = (float)t.rgb[1]; o.b = (float)t.rgb[2];
    } else {
      return fail(SOL_EINVAL, "texture %u: bad kind %d", i, t.kind);
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
        if (!tex_ok(m.albedo_tex, false) || !tex_ok(m.normal_tex, true)) return fail(SOL_EINVAL, "material %u: bad texture id", i);
        break;
      case SOL_MAT_DIFFUSE_LIGHT: case SOL_MAT_ISOTROPIC:
        if (!tex_ok(m.albedo_tex, false)) return fail(SOL_EINVAL, "material %u: bad texture id", i);
        o.normal = -1;
        break;
      case SOL_MAT_BLEND:
        if (m.m1 < 0 || m.m2 < 0 || (uint32_t)m.m1 >= d->n_materials || (uint32_t)m.m2 >= d->n_materials || (uint32_t)m.m1 == i || (uint32_t)m.m2 == i)
          return fail(SOL_EINVAL, "material %u: bad blend children", i);
        break;
      default: return fail(SOL_EINVAL, "material %u: bad kind %d", i, m.kind);
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
  for (uint32_t i = 0; i < d->n_triangles; ++i) {
    const SolTriangle& t = d->triangles[i];
    if (!mat_ok(t.material)) return fail(SOL_EINVAL, "triangle %u: bad material", i);
    DTri& o = tris[i];
    o.v0x = (float)t.v0[0]; o.v0y = (float)t.v0[1]; o.v0z = (float)t.v0[2];
    o.e1x = (float)t.v0v1[0]; o.e1y = (float)t.v0v1[1]; o.e1z = (float)t.v0v1[2];
    o.e2x = (float)t.v0v2[0]; o.e2y = (float)t.v0v2[1]; o.e2z = (float)t.v0v2[2];
    o.dfs = t.dfs_index; o.mat = t.material; o.area = (float)t.area;
    DTriShade& s = tshade[i];
    s.nx = (float)t.normal[0]; s.ny = (float)t.normal[1]; s.nz = (float)t.normal[2]; s.mat = t.material;
    s.tx = (float)t.tangent[0]; s.ty = (float)t.tangent[1]; s.tz = (float)t.tangent[2];
    s.bx = (float)t.bi_tangent[0]; s.by = (float)t.bi_tangent[1]; s.bz = (float)t.bi_tangent[2];
    s.u0 = t.uv0[0]; s.v0 = t.uv0[1]; s.u1 = t.uv1[0]; s.v1 = t.uv1[1]; s.u2 = t.uv2[0]; s.v2 = t.uv2[1];
  }
  std::vector<DQuad> quads(d->n_quads);
  for (uint32_t i = 0; i < d->n_quads; ++i) {
    const SolQuad& q = d->quads[i];
    if (!mat_ok(q.material)) return fail(SOL_EINVAL, "quad %u: bad material", i);
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
    if (!mat_ok(s.material)) return fail(SOL_EINVAL, "sphere %u: bad material", i);
    DSphere& o = spheres[i];
    o.cx = (float)s.center[0]; o.cy = (float)s.center[1]; o.cz = (float)s.center[2]; o.radius = (float)s.radius;
    o.dfs = s.dfs_index; o.mat = s.material; o.pad0 = o.pad1 = 0;
  }

  // ---- tree ----
  const float box_pad = box_pad_for(*d);
  TreeBuilder tb(*d, box_pad);
  uint32_t root_ref;
  Box root_box;
  if (!tb.resolve(d->root, 0, root_ref, root_box)) return fail(SOL_EINVAL, "world: %s", tb.error.c_str());
  if (SOL_REF_KIND(root_ref) == SOL_REF_NONE) return fail(SOL_EINVAL, "world is empty");
  const uint32_t world_depth = tb.max_depth;
  std::vector<DMedium> mediums(d->n_mediums);
  uint32_t medium_depth = 0;
  for (uint32_t i = 0; i < d->n_mediums; ++i) {
    const SolMedium& m = d->mediums[i];
    if (!mat_ok(m.material)) return fail(SOL_EINVAL, "medium %u: bad material", i);
    if (i >= 0x1000u) return fail(SOL_EINVAL, "more than 4096 constant mediums");
    DMedium& o = mediums[i];
    std::memset(&o, 0, sizeof o);
    tb.max_depth = 0;
    uint32_t bref;
    Box bb;
    if (!tb.resolve(m.boundary, 0, bref, bb)) return fail(SOL_EINVAL, "medium %u boundary: %s", i, tb.error.c_str());
    if (SOL_REF_KIND(bref) == SOL_REF_MEDIUM || SOL_REF_KIND(bref) == SOL_REF_NONE) return fail(SOL_EINVAL, "medium %u: unsupported boundary", i);
    medium_depth = std::max(medium_depth, tb.max_depth);
    o.boundary = bref; o.mat = m.material; o.nid = (float)m.negative_inverse_density; o.dfs = m.dfs_index;
    o.bxmin = bb.v[0]; o.bxmax = bb.v[1]; o.bymin = bb.v[2]; o.bymax = bb.v[3]; o.bzmin = bb.v[4]; o.bzmax = bb.v[5];
  }
  // a medium inside a medium boundary would recurse in the device search: reject (never built by the reference's scenes)
  for (uint32_t i = 0; i < d->n_mediums; ++i) {
    std::vector<uint32_t> stk{mediums[i].boundary};
    while (!stk.empty()) {
      uint32_t r = stk.back(); stk.pop_back();
      if (SOL_REF_KIND(r) == SOL_REF_MEDIUM) return fail(SOL_EINVAL, "medium %u: nested ConstantMedium in a boundary is unsupported", i);
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
  auto depth_of = [&](const WideLayout& l) { return (SOL_WORLD_BINARY ? world_depth : 2u * l.depth) + medium_depth + 2; };
  const uint32_t stack_limit = SOL_LDS_STACK + SOL_SPILL_STACK;
  const char* bvh_env = std::getenv("SOL_BVH");  // developer override of SolCreateOptions.world_tree
  // AUTO = the device build: as good a tree as the probed host candidates (node visits per ray, host probe / device: C2 11.0 /
  // 10.9, C3 12.8 / 13.0, C5 6.8 / 6.9) in a sixth to an eighth of the time (sol_scene_create, C3: 0.40 s -> 0.06 s, C5 2.2 s -> 0.3 s)
  static const char* const tree_names[] = {"device", "ref", "sah8", "sah16", "sah64", "device", ""};
  std::string want = bvh_env ? (std::strcmp(bvh_env, "sah") == 0 ? "sah16" : bvh_env) : tree_names[opt.world_tree];
  if (want == "host") want = "";  // all host candidates + the probe
  const bool greedy = std::getenv("SOL_COLLAPSE") && std::strcmp(std::getenv("SOL_COLLAPSE"), "greedy") == 0;
  auto finish_cand = [&](TreeCand& c, uint32_t wide_root) {  // explicit tree -> device layout
    if (c.wb->range_error || !c.lay.run(c.wb->out, SOL_REF_INDEX(wide_root), c.wb->emin, d->n_triangles, d->n_spheres, d->n_quads)) {
      c.wb.reset();
      return;
    }
    c.depth = depth_of(c.lay);
    c.emin = c.wb->emin;
  };
  std::string layout_error;
  const bool device_build = want == "device";
  if (device_build) {
    // built below, once the device is set up (sol_build.hip): no host candidates, no tree probe
  } else if (SOL_REF_KIND(root_ref) == SOL_REF_NODE) {
    {
      TreeCand c;
      c.name = "ref";
      c.wb.reset(new WideBuilder(tb.nodes, box_pad));
      c.wb->dp_collapse = !greedy;
      c.wb->set_exponent_range(root_box);
      finish_cand(c, c.wb->build(SOL_REF_INDEX(root_ref), 0));
      if (!c.wb) layout_error = c.lay.error.empty() ? "wide tree: exponent range" : c.lay.error;
      else cands.push_back(std::move(c));
    }
    std::vector<int> bin_list = {8, 16, 64};
    if (const char* bl = std::getenv("SOL_SAH_LIST")) {  // experiment: other candidate sets, e.g. SOL_SAH_LIST=4,12,32
      bin_list.clear();
      for (const char* p = bl; *p;) { bin_list.push_back(std::max(2, std::min(64, std::atoi(p)))); while (*p && *p != ',') ++p; if (*p) ++p; }
    }
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
        c.wb->dp_collapse = !greedy;
        c.wb->set_exponent_range(root_box);
        finish_cand(c, c.wb->build(SOL_REF_INDEX(r), 0));
        return c;
      }));
    }
    for (auto& j : jobs) {
      TreeCand c = j.get();
      if (c.wb) cands.push_back(std::move(c));
    }
    if (cands.empty()) return fail(SOL_EINVAL, "world: %s", layout_error.c_str());
    // drop what cannot run; a forced choice drops the rest
    std::vector<TreeCand> keep;
    for (auto& c : cands)
      if (c.depth <= stack_limit && (want.empty() || c.name == want || (want != "ref" && c.name == "ref" && cands.size() == 1))) keep.push_back(std::move(c));
    if (keep.empty()) {
      uint32_t dmin = 0xFFFFFFFFu;
      for (auto& c : cands) if (c.wb) dmin = std::min(dmin, c.depth);
      return fail(SOL_EDEPTH, "BVH depth %u exceeds the traversal stack (%d)", dmin, stack_limit);
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
    if (!c.wb) return fail(SOL_EINVAL, "world: %s", c.lay.error.c_str());
    cands.push_back(std::move(c));
  }
  const bool calibrate = cands.size() > 1 && !SOL_WORLD_BINARY && !device_build;

  const double t_host_trees = seconds_since(t_begin);

  // ---- lights ----
  std::vector<uint32_t> lights(d->lights, d->lights + d->n_lights);
  for (uint32_t i = 0; i < d->n_lights; ++i) {
    uint32_t k = SOL_REF_KIND(lights[i]), x = SOL_REF_INDEX(lights[i]);
    bool ok = (k == SOL_REF_SPHERE && x < d->n_spheres) || (k == SOL_REF_QUAD && x < d->n_quads) || (k == SOL_REF_TRIANGLE && x < d->n_triangles);
    if (!ok) return fail(SOL_EINVAL, "light %u: not a sphere/quad/triangle reference", i);
  }

  // ---- device ----
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SOL_EDEVICE, "no HIP device available");
  if (device < 0 || device >= ndev) return fail(SOL_EDEVICE, "device %d out of range (%d devices)", device, ndev);
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
  if (device_build) {
    const auto t_dev0 = std::chrono::steady_clock::now();
    TreeCand c;
    c.name = "device";
    const uint32_t counts[3] = {d->n_triangles, d->n_spheres, d->n_quads};
    if ((rc = device_world_tree(tb.nodes, root_ref, root_box, box_pad, counts, s->stream, c.lay, c.emin))) return rc;
    c.depth = depth_of(c.lay);
    if (c.depth > stack_limit) return fail(SOL_EDEPTH, "BVH depth %u exceeds the traversal stack (%d)", c.depth, stack_limit);
    cands.push_back(std::move(c));
    s->build_times[2] = seconds_since(t_dev0);
  }
  // everything that depends on the choice of the world tree: the tree itself, the permuted primitive arrays and every table of
  // references into them (DevTree); candidate 0 first, the others only if a probe has to decide
  const bool need_binary = SOL_WORLD_BINARY || d->n_mediums > 0;  // the 2-wide tree serves medium boundaries (and the A/B build) only
  auto upload_tree = [&](TreeCand& c) -> int {
    const WideLayout& L = c.lay;
    DevTree& t = c.dev;
    std::vector<DTri> ptris(tris.size());
    std::vector<DTriShade> pshade(tshade.size());
    std::vector<DQuad> pquads(quads.size());
    std::vector<DSphere> pspheres(spheres.size());
    for (size_t i = 0; i < tris.size(); ++i) { ptris[i] = tris[L.old_of_new[0][i]]; pshade[i] = tshade[L.old_of_new[0][i]]; }
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
    if ((e = upload(L.nodes, &t.wides)) || (e = upload(L.leaf_refs, &t.leaf_refs)) || (e = upload(ptris, &t.tris)) || (e = upload(pshade, &t.tri_shade)) ||
        (e = upload(pquads, &t.quads)) || (e = upload(pspheres, &t.spheres)) || (e = upload(pnodes, &t.nodes)) || (e = upload(pmed, &t.mediums)) ||
        (e = upload(plights, &t.lights))) {
      t.release();
      return e;
    }
    t.emin = c.emin; t.depth = c.depth; t.root = L.remap(root_ref); t.light0 = plights.empty() ? 0u : plights[0];
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
    s->tree_depth = t.depth;
    t = DevTree{};
  };
  if ((rc = upload_tree(cands[0])) || (rc = upload(mats, &s->mats)) || (rc = upload(texs, &s->texs))) return rc;
  {
    std::vector<uint8_t> texels(d->texels, d->texels + d->n_texel_bytes);
    if ((rc = upload(texels, &s->texels))) return rc;
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
  S.env = nullptr; S.env_w = S.env_h = 0; S.env_scale = 1.0f;
  if (has_env) {
    std::vector<float> env(d->env_texels, d->env_texels + (size_t)d->env_width * d->env_height * 3);
    if ((rc = upload(env, &s->env))) return rc;
    S.env = s->env; S.env_w = d->env_width; S.env_h = d->env_height; S.env_scale = (float)d->env_scale;
  }
  S.bgx = (float)d->background[0]; S.bgy = (float)d->background[1]; S.bgz = (float)d->background[2];
  const SolCamera& c = d->camera;
  S.cam = DCamera{(float)c.origin[0], (float)c.origin[1], (float)c.origin[2],
                  (float)c.lower_left_corner[0], (float)c.lower_left_corner[1], (float)c.lower_left_corner[2],
                  (float)c.horizontal[0], (float)c.horizontal[1], (float)c.horizontal[2],
                  (float)c.vertical[0], (float)c.vertical[1], (float)c.vertical[2],
                  (float)c.u[0], (float)c.u[1], (float)c.u[2], (float)c.v[0], (float)c.v[1], (float)c.v[2],
                  (float)c.lens_radius};
  if (const char* kv = std::getenv("SOL_KERNEL")) {
    if (kv[0] == 'v') kv++;
    s->kernel_version = (kv[0] >= '1' && kv[0] <= '3') ? kv[0] - '0' : 0;
  }
  // v1's search/shade switch (RenderParams::switch_below), measured on MI355X at 1080p x 128 spp (ms, C1 / C2 / C3 / test scene):
  // 0: 29.2 / 162.9 / 236.7 / 25.9, 8: 27.7 / 124.3 / 193.1 / 25.5, 16: 27.5 / 111.8 / 186.9 / 25.6, 24: 28.6 / 108.6 / 192.3 / 26.8.
  s->switch_below = 16u;
  if (const char* ps = std::getenv("SOL_SWITCH")) s->switch_below = (uint32_t)std::min(64, std::max(0, std::atoi(ps)));
  if (const char* mb = std::getenv("SOL_MAX_BPC")) s->max_bpc = std::max(0, std::atoi(mb));  // occupancy experiments
  if (const char* ps = std::getenv("SOL_POOL_SLOTS")) s->pool_slots_override = (uint32_t)std::atoi(ps);
  if (const char* ps = std::getenv("SOL_WF_SLOTS")) s->wf_slots = std::max(4096, std::atoi(ps));
  if (const char* ps = std::getenv("SOL_FINE_TAIL")) s->fine_tail = std::max(-1, std::atoi(ps));
  if (const char* ps = std::getenv("SOL_WF_MIN_ITEMS")) s->wf_min_items = (uint32_t)std::max(0, std::atoi(ps));
  s->has_medium = d->n_mediums > 0;
  s->blocks_x = (d->width + SOL_TILE - 1) / SOL_TILE;
  s->blocks_y = (d->height + SOL_TILE - 1) / SOL_TILE;
  if ((rc = set_partition(s, 0, 1))) return rc;
  {  // a null table would be a GPU memory fault at the first launch, not an error code: refuse here
    const void* tables[] = {S.nodes, S.wides, S.leaf_refs, S.tris, S.tri_shade, S.quads, S.spheres, S.mediums, S.mats, S.texs, S.texels, S.lights};
    for (const void* p : tables)
      if (!p) return fail(SOL_EDEVICE, "internal error: a device table of the scene is missing");
  }
  s->tree_name = cands[0].name;
  if (calibrate) {
    // Probe every candidate tree with a counted render of 16 samples per pixel over ~256 pixel blocks spread across the image
    // and keep the one with the least search work (a wide-node visit weighs ~2.5 primitive tests, by instruction count).
    // Images do not depend on the tree, the counters are deterministic, so is the choice.
    auto free_cands = [&]() { for (auto& c : cands) c.dev.release(); };
    for (size_t k = 1; k < cands.size(); ++k)
      if ((rc = upload_tree(cands[k]))) { free_cands(); return rc; }
    const uint32_t nb = s->blocks_x * s->blocks_y;
    rc = set_partition(s, 0, (int)std::max(1u, nb / 256u));
    size_t pick = 0, current = 0;  // `current`: the candidate whose arrays the scene holds at the moment
    auto swap_in = [&](size_t k) {  // hand the scene's tree back to its candidate, adopt candidate k's
      if (k == current) return;
      DevTree& back = cands[current].dev;
      back.nodes = s->nodes; back.wides = s->wides; back.leaf_refs = s->leaf_refs; back.tris = s->tris; back.tri_shade = s->tri_shade;
      back.quads = s->quads; back.spheres = s->spheres; back.mediums = s->mediums; back.lights = s->lights;
      back.emin = S.wide_emin; back.depth = s->tree_depth; back.root = S.root; back.light0 = S.light0;
      back.old_tri = std::move(s->old_index[0]); back.old_sphere = std::move(s->old_index[1]); back.old_quad = std::move(s->old_index[2]);
      adopt_tree(cands[k].dev);
      current = k;
    };
    for (size_t k = 0; k < cands.size() && !rc; ++k) {
      swap_in(k);
      if (!(rc = sol_clear(s)) && !(rc = render_probe(s)))
        cands[k].cost = 2.5 * (double)s->stats.node_visits + (double)(s->stats.sphere_tests + s->stats.quad_tests + s->stats.triangle_tests);
      if (!rc && cands[k].cost < cands[pick].cost) pick = k;
    }
    if (std::getenv("SOL_VERBOSE")) {
      std::fprintf(stderr, "[solstrale] world tree probe:");
      for (auto& c : cands) std::fprintf(stderr, " %s %.4g (%zu nodes)", c.name.c_str(), c.cost, c.lay.nodes.size());
      std::fprintf(stderr, " -> %s\n", cands[pick].name.c_str());
    }
    if (hipStreamSynchronize(s->stream) != hipSuccess && !rc) rc = SOL_EDEVICE;
    swap_in(pick);
    s->tree_name = cands[pick].name;
    free_cands();
    if (rc) return rc;
    s->stats = SolStats{};
    if ((rc = set_partition(s, 0, 1)) || (rc = sol_clear(s))) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  // Cost probe: per 8x8 block, the ray count of the longest 4-sample item in a counted render of the whole frame, for the
  // heavy-first work order
  // (rebuild_order; sol_path.h decode_item_ordered). SOL_ORDER=0 switches it off.
  if (!opt.no_work_order_probe && !(std::getenv("SOL_ORDER") && std::atoi(std::getenv("SOL_ORDER")) == 0) && s->blocks_x * s->blocks_y >= 64u) {
    const uint32_t nb = s->blocks_x * s->blocks_y;
    uint32_t* cost_dev = nullptr;
    HIP_TRY(hipMalloc((void**)&cost_dev, (size_t)nb * sizeof(uint32_t)));
    hipError_t e = hipMemset(cost_dev, 0, (size_t)nb * sizeof(uint32_t));
    S.block_cost = cost_dev;
    rc = e == hipSuccess ? render_impl(s, 0, 4, 0xC057ull, true) : SOL_EDEVICE;
    S.block_cost = nullptr;
    s->block_cost.assign(nb, 0u);
    if (rc == SOL_OK && hipMemcpy(s->block_cost.data(), cost_dev, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = SOL_EDEVICE;
    hipFree(cost_dev);
    if (rc != SOL_OK) return rc == SOL_EDEVICE ? fail(SOL_EDEVICE, "cost probe failed") : rc;
    // The fine tail takes an item fetch per SAMPLE (a dependent load, three integer divisions: ~2 us): worth it where a sample
    // is long. MI355X, 1080p x 64 spp, ms with 0 / 1 / 2 whole items per lane in the tail: C3 (38 node visits per sample) 76.5 /
    // 75.5 / 74.4, C2 (22) 45.5 / 45.7 / 46.3, C1 (2) 10.5 / 11.2 / 11.8.
    if (s->stats.samples > 0) {
      const double vps = (double)s->stats.node_visits / (double)s->stats.samples;
      s->fine_tail_auto = vps >= 30.0 ? 8 : 0;
    }
    s->stats = SolStats{};
    if ((rc = sol_clear(s)) || (rc = rebuild_order(s))) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (std::getenv("SOL_VERBOSE")) std::fprintf(stderr, "[solstrale] work order: %u of %u blocks heavy (first)\n", S.n_first, s->n_local_blocks);
  }
  s->build_times[3] = seconds_since(t_probe0);
  cleanup.keep = true;
  *out = s;
  return SOL_OK;
}

int sol_scene_build_times(const SolScene* s, double out[4]) {
  if (!s || !out) return fail(SOL_EINVAL, "null argument");
  for (int k = 0; k < 4; ++k) out[k] = s->build_times[k];
  return SOL_OK;
}

int sol_scene_set_option(SolScene* s, int option, int64_t value) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  switch (option) {
    case SOL_OPT_SWITCH_BELOW:
      if (value < 0 || value > 64) return fail(SOL_EINVAL, "SOL_OPT_SWITCH_BELOW: 0..64");
      s->switch_below = (uint32_t)value;
      return SOL_OK;
    case SOL_OPT_MAX_BLOCKS_PER_CU:
      if (value < 0 || value > 64) return fail(SOL_EINVAL, "SOL_OPT_MAX_BLOCKS_PER_CU: 0..64");
      s->max_bpc = (int)value;
      return SOL_OK;
    case SOL_OPT_KERNEL:
      if (value < 0 || value > 3) return fail(SOL_EINVAL, "SOL_OPT_KERNEL: 0..3");
      s->kernel_version = (int)value;
      return SOL_OK;
    case SOL_OPT_FINE_TAIL:
      if (value < -1 || value > 64) return fail(SOL_EINVAL, "SOL_OPT_FINE_TAIL: -1 (by the probe), 0 (off) .. 64 quarters of an item per lane");
      s->fine_tail = (int)value;
      return SOL_OK;
    case SOL_OPT_WORK_ORDER:
      s->order_enabled = value != 0;
      HIP_TRY(hipSetDevice(s->device));
      return rebuild_order(s);
    default: return fail(SOL_EINVAL, "unknown option %d", option);
  }
}

int sol_scene_set_partition(SolScene* s, int rank, int world) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (world >= 1 && s->acc != s->acc_own) {
    const uint32_t nb = s->blocks_x * s->blocks_y;
    const size_t floats = (size_t)((nb + (uint32_t)world - 1u) / (uint32_t)world) * 64u * 3u;
    if (floats != s->acc_floats)
      return fail(SOL_EINVAL, "a caller-bound accumulator of %zu floats cannot follow the new partition (%zu floats): unbind it first", s->acc_floats, floats);
  }
  return set_partition(s, rank, world);
}

size_t sol_accum_floats(const SolScene* s) { return s ? s->acc_floats : 0; }
void* sol_accum_ptr(SolScene* s) { return s ? s->acc : nullptr; }

int sol_scene_bind_accum(SolScene* s, void* p, size_t n_floats) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  if (!p) { s->acc = s->acc_own; return SOL_OK; }
  if (n_floats < s->acc_floats) return fail(SOL_EINVAL, "bound accumulator too small: %zu < %zu floats", n_floats, s->acc_floats);
  s->acc = (float*)p;
  return SOL_OK;
}

int sol_scene_set_stream(SolScene* s, void* stream) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->stream = stream ? (hipStream_t)stream : s->own_stream;
  return SOL_OK;
}

int sol_clear(SolScene* s) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemsetAsync(s->acc, 0, s->acc_floats * sizeof(float), s->stream));
  return SOL_OK;
}

static int render_impl(SolScene* s, uint32_t first, uint32_t n, uint64_t seed, bool count) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  if (n == 0) return SOL_OK;
  if ((uint64_t)first + n > 0xFFFFFFFFull) return fail(SOL_EINVAL, "sample range overflows 32 bits");
  HIP_TRY(hipSetDevice(s->device));
  RenderParams P{};
  P.first_sample = first; P.n_samples = n;
  P.n_chunks = (n + SOL_CHUNK - 1) / SOL_CHUNK;
  P.rank = (uint32_t)s->rank; P.world = (uint32_t)s->world;
  P.n_local_blocks = s->n_local_blocks; P.blocks_x = s->blocks_x;
  P.seed_lo = (uint32_t)seed; P.seed_hi = (uint32_t)(seed >> 32);
  const uint64_t items = (uint64_t)P.n_chunks * P.n_local_blocks * 64u;
  // the 32-bit work counter keeps counting after the items run out (every wave adds 64 per refused fetch until all its
  // lanes have left): 16 M of headroom is > 100 times what 5120 resident waves can add
  if (items > SOL_MAX_ITEMS) return fail(SOL_EINVAL, "too many work items in one call (%llu): split the sample range", (unsigned long long)items);
  P.n_items = (uint32_t)items;
  if (P.n_items == 0) return SOL_OK;
  P.switch_below = s->switch_below;
  // kernel choice: 0 = auto (two-kernel wavefront for large jobs, one-path-per-lane kernel for small ones)
  int version = s->kernel_version;
  // Measured on MI355X (C3, 128 spp): v1 997, v2 905, v3 684 Msamples/s - the wavefront variants raise the traversal's lane
  // occupancy (0.45 -> 0.67-0.73) but pay for it in state traffic, refill stalls and per-round tails, so v1 is the default.
  if (version == 0) version = 1;
  int bpc = version == 3 ? sol_wf_trace_blocks_per_cu(count, s->has_medium) : sol_render_blocks_per_cu(version, count, s->has_medium);
  if (s->max_bpc > 0) bpc = std::max(1, std::min(bpc, s->max_bpc));  // SOL_OPT_MAX_BLOCKS_PER_CU
  uint32_t grid = (uint32_t)(s->n_cu * bpc);
  const uint32_t need_blocks = (P.n_items + SOL_WG - 1) / SOL_WG;
  if (grid > need_blocks) grid = need_blocks;
  P.total_threads = grid * SOL_WG;
  const uint32_t lds_depth = version == 3 ? (uint32_t)sol_wf_lds_stack_depth() : (uint32_t)SOL_LDS_STACK;
  if (version == 3) {
    // one global pool: enough slots that the trace kernel has >= 16 rays per resident lane, never more than the items
    uint64_t want = std::min<uint64_t>(P.n_items, s->wf_slots);
    want = ((want + SOL_WG - 1) / SOL_WG) * SOL_WG;
    P.pool_slots = (uint32_t)want;
    const size_t need = sol_wf_pool_bytes(P.pool_slots);  // POOL_RECORDS float4 per slot
    if (need > s->pool_bytes) {
      HIP_TRY(hipStreamSynchronize(s->stream));
      if (s->pool) hipFree(s->pool);
      s->pool = nullptr;
      s->pool_bytes = 0;
      HIP_TRY(hipMalloc(&s->pool, need));
      s->pool_bytes = need;
    }
    if ((size_t)P.pool_slots > s->queue_slots) {  // item reservoirs: one uint2 per 64 slots (shade wave)
      HIP_TRY(hipStreamSynchronize(s->stream));
      if (s->queue) hipFree(s->queue);
      s->queue = nullptr;
      s->queue_slots = 0;
      HIP_TRY(hipMalloc((void**)&s->queue, (size_t)P.pool_slots / 64 * 8));
      s->queue_slots = P.pool_slots;
    }
    if (!s->wf_ctr) {
      HIP_TRY(hipMalloc(&s->wf_ctr, 64));
      HIP_TRY(hipHostMalloc((void**)&s->wf_ctr_host, 64, hipHostMallocDefault));
    }
  } else if (version == 2) {
    // pool of path slots: per wave a multiple of 64, enough that every wave has several rays per lane in flight
    const uint32_t waves = grid * (SOL_WG / 64);
    uint32_t per_wave = (P.n_items + waves - 1) / waves;
    per_wave = ((per_wave + 63u) / 64u) * 64u;
    P.pool_slots = std::min<uint32_t>(SOL_POOL_MAX, std::max<uint32_t>(64u, per_wave));
    if (s->pool_slots_override) P.pool_slots = std::min<uint32_t>(SOL_POOL_MAX, ((s->pool_slots_override + 63u) / 64u) * 64u);
    const size_t need = sol_pool_bytes_per_wave(P.pool_slots) * waves;
    if (need > s->pool_bytes) {
      HIP_TRY(hipStreamSynchronize(s->stream));
      if (s->pool) hipFree(s->pool);
      s->pool = nullptr;
      s->pool_bytes = 0;
      HIP_TRY(hipMalloc(&s->pool, need));
      s->pool_bytes = need;
    }
  }
  // spill stack only when the tree can out-grow the LDS stack
  size_t spill_words = s->tree_depth > lds_depth ? (size_t)P.total_threads * (s->tree_depth - lds_depth) : 16;
  if (spill_words > s->spill_words) {
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->spill) hipFree(s->spill);
    s->spill = nullptr;
    HIP_TRY(hipMalloc((void**)&s->spill, spill_words * sizeof(uint32_t)));
    s->spill_words = spill_words;
  }
  const size_t slots3 = (size_t)P.n_local_blocks * 64u * 3u;
  // Fine tail (v1): the last pairs (block, chunk) of the work order - all of them in the last chunk, blocks the probe found light -
  // are handed out one sample at a time: about one whole item per resident lane (SOL_FINE_TAIL quarters), so that single
  // samples are still on offer while the slowest lanes finish their last whole item. Counted launches keep whole items
  // (per-item ray counts).
  P.n_coarse = P.n_items;
  P.fine_count = n - (P.n_chunks - 1u) * SOL_CHUNK;
  P.stage_at = P.n_items;  // (= n_chunks * slots: right behind the chunk sums)
  size_t stage_floats = 0;
  const int fine_tail = s->fine_tail >= 0 ? s->fine_tail : s->fine_tail_auto;
  if (version == 1 && !count && fine_tail > 0) {
    const uint32_t rest = P.n_local_blocks - std::min(P.n_local_blocks, s->S.n_first);
    const uint32_t pairs = std::min<uint32_t>(rest, (uint32_t)(((uint64_t)fine_tail * (P.total_threads / 64u) + 3u) / 4u));
    const uint64_t total = items - (uint64_t)pairs * 64u + (uint64_t)pairs * 64u * SOL_CHUNK;
    if (pairs > 0 && total <= SOL_MAX_ITEMS && items + (uint64_t)pairs * 64u * SOL_CHUNK <= 0xFFFFFFFFull) {
      stage_floats = (size_t)pairs * 64u * SOL_CHUNK * 3u;
      P.n_coarse = (uint32_t)(items - (uint64_t)pairs * 64u);
      P.n_items = (uint32_t)total;
    }
  }
  // v1 writes every chunk sum into `partial` (also when the call has a single chunk); v2 / v3 add a single chunk straight
  // into the accumulator
  const bool via_partial = version == 1 || P.n_chunks > 1;
  if (via_partial) {
    size_t need = slots3 * P.n_chunks + stage_floats;
    if (need > s->partial_floats) {
      HIP_TRY(hipStreamSynchronize(s->stream));
      if (s->partial) hipFree(s->partial);
      s->partial = nullptr;
      s->partial_floats = 0;
      HIP_TRY(hipMalloc((void**)&s->partial, need * sizeof(float)));
      s->partial_floats = need;
    }
    // padding pixels of edge blocks are never written: keep them zero
    if ((s->S.width % SOL_TILE) || (s->S.height % SOL_TILE)) HIP_TRY(hipMemsetAsync(s->partial, 0, slots3 * P.n_chunks * sizeof(float), s->stream));
  }
  HIP_TRY(hipMemsetAsync(s->work, 0, sizeof(uint32_t), s->stream));
  if (count) HIP_TRY(hipMemsetAsync(s->counters, 0, sizeof(DevCounters), s->stream));
  if (s->timing) HIP_TRY(hipEventRecord(s->ev_start, s->stream));
  if (version == 3) {
    // rounds of (shade, trace) until no slot holds work; the live-slot count is read back every few rounds
    uint32_t* ctr = (uint32_t*)s->wf_ctr;  // WfCounters {work_next, slot_cursor, live, pad}
    HIP_TRY(hipMemsetAsync(ctr, 0, 16, s->stream));
    HIP_TRY(hipMemsetAsync((char*)s->pool + (size_t)P.pool_slots * 16, 0, (size_t)P.pool_slots * 16, s->stream));  // record 1: flags
    HIP_TRY(hipMemsetAsync(s->queue, 0, (size_t)P.pool_slots / 64 * 8, s->stream));                              // reservoirs
    const uint32_t check_every = 16;
    uint32_t rounds = 0;
    for (;;) {
      HIP_TRY(hipMemsetAsync(ctr + 1, 0, 8, s->stream));  // slot_cursor, live
      HIP_TRY(sol_launch_wf_shade(s->S, P, s->acc, s->partial, ctr, s->pool, s->queue, s->counters, count, s->stream));
      HIP_TRY(sol_launch_wf_trace(s->S, P, ctr, s->pool, s->spill, s->counters, grid, count, s->has_medium, s->stream));
      ++rounds;
      if (rounds % check_every == 0) {
        HIP_TRY(hipMemcpyAsync(s->wf_ctr_host, ctr, 16, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (s->wf_ctr_host[2] == 0) break;
        if (rounds > 4000000u) return fail(SOL_EDEVICE, "wavefront did not drain");
      }
    }
    s->last_rounds = rounds;
  } else {
    if (!s->dscene) HIP_TRY(hipMalloc((void**)&s->dscene, sizeof(DevScene)));
    if (!s->dscene_valid || std::memcmp(&s->S, &s->S_uploaded, sizeof(DevScene)) != 0) {
      // rare (scene creation, tree probe, auxiliary renders): launches already queued may still read the old copy
      HIP_TRY(hipStreamSynchronize(s->stream));
      HIP_TRY(hipMemcpy(s->dscene, &s->S, sizeof(DevScene), hipMemcpyHostToDevice));
      std::memcpy(&s->S_uploaded, &s->S, sizeof(DevScene));
      s->dscene_valid = true;
    }
    HIP_TRY(sol_launch_render(version, s->S, s->dscene, P, s->acc, s->partial, s->work, s->spill, s->pool, s->counters, grid, count,
                              s->has_medium, s->tree_depth > (uint32_t)SOL_LDS_STACK, s->stream));
  }
  if (s->timing) { HIP_TRY(hipEventRecord(s->ev_stop, s->stream)); s->timed_launches++; }
  s->last_grid = grid;
  s->last_version = version;
  if (P.n_coarse != P.n_items) HIP_TRY(sol_launch_stage_resolve(s->dscene, P, s->partial, s->stream));
  if (via_partial) HIP_TRY(sol_launch_resolve(s->acc, s->partial, (uint32_t)slots3, P.n_chunks, s->stream));
  if (count) {
    DevCounters c;
    HIP_TRY(hipMemcpyAsync(&c, s->counters, sizeof c, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->stats.samples = c.samples; s->stats.rays = c.rays; s->stats.node_visits = c.node_visits;
    s->stats.sphere_tests = c.sphere_tests; s->stats.quad_tests = c.quad_tests; s->stats.triangle_tests = c.triangle_tests;
    s->stats.shades = c.shades; s->stats.texel_fetches = c.texel_fetches; s->stats.max_stack = c.max_stack;
    for (int k = 0; k < 6; ++k) s->stats.phase[k] = c.phase[k];
  }
  return SOL_OK;
}

static int render_probe(SolScene* s) { return render_impl(s, 0, SOL_CHUNK, 0x50B3ull, true); }
int sol_render(SolScene* s, uint32_t first, uint32_t n, uint64_t seed) { return render_impl(s, first, n, seed, false); }
int sol_render_counted(SolScene* s, uint32_t first, uint32_t n, uint64_t seed) { return render_impl(s, first, n, seed, true); }

int sol_sync(SolScene* s) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return SOL_OK;
}

int sol_read(SolScene* s, float* rgb_sum) {
  if (!s || !rgb_sum) return fail(SOL_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(sol_launch_unpermute(s->acc, s->image, s->S.width, s->S.height, s->blocks_x, (uint32_t)s->world, (uint32_t)s->rank,
                               s->acc_floats, s->stream));
  HIP_TRY(hipMemcpyAsync(rgb_sum, s->image, (size_t)s->S.width * s->S.height * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return SOL_OK;
}

// Auxiliary albedo / normal buffers (src/renderer/mod.rs:175-204): at depth 0 the reference evaluates AlbedoShader and
// NormalShader on the hit of the primary ray (background / zero on a miss) and accumulates them beside the pixel colour.
// Those are exactly the single-hit shaders of this library evaluated on the same primary ray - same (seed, pixel, sample)
// key, hence the same jitter and camera ray - so the two planes are two primary-ray-only renders into their own
// accumulators; no path-tracing kernel variant is needed. (For Blend materials the reference's extra scatter call draws its
// branch independently of the path's, as the separate render does.)
int sol_render_aux(SolScene* s, uint32_t first, uint32_t n, uint64_t seed) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  if (s->aux_floats != s->acc_floats || !s->aux[0]) {
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (int k = 0; k < 2; ++k) {
      if (s->aux[k]) hipFree(s->aux[k]);
      s->aux[k] = nullptr;
      HIP_TRY(hipMalloc((void**)&s->aux[k], std::max<size_t>(s->acc_floats * sizeof(float), 64)));
      HIP_TRY(hipMemsetAsync(s->aux[k], 0, std::max<size_t>(s->acc_floats * sizeof(float), 64), s->stream));
    }
    s->aux_floats = s->acc_floats;
  }
  float* const acc = s->acc;
  const uint32_t shader = s->S.shader;
  const uint32_t kinds[2] = {SOL_SHADER_ALBEDO, SOL_SHADER_NORMAL};
  int rc = SOL_OK;
  const float bg[3] = {s->S.bgx, s->S.bgy, s->S.bgz};
  const float* const env = s->S.env;
  for (int k = 0; k < 2 && rc == SOL_OK; ++k) {
    s->acc = s->aux[k];
    s->S.shader = kinds[k];
    if (k == 1) { s->S.bgx = s->S.bgy = s->S.bgz = 0.0f; s->S.env = nullptr; }  // a miss: albedo = background colour, normal = ZERO_VECTOR (mod.rs:197-204)
    rc = render_impl(s, first, n, seed, false);
  }
  s->acc = acc;
  s->S.shader = shader;
  s->S.bgx = bg[0]; s->S.bgy = bg[1]; s->S.bgz = bg[2];
  s->S.env = env;
  return rc;
}

int sol_clear_aux(SolScene* s) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  for (int k = 0; k < 2; ++k)
    if (s->aux[k] && s->aux_floats == s->acc_floats) HIP_TRY(hipMemsetAsync(s->aux[k], 0, s->aux_floats * sizeof(float), s->stream));
  return SOL_OK;
}

int sol_read_aux(SolScene* s, float* albedo_sum, float* normal_sum) {
  if (!s || (!albedo_sum && !normal_sum)) return fail(SOL_EINVAL, "null argument");
  if (!s->aux[0] || s->aux_floats != s->acc_floats) return fail(SOL_EINVAL, "no auxiliary buffers: call sol_render_aux first");
  HIP_TRY(hipSetDevice(s->device));
  float* outs[2] = {albedo_sum, normal_sum};
  for (int k = 0; k < 2; ++k) {
    if (!outs[k]) continue;
    HIP_TRY(sol_launch_unpermute(s->aux[k], s->image, s->S.width, s->S.height, s->blocks_x, (uint32_t)s->world, (uint32_t)s->rank,
                                 s->acc_floats, s->stream));
    HIP_TRY(hipMemcpyAsync(outs[k], s->image, (size_t)s->S.width * s->S.height * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  return SOL_OK;
}

int sol_unpermute(SolScene* s, const void* gathered, int world, void* image) {
  if (!s || !gathered || !image) return fail(SOL_EINVAL, "null argument");
  if (world != s->world) return fail(SOL_EINVAL, "world %d differs from the scene's partition (%d)", world, s->world);
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(sol_launch_unpermute((const float*)gathered, (float*)image, s->S.width, s->S.height, s->blocks_x, (uint32_t)world,
                               0xFFFFFFFFu, s->acc_floats, s->stream));
  return SOL_OK;
}

int sol_tonemap_rgb8(SolScene* s, const void* image, uint32_t spp, uint8_t* out) {
  if (!s || !image || !out || spp == 0) return fail(SOL_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(s->device));
  const uint32_t n = s->S.width * s->S.height * 3;
  HIP_TRY(sol_launch_tonemap((const float*)image, s->rgb8, n, spp, s->stream));
  HIP_TRY(hipMemcpyAsync(out, s->rgb8, n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return SOL_OK;
}

int sol_resolve_image(SolScene* s, void** image_dev) {
  if (!s || !image_dev) return fail(SOL_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(sol_launch_unpermute(s->acc, s->image, s->S.width, s->S.height, s->blocks_x, (uint32_t)s->world, (uint32_t)s->rank,
                               s->acc_floats, s->stream));
  *image_dev = s->image;
  return SOL_OK;
}

// create_gaussian_blur_weights (src/util/gaussian.rs:3-25)
static std::vector<double> gaussian_blur_weights(size_t kernel_size, double std_dev) {
  std::vector<double> w(kernel_size);
  const double mean = (double)(kernel_size - 1) / 2.0;
  double sum = 0.0;
  for (size_t i = 0; i < kernel_size; ++i) {
    const double a = ((double)i - mean) / std_dev;
    w[i] = std::exp(-0.5 * a * a);
  }
  for (size_t i = 0; i < kernel_size; ++i) sum += w[i];  // iter().sum(): left to right from 0.0
  for (size_t i = 0; i < kernel_size; ++i) w[i] /= sum;
  return w;
}

int sol_gaussian_blur_weights(uint32_t kernel_size, double std_dev, double* out) {
  if (!out || kernel_size == 0) return fail(SOL_EINVAL, "bad argument");
  std::vector<double> w = gaussian_blur_weights(kernel_size, std_dev);
  std::memcpy(out, w.data(), w.size() * sizeof(double));
  return SOL_OK;
}

static int bloom_impl(SolScene* s, void* image, uint32_t spp, double ksf, double threshold, double max_intensity, uint8_t* out) {
  if (!s || !image || spp == 0) return fail(SOL_EINVAL, "bad argument");
  if (!(ksf >= 0.0 && ksf <= 0.5)) return fail(SOL_EINVAL, "kernel_size_fraction must be between 0 and 0.5");  // bloom.rs:33-37
  HIP_TRY(hipSetDevice(s->device));
  const uint32_t W = s->S.width, H = s->S.height;
  const size_t n = (size_t)W * H * 3;
  // bloom.rs:86-91
  const double thr = threshold * (double)spp, maxi = max_intensity * (double)spp;
  const size_t k = (size_t)(ksf * (double)W) * 2 + 1;
  std::vector<double> w = gaussian_blur_weights(k, (double)k / 5.0);
  if (!s->bloom_a) HIP_TRY(hipMalloc((void**)&s->bloom_a, n * sizeof(double)));
  if (!s->bloom_b) HIP_TRY(hipMalloc((void**)&s->bloom_b, n * sizeof(double)));
  if (k > s->bloom_w_cap) {
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->bloom_w) hipFree(s->bloom_w);
    s->bloom_w = nullptr; s->bloom_w_cap = 0;
    HIP_TRY(hipMalloc((void**)&s->bloom_w, k * sizeof(double)));
    s->bloom_w_cap = k;
  }
  HIP_TRY(hipMemcpyAsync(s->bloom_w, w.data(), k * sizeof(double), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));  // `w` is pageable host memory about to go out of scope
  HIP_TRY(sol_launch_bloom((float*)image, s->bloom_a, s->bloom_b, s->bloom_w, (uint32_t)k, W, H, thr, maxi, out ? s->rgb8 : nullptr, spp,
                           s->stream));
  if (out) {
    HIP_TRY(hipMemcpyAsync(out, s->rgb8, n, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  return SOL_OK;
}
int sol_bloom(SolScene* s, void* image, uint32_t spp, double ksf, double threshold, double max_intensity) {
  return bloom_impl(s, image, spp, ksf, threshold, max_intensity, nullptr);
}
int sol_bloom_rgb8(SolScene* s, const void* image, uint32_t spp, double ksf, double threshold, double max_intensity, uint8_t* out) {
  if (!out) return fail(SOL_EINVAL, "bad argument");
  return bloom_impl(s, const_cast<void*>(image), spp, ksf, threshold, max_intensity, out);
}

int sol_eval(int device, uint32_t fn, const float* in, uint32_t n, uint32_t in_stride, float* out, uint32_t out_stride) {
  if (!in || !out || !in_stride || !out_stride) return fail(SOL_EINVAL, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SOL_EDEVICE, "no HIP device available");
  HIP_TRY(hipSetDevice(device));
  float *din = nullptr, *dout = nullptr;
  const size_t ib = (size_t)n * in_stride * sizeof(float), ob = (size_t)n * out_stride * sizeof(float);
  HIP_TRY(hipMalloc((void**)&din, std::max<size_t>(ib, 64)));
  if (hipMalloc((void**)&dout, std::max<size_t>(ob, 64)) != hipSuccess) { hipFree(din); return fail(SOL_ENOMEM, "hipMalloc failed"); }
  hipError_t e = hipMemcpy(din, in, ib, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(dout, 0, std::max<size_t>(ob, 64));
  if (e == hipSuccess) e = sol_launch_eval(fn, din, n, in_stride, dout, out_stride, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, dout, ob, hipMemcpyDeviceToHost);
  hipFree(din);
  hipFree(dout);
  if (e != hipSuccess) return fail(SOL_EDEVICE, "sol_eval: %s", hipGetErrorString(e));
  return SOL_OK;
}

int sol_debug_path(SolScene* s, uint32_t x, uint32_t y, uint32_t sample, uint64_t seed, float* rows, uint32_t max_rows) {
  if (!s || !rows || max_rows < 2 || x >= s->S.width || y >= s->S.height) return fail(SOL_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(s->device));
  RenderParams P{};
  P.seed_lo = (uint32_t)seed; P.seed_hi = (uint32_t)(seed >> 32);
  P.total_threads = 1;
  float* dout = nullptr;
  uint32_t* dspill = nullptr;
  const size_t ob = (size_t)max_rows * 12 * sizeof(float);
  HIP_TRY(hipMalloc((void**)&dout, ob));
  hipError_t e = hipMalloc((void**)&dspill, (size_t)(SOL_SPILL_STACK + 8) * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(dout, 0, ob);
  if (e == hipSuccess) e = sol_launch_debug_path(s->S, P, x, y, sample, dspill, dout, max_rows, s->has_medium, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(rows, dout, ob, hipMemcpyDeviceToHost);
  hipFree(dout);
  if (dspill) hipFree(dspill);
  if (e != hipSuccess) return fail(SOL_EDEVICE, "sol_debug_path: %s", hipGetErrorString(e));
  for (uint32_t r = 0; r < max_rows && rows[r * 12 + 3] != -1.0f; ++r) {  // hit references: device order -> the caller's indices
    uint32_t ref;
    std::memcpy(&ref, &rows[r * 12 + 7], 4);
    const int a = WideLayout::arr(SOL_REF_KIND(ref));
    if (a >= 0 && SOL_REF_INDEX(ref) < s->old_index[a].size()) {
      ref = SOL_MAKE_REF(SOL_REF_KIND(ref), s->old_index[a][SOL_REF_INDEX(ref)]);
      std::memcpy(&rows[r * 12 + 7], &ref, 4);
    }
  }
  return SOL_OK;
}

int sol_kernel_timing(SolScene* s, int enable) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  if (enable && !s->ev_start) {
    HIP_TRY(hipEventCreate(&s->ev_start));
    HIP_TRY(hipEventCreate(&s->ev_stop));
  }
  s->timing = enable != 0;
  s->timed_launches = 0;
  return SOL_OK;
}

int sol_last_kernel_ms(SolScene* s, float* ms, uint32_t* grid_blocks) {
  if (!s || !ms) return fail(SOL_EINVAL, "null argument");
  if (!s->timing || s->timed_launches == 0) return fail(SOL_EINVAL, "no timed render launch (call sol_kernel_timing first)");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipEventSynchronize(s->ev_stop));
  HIP_TRY(hipEventElapsedTime(ms, s->ev_start, s->ev_stop));
  if (grid_blocks) *grid_blocks = s->last_grid;
  return SOL_OK;
}

int sol_stats(const SolScene* s, SolStats* out) {
  if (!s || !out) return fail(SOL_EINVAL, "null argument");
  *out = s->stats;
  return SOL_OK;
}

}  // extern "C"

// ---- multi-GPU behind the ABI: RCCL communicator + gather of the tile accumulators to rank 0 -------------------------------
// RCCL is loaded on first use (dlopen), so that a single-GPU process has no dependency on it; when the host process has
// already loaded an RCCL (e.g. torch's), the loader hands back that one (same soname).
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) { r.error = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "?"); return; }
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + n; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  return r;
}
#define RCCL_TRY(expr)                                                                                   \
  do {                                                                                                   \
    ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess) return fail(SOL_EDEVICE, "%s: %s", #expr, rccl().GetErrorString(r_));         \
  } while (0)
}  // namespace

extern "C" {

static_assert(sizeof(ncclUniqueId) == SOL_UNIQUE_ID_BYTES, "ncclUniqueId size");

int sol_comm_unique_id(uint8_t id[SOL_UNIQUE_ID_BYTES]) {
  if (!id) return fail(SOL_EINVAL, "null argument");
  Rccl& R = rccl();
  if (!R.error.empty()) return fail(SOL_EDEVICE, "%s", R.error.c_str());
  ncclUniqueId u;
  RCCL_TRY(R.GetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return SOL_OK;
}

int sol_comm_destroy(SolScene* s) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  if (s->comm) {
    hipSetDevice(s->device);
    hipStreamSynchronize(s->stream);
    rccl().CommDestroy((ncclComm_t)s->comm);
    s->comm = nullptr;
  }
  if (s->gathered) { hipFree(s->gathered); s->gathered = nullptr; s->gathered_floats = 0; }
  return SOL_OK;
}

int sol_comm_init(SolScene* s, int rank, int world, const uint8_t id[SOL_UNIQUE_ID_BYTES]) {
  if (!s || !id) return fail(SOL_EINVAL, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(SOL_EINVAL, "bad rank %d of %d", rank, world);
  Rccl& R = rccl();
  if (!R.error.empty()) return fail(SOL_EDEVICE, "%s", R.error.c_str());
  int rc = sol_comm_destroy(s);
  if (rc || (rc = sol_scene_set_partition(s, rank, world))) return rc;
  HIP_TRY(hipSetDevice(s->device));
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclComm_t comm = nullptr;
  RCCL_TRY(R.CommInitRank(&comm, world, u, rank));
  s->comm = comm;
  return SOL_OK;
}

// One collective per emitted image: every rank's compact accumulator (equal sizes, sol_accum_floats) goes to
// rank 0 in ONE group of point-to-point transfers - each shard rides its own xGMI link into the root - and rank 0 un-permutes
// the `world` compact buffers into the row-major image. Without a communicator (world 1) it is sol_resolve_image into image_dev.
int sol_gather(SolScene* s, void* image_dev) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  if (s->world > 1 && !s->comm) return fail(SOL_EINVAL, "the scene is partitioned %d-way but has no communicator: call sol_comm_init", s->world);
  float* image = image_dev ? (float*)image_dev : s->image;
  if (s->world == 1) {
    HIP_TRY(sol_launch_unpermute(s->acc, image, s->S.width, s->S.height, s->blocks_x, 1u, 0u, s->acc_floats, s->stream));
    return SOL_OK;
  }
  Rccl& R = rccl();
  ncclComm_t comm = (ncclComm_t)s->comm;
  const size_t n = s->acc_floats;
  if (s->rank == 0) {
    if (s->gathered_floats != n * (size_t)s->world) {
      HIP_TRY(hipStreamSynchronize(s->stream));
      if (s->gathered) hipFree(s->gathered);
      s->gathered = nullptr; s->gathered_floats = 0;
      HIP_TRY(hipMalloc((void**)&s->gathered, n * (size_t)s->world * sizeof(float)));
      s->gathered_floats = n * (size_t)s->world;
    }
    HIP_TRY(hipMemcpyAsync(s->gathered, s->acc, n * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    RCCL_TRY(R.GroupStart());
    for (int r = 1; r < s->world; ++r) {
      ncclResult_t e = R.Recv(s->gathered + (size_t)r * n, n, ncclFloat, r, comm, s->stream);
      if (e != ncclSuccess) { R.GroupEnd(); return fail(SOL_EDEVICE, "ncclRecv from rank %d: %s", r, R.GetErrorString(e)); }
    }
    RCCL_TRY(R.GroupEnd());
    HIP_TRY(sol_launch_unpermute(s->gathered, image, s->S.width, s->S.height, s->blocks_x, (uint32_t)s->world, 0xFFFFFFFFu, n, s->stream));
  } else {
    RCCL_TRY(R.Send(s->acc, n, ncclFloat, 0, comm, s->stream));
  }
  return SOL_OK;
}

// Diagnostic for boxes with ONE GPU (where no second rank can exist): moves this rank's accumulator to itself through the
// communicator - grouped ncclSend + ncclRecv with peer = own rank, the same calls sol_gather issues - and compares the bytes.
int sol_comm_self_check(SolScene* s) {
  if (!s) return fail(SOL_EINVAL, "null scene");
  if (!s->comm) return fail(SOL_EINVAL, "no communicator: call sol_comm_init");
  HIP_TRY(hipSetDevice(s->device));
  Rccl& R = rccl();
  const size_t n = s->acc_floats;
  float* tmp = nullptr;
  HIP_TRY(hipMalloc((void**)&tmp, n * sizeof(float)));
  hipError_t e = hipMemsetAsync(tmp, 0xFF, n * sizeof(float), s->stream);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) {
    r = R.GroupStart();
    if (r == ncclSuccess) r = R.Send(s->acc, n, ncclFloat, s->rank, (ncclComm_t)s->comm, s->stream);
    if (r == ncclSuccess) r = R.Recv(tmp, n, ncclFloat, s->rank, (ncclComm_t)s->comm, s->stream);
    ncclResult_t r2 = R.GroupEnd();
    if (r == ncclSuccess) r = r2;
  }
  std::vector<float> a(n), b(n);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(a.data(), s->acc, n * sizeof(float), hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(b.data(), tmp, n * sizeof(float), hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  hipFree(tmp);
  if (r != ncclSuccess) return fail(SOL_EDEVICE, "RCCL self transfer: %s", R.GetErrorString(r));
  if (e != hipSuccess) return fail(SOL_EDEVICE, "self check: %s", hipGetErrorString(e));
  if (std::memcmp(a.data(), b.data(), n * sizeof(float)) != 0) return fail(SOL_EDEVICE, "RCCL self transfer returned different bytes");
  return SOL_OK;
}

int sol_read_image(SolScene* s, float* rgb_sum) {
  if (!s || !rgb_sum) return fail(SOL_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(rgb_sum, s->image, (size_t)s->S.width * s->S.height * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return SOL_OK;
}

uint32_t sol_max_samples_per_call(const SolScene* s) {
  if (!s || s->n_local_blocks == 0) return 0xFFFFFFFFu;
  const uint64_t chunks = SOL_MAX_ITEMS / ((uint64_t)s->n_local_blocks * 64u);
  return (uint32_t)std::min<uint64_t>(chunks * SOL_CHUNK, 0xFFFFFFF0ull);
}

}  // extern "C"
