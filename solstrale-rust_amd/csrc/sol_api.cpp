// sol_api.cpp -- the handle of the C ABI (include/solstrale_hip.h): error reporting, developer overrides, life cycle, options,
// tile partition and work order, accumulators, read-back, statistics. (sol_scene.h lists the other translation units.)
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
#include <vector>

#include "sol_scene.h"

namespace {
thread_local std::string g_err;
}

int sol_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

SolDevOverrides sol_dev_overrides() {
  SolDevOverrides o;
  auto num = [](const char* name, int unset) { const char* v = std::getenv(name); return v ? std::atoi(v) : unset; };
  if (const char* kv = std::getenv("SOL_KERNEL")) {
    if (kv[0] == 'v') kv++;
    o.kernel_version = (kv[0] >= '1' && kv[0] <= '4') ? kv[0] - '0' : 0;
  }
  if (const char* v = std::getenv("SOL_BVH")) o.bvh = std::strcmp(v, "sah") == 0 ? "sah16" : v;
  if (const char* v = std::getenv("SOL_COLLAPSE")) o.greedy_collapse = std::strcmp(v, "greedy") == 0;
  if (const char* v = std::getenv("SOL_SLOTS")) o.octant_slots = std::strcmp(v, "octant") == 0;
  if (const char* v = std::getenv("SOL_NODE_COST")) o.node_cost = std::atof(v);
  if (const char* bl = std::getenv("SOL_SAH_LIST"))  // experiment: other candidate sets, e.g. SOL_SAH_LIST=4,12,32
    for (const char* p = bl; *p;) { o.sah_bins.push_back(std::max(2, std::min(64, std::atoi(p)))); while (*p && *p != ',') ++p; if (*p) ++p; }
  o.ploc_radius = std::max(0, num("SOL_PLOC_R", 0));
  o.split_percent = num("SOL_SPLIT", -1);
  o.split_slack = num("SOL_SPLIT_SLACK", -1);
  o.split_keep = num("SOL_SPLIT_KEEP", -1);
  o.background_blocks = num("SOL_BACKGROUND_BLOCKS", -1);
  o.reinsert_rounds = num("SOL_REINSERT", -1);
  o.reinsert_stride = std::max(0, num("SOL_REINSERT_STRIDE", 0));
  o.order_mode = num("SOL_ORDER", 2);
  o.switch_below = std::min(64, num("SOL_SWITCH", -1));
  o.max_bpc = num("SOL_MAX_BPC", -1);
  o.fine_tail = std::max(-2, num("SOL_FINE_TAIL", -2));
  o.pool_swap_min = std::max(0, num("SOL_POOL_SWAP", 0));
  o.probe_radii = num("SOL_PROBE_RADII", -1);
  o.pool_slots = std::max(0, num("SOL_POOL_SLOTS", 0));
  o.wf_slots = std::max(0, num("SOL_WF_SLOTS", 0));
  o.wf_min_items = num("SOL_WF_MIN_ITEMS", -1);
  if (const char* v = std::getenv("SOL_RCCL_LIB")) o.rccl_lib = v;
  o.verbose = std::getenv("SOL_VERBOSE") != nullptr;
  return o;
}

// DevScene::block_order for the current partition from the probe's per-block costs: the blocks more than three times as
// costly as the average come first, costliest first; everything else keeps its order. No heavy blocks: identity (null).
int sol_rebuild_order(SolScene* s) {
  s->S.block_order = nullptr;
  s->S.n_first = 0;
  s->n_background_local = 0;
  const uint32_t n = s->n_local_blocks;
  auto global_of = [&](uint32_t lb) -> size_t { return s->local_blocks.empty() ? (size_t)lb * s->world + s->rank : s->local_blocks[lb]; };
  // background blocks (SolSceneInfo::background_blocks) go LAST, whatever else the order does: a launch that does not trace them
  // stops in front of them (RenderParams::n_traced_blocks)
  std::vector<uint32_t> background;
  std::vector<uint8_t> is_background(n, 0);
  for (uint32_t lb = 0; lb < n && !s->background_block.empty(); ++lb) {
    const size_t b = global_of(lb);
    if (b < s->background_block.size() && s->background_block[b]) { is_background[lb] = 1; background.push_back(lb); }
  }
  const bool by_cost = s->order_enabled && !s->block_cost.empty() && n >= 2;
  if (!by_cost && background.empty()) return SOL_OK;
  std::vector<uint32_t> cost(n, 0u);
  double sum = 0.;
  for (uint32_t lb = 0; lb < n && by_cost; ++lb) {
    const size_t b = global_of(lb);
    cost[lb] = b < s->block_cost.size() ? s->block_cost[b] : 0u;
    sum += cost[lb];
  }
  const double limit = 3.0 * sum / n;
  std::vector<uint32_t> heavy, rest;
  for (uint32_t lb = 0; lb < n; ++lb)
    if (!is_background[lb]) (by_cost && cost[lb] > limit ? heavy : rest).push_back(lb);
  // The rest keeps the chunk-major order, but within a chunk the blocks go from costly to cheap in eight cost classes (octiles of
  // the probe's ray counts; raster order inside a class, so neighbours stay together): the launch then ENDS on the cheapest blocks
  // (sky) of the last chunk instead of on whatever the raster order ends on (floor and walls: one 16-sample item of a long path
  // is milliseconds of one lane). A fixed ~6 ms of tail per launch otherwise - 1 % of a 1080p x 512 spp frame on one GPU, 7 % of
  // its eighth on eight.
  const int order_mode = s->order_mode;  // (SOL_ORDER) 1: heavy-first only (round 1)
  if (by_cost && order_mode >= 2 && rest.size() >= 64) {
    std::vector<uint32_t> sorted_cost;
    sorted_cost.reserve(rest.size());
    for (uint32_t lb : rest) sorted_cost.push_back(cost[lb]);
    std::sort(sorted_cost.begin(), sorted_cost.end());
    uint32_t edge[7];
    for (int k = 0; k < 7; ++k) edge[k] = sorted_cost[(size_t)(k + 1) * sorted_cost.size() / 8];
    auto cls = [&](uint32_t lb) { int c = 0; while (c < 7 && cost[lb] >= edge[c]) ++c; return c; };
    std::stable_sort(rest.begin(), rest.end(), [&](uint32_t a, uint32_t b) { return cls(a) > cls(b); });
  } else if ((heavy.empty() || rest.empty()) && background.empty()) {
    return SOL_OK;
  }
  std::stable_sort(heavy.begin(), heavy.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
  const uint32_t n_first = rest.empty() ? 0u : (uint32_t)heavy.size();  // (nothing but heavy blocks: they are the chunk-major "rest")
  heavy.insert(heavy.end(), rest.begin(), rest.end());
  heavy.insert(heavy.end(), background.begin(), background.end());
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
  s->S.n_first = n_first;
  s->n_background_local = (uint32_t)background.size();
  return SOL_OK;
}

int sol_set_partition(SolScene* s, int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world) return sol_fail(SOL_EINVAL, "bad partition %d/%d", rank, world);
  const int rank_before = s->rank, world_before = s->world;
  s->rank = rank; s->world = world;
  const uint32_t nb = s->blocks_x * s->blocks_y;
  s->n_local_blocks = (nb + (uint32_t)world - 1u - (uint32_t)rank) / (uint32_t)world;  // blocks b with b % world == rank
  // every rank's compact buffer has the size of rank 0's (the largest) so that a gather has equal counts
  const uint32_t max_blocks = (nb + (uint32_t)world - 1u) / (uint32_t)world;
  // Balanced partition (SOL_OPT_BALANCED_PARTITION): the blocks sorted by their rays in the creation probe, costliest first, are
  // dealt out in rounds of `world`, forwards and backwards in turn (0 .. N-1, N-1 .. 0, ..): every rank gets one block of each
  // round, so the ranks' sums differ by less than one block's cost instead of by what the raster order happens to give b mod N
  // (C3 at 8 ranks: 1.7 % between the fastest and the slowest rank). The probe is deterministic - every rank derives the same table.
  s->local_blocks.clear();
  s->S.block_of_local = nullptr;
  const bool table = s->balanced && world > 1 && s->block_work.size() == nb;
  const uint32_t crc_before = s->partition_crc;
  const bool had_sums = s->acc_own != nullptr && s->acc_floats > 0;
  // Which block sits where: a checksum of the block -> (rank, local block) table that every rank must agree on (SolSceneInfo;
  // bench.py compares it across the ranks before the timed region). Modulo partition: a function of (blocks, world) alone.
  s->partition_table = table ? 1u : 0u;
  s->partition_crc = 0x9E3779B9u * (uint32_t)world + nb;
  if (table) {
    std::vector<uint32_t> order(nb);
    for (uint32_t b = 0; b < nb; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return s->block_work[a] > s->block_work[b]; });
    std::vector<uint32_t> slot(nb);
    for (uint32_t j = 0; j < nb; ++j) {
      const uint32_t g = j / (uint32_t)world, i = j % (uint32_t)world, r = (g & 1u) ? (uint32_t)world - 1u - i : i;
      slot[order[j]] = r * max_blocks + g;
      if (r == (uint32_t)rank) s->local_blocks.push_back(order[j]);  // (local block g: the rounds come in order)
    }
    for (uint32_t b = 0; b < nb; ++b) { s->partition_crc ^= slot[b] + 0x9E3779B9u + (s->partition_crc << 6) + (s->partition_crc >> 2); }
    // a rank without a block in the last, partial round has one local block less; local indices stay dense because every rank
    // takes part in every full round
    s->n_local_blocks = (uint32_t)s->local_blocks.size();
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (!s->slot_of_block) HIP_TRY(hipMalloc((void**)&s->slot_of_block, std::max<size_t>((size_t)nb * 4, 64)));
    HIP_TRY(hipMemcpy(s->slot_of_block, slot.data(), (size_t)nb * 4, hipMemcpyHostToDevice));
    if (s->local_blocks.size() > s->block_of_local_cap) {
      if (s->block_of_local_dev) hipFree(s->block_of_local_dev);
      s->block_of_local_dev = nullptr; s->block_of_local_cap = 0;
      HIP_TRY(hipMalloc((void**)&s->block_of_local_dev, std::max<size_t>(s->local_blocks.size() * 4, 64)));
      s->block_of_local_cap = s->local_blocks.size();
    }
    if (!s->local_blocks.empty()) HIP_TRY(hipMemcpy(s->block_of_local_dev, s->local_blocks.data(), s->local_blocks.size() * 4, hipMemcpyHostToDevice));
    s->S.block_of_local = s->block_of_local_dev;
  } else if (s->slot_of_block) {
    HIP_TRY(hipStreamSynchronize(s->stream));
    hipFree(s->slot_of_block);
    s->slot_of_block = nullptr;
  }
  size_t floats = (size_t)max_blocks * 64u * 3u;
  // a different block -> slot mapping: sums already in the accumulators (and the auxiliary planes) lie in the old layout - cleared,
  // so that a later sol_read cannot mix the two (the caller re-renders; a caller-bound accumulator was refused above)
  // (the checksum is rank-independent - every rank of a job reports the same one -, so a change of rank alone is compared too)
  const bool layout_changed = crc_before != s->partition_crc || rank_before != rank || world_before != world;
  if (had_sums && layout_changed && floats == s->acc_floats && s->acc == s->acc_own) {
    HIP_TRY(hipMemsetAsync(s->acc_own, 0, s->acc_floats * sizeof(float), s->stream));
    for (int k = 0; k < 2; ++k)
      if (s->aux[k] && s->aux_floats == s->acc_floats) HIP_TRY(hipMemsetAsync(s->aux[k], 0, s->aux_floats * sizeof(float), s->stream));
  }
  if (floats != s->acc_floats || !s->acc_own) {
    if (s->acc_own) { hipFree(s->acc_own); s->acc_own = nullptr; }
    HIP_TRY(hipMalloc((void**)&s->acc_own, std::max<size_t>(floats * sizeof(float), 64)));
    HIP_TRY(hipMemset(s->acc_own, 0, std::max<size_t>(floats * sizeof(float), 64)));
    s->acc = s->acc_own;
    s->acc_floats = floats;
  }
  return sol_rebuild_order(s);
}

extern "C" {

const char* sol_last_error(void) { return g_err.c_str(); }

int sol_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sol_record_sizes(uint32_t out[6]) {
  out[0] = sizeof(DWide); out[1] = sizeof(DSphere); out[2] = sizeof(DQuad); out[3] = sizeof(DTri);
  out[4] = sizeof(DTriShade); out[5] = sizeof(DMat);
  return SOL_OK;
}

void sol_scene_destroy(SolScene* s) {
  if (!s) return;
  hipSetDevice(s->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  sol_comm_destroy(s);
  void* ptrs[] = {s->leaf_refs, s->nodes, s->wides, s->tris, s->tri_shade, s->quads, s->spheres, s->mediums, s->mats, s->texs, s->texels, s->env, s->lights, s->light_tri,
                  s->acc_own, s->partial, s->image, s->rgb8, s->work, s->spill, s->counters, s->pool, s->queue, s->wf_ctr,
                  s->bloom_a, s->bloom_b, s->bloom_w, s->aux[0], s->aux[1], s->dscene, s->order_dev, s->block_of_local_dev, s->slot_of_block};
  if (s->wf_ctr_host) hipHostFree(s->wf_ctr_host);
  for (void* p : ptrs)
    if (p) hipFree(p);
  if (s->ev_start) hipEventDestroy(s->ev_start);
  if (s->ev_stop) hipEventDestroy(s->ev_stop);
  if (s->own_stream) hipStreamDestroy(s->own_stream);
  delete s;
}

int sol_scene_info(const SolScene* s, SolSceneInfo* out) {
  if (!s || !out || out->size < 8 || out->size > 4096) return sol_fail(SOL_EINVAL, "bad argument (SolSceneInfo.size?)");
  SolSceneInfo r{};
  r.size = (uint32_t)std::min<size_t>(out->size, sizeof r);
  r.stack_bound = s->tree_depth;
  r.lds_stack = SOL_LDS_STACK;
  r.spill_stack = SOL_SPILL_STACK;
  r.tree_fallback = s->tree_note.empty() ? 0u : 1u;
  std::snprintf(r.tree_name, sizeof r.tree_name, "%s", s->tree_name.c_str());
  std::snprintf(r.tree_note, sizeof r.tree_note, "%s", s->tree_note.c_str());
  r.split_references = s->split_references;
  r.split_triangles = s->split_triangles;
  r.split_area_ratio = s->split_area_ratio;
  r.reinsertion_moves = s->reinsertion_moves;
  r.reinsertion_area_ratio = s->reinsertion_area_ratio;
  r.partition_table = s->partition_table;
  r.partition_crc = s->partition_crc;
  r.strict_triangles = s->strict_triangles ? 1u : 0u;
  r.background_blocks = s->n_background;
  r.background_pixels = s->background_pixels;
  std::memcpy(out, &r, r.size);
  return SOL_OK;
}

int sol_path_stats(const SolScene* s, SolPathStats* out) {
  if (!s || !out || out->size < 8 || out->size > 4096) return sol_fail(SOL_EINVAL, "bad argument (SolPathStats.size?)");
  SolPathStats r = s->path_stats;
  r.size = (uint32_t)std::min<size_t>(out->size, sizeof r);
  std::memcpy(out, &r, r.size);
  return SOL_OK;
}

int sol_scene_set_option(SolScene* s, int option, int64_t value) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
  switch (option) {
    case SOL_OPT_SWITCH_BELOW:
      if (value < 0 || value > 64) return sol_fail(SOL_EINVAL, "SOL_OPT_SWITCH_BELOW: 0..64");
      s->switch_below = (uint32_t)value;
      return SOL_OK;
    case SOL_OPT_MAX_BLOCKS_PER_CU:
      if (value < 0 || value > 64) return sol_fail(SOL_EINVAL, "SOL_OPT_MAX_BLOCKS_PER_CU: 0..64");
      s->max_bpc = (int)value;
      return SOL_OK;
    case SOL_OPT_KERNEL:
      if (value < 0 || value > 4) return sol_fail(SOL_EINVAL, "SOL_OPT_KERNEL: 0..4");
#ifndef SOL_AB_KERNELS
      if (value > 1) return sol_fail(SOL_EINVAL, "SOL_OPT_KERNEL %d: the wavefront and pool variants exist only in -DSOL_AB_KERNELS builds of the library", (int)value);
#endif
      s->kernel_version = (int)value;
      return SOL_OK;
    case SOL_OPT_FINE_TAIL:
      if (value < -1 || value > 64) return sol_fail(SOL_EINVAL, "SOL_OPT_FINE_TAIL: -1 (by the probe), 0 (off) .. 64 quarters of an item per lane");
      s->fine_tail = (int)value;
      return SOL_OK;
    case SOL_OPT_BALANCED_PARTITION:
      if (value != 0 && value != 1) return sol_fail(SOL_EINVAL, "SOL_OPT_BALANCED_PARTITION: 0 or 1");
      if (s->comm) return sol_fail(SOL_EINVAL, "SOL_OPT_BALANCED_PARTITION: set it before sol_comm_init (every rank the same)");
      HIP_TRY(hipSetDevice(s->device));
      if (s->acc != s->acc_own && s->world > 1) return sol_fail(SOL_EINVAL, "unbind the caller's accumulator before changing the partition");
      {  // (validated first; the flag changes only together with the tables that describe the partition)
        const bool before = s->balanced;
        s->balanced = value != 0;
        const int rc = sol_set_partition(s, s->rank, s->world);
        if (rc != SOL_OK) { s->balanced = before; return rc; }
      }
      return SOL_OK;
    case SOL_OPT_BACKGROUND_BLOCKS:
      s->background_enabled = value != 0;
      s->background_in_counted = value == 2;
      return SOL_OK;
    case SOL_OPT_WORK_ORDER:
      s->order_enabled = value != 0;
      HIP_TRY(hipSetDevice(s->device));
      return sol_rebuild_order(s);
    default: return sol_fail(SOL_EINVAL, "unknown option %d", option);
  }
}

int sol_scene_set_partition(SolScene* s, int rank, int world) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (world >= 1 && s->acc != s->acc_own) {
    const uint32_t nb = s->blocks_x * s->blocks_y;
    const size_t floats = (size_t)((nb + (uint32_t)world - 1u) / (uint32_t)world) * 64u * 3u;
    if (floats != s->acc_floats)
      return sol_fail(SOL_EINVAL, "a caller-bound accumulator of %zu floats cannot follow the new partition (%zu floats): unbind it first", s->acc_floats, floats);
  }
  return sol_set_partition(s, rank, world);
}

size_t sol_accum_floats(const SolScene* s) { return s ? s->acc_floats : 0; }
void* sol_accum_ptr(SolScene* s) { return s ? s->acc : nullptr; }

int sol_scene_bind_accum(SolScene* s, void* p, size_t n_floats) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
  if (!p) { s->acc = s->acc_own; return SOL_OK; }
  if (n_floats < s->acc_floats) return sol_fail(SOL_EINVAL, "bound accumulator too small: %zu < %zu floats", n_floats, s->acc_floats);
  s->acc = (float*)p;
  return SOL_OK;
}

int sol_scene_set_stream(SolScene* s, void* stream) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->stream = stream ? (hipStream_t)stream : s->own_stream;
  return SOL_OK;
}

int sol_clear(SolScene* s) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemsetAsync(s->acc, 0, s->acc_floats * sizeof(float), s->stream));
  return SOL_OK;
}

int sol_sync(SolScene* s) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return SOL_OK;
}

int sol_read(SolScene* s, float* rgb_sum) {
  if (!s || !rgb_sum) return sol_fail(SOL_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(sol_launch_unpermute(s->acc, s->image, s->S.width, s->S.height, s->blocks_x, (uint32_t)s->world, (uint32_t)s->rank,
                               s->acc_floats, s->slot_of_block, s->stream));
  HIP_TRY(hipMemcpyAsync(rgb_sum, s->image, (size_t)s->S.width * s->S.height * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return SOL_OK;
}

int sol_kernel_timing(SolScene* s, int enable) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
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
  if (!s || !ms) return sol_fail(SOL_EINVAL, "null argument");
  if (!s->timing || s->timed_launches == 0) return sol_fail(SOL_EINVAL, "no timed render launch (call sol_kernel_timing first)");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipEventSynchronize(s->ev_stop));
  HIP_TRY(hipEventElapsedTime(ms, s->ev_start, s->ev_stop));
  if (grid_blocks) *grid_blocks = s->last_grid;
  return SOL_OK;
}

int sol_stats(const SolScene* s, SolStats* out) {
  if (!s || !out) return sol_fail(SOL_EINVAL, "null argument");
  *out = s->stats;
  return SOL_OK;
}

int sol_read_image(SolScene* s, float* rgb_sum) {
  if (!s || !rgb_sum) return sol_fail(SOL_EINVAL, "null argument");
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
