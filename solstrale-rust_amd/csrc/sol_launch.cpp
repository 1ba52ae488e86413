// sol_launch.cpp -- sol_render and its relatives: launch parameters, scratch buffers and launches of the render kernels
// (sol_render.hip), the auxiliary albedo / normal planes, the function-level and path-level debug hooks.
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

int sol_render_impl(SolScene* s, uint32_t first, uint32_t n, uint64_t seed, bool count) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
  if (n == 0) return SOL_OK;
  if ((uint64_t)first + n > 0xFFFFFFFFull) return sol_fail(SOL_EINVAL, "sample range overflows 32 bits");
  HIP_TRY(hipSetDevice(s->device));
  RenderParams P{};
  P.first_sample = first; P.n_samples = n;
  P.n_chunks = (uint32_t)(((uint64_t)n + SOL_CHUNK - 1) / SOL_CHUNK);  // (in 64 bits: n + 15 wraps for the last fifteen values of n)
  P.rank = (uint32_t)s->rank; P.world = (uint32_t)s->world;
  P.n_local_blocks = s->n_local_blocks; P.blocks_x = s->blocks_x;
  P.seed_lo = (uint32_t)seed; P.seed_hi = (uint32_t)(seed >> 32);
  int version = s->kernel_version ? s->kernel_version : 1;
  // version 4, the pool kernel (sol_pool.hip): plain renders of the path-tracing... any shader; counted renders (probes, statistics) and trees
  // of 2^17 wide nodes or more (its one-dword node groups carry a 17-bit base) stay with the one-path-per-lane kernel
  if (version == 4 && ((count && !std::getenv("SOL_POOL_COUNT")) || s->n_wide >= SOL_PACK_MAX_NODES)) version = 1;  // (SOL_POOL_COUNT: phase statistics of the pool kernel)
  // one launch of the pool kernel stages ONE COLOUR PER SAMPLE: a long sample range goes through several launches (chunk-aligned, so the sums
  // do not depend on the split), each within the staging budget
  if (version == 4) {
    const uint64_t pairs_per_chunk = (uint64_t)s->n_local_blocks;
    const uint64_t max_chunks = std::max<uint64_t>(1, std::min<uint64_t>((6ull << 30) / std::max<uint64_t>(1, pairs_per_chunk * 64u * SOL_CHUNK * 12u),
                                                                         (uint64_t)SOL_MAX_ITEMS / std::max<uint64_t>(1, pairs_per_chunk * 64u * SOL_CHUNK)));
    if (((uint64_t)n + SOL_CHUNK - 1) / SOL_CHUNK > max_chunks) {
      uint32_t f = first, left = n;
      while (left) {
        const uint32_t k = (uint32_t)std::min<uint64_t>(left, max_chunks * SOL_CHUNK);
        const int rc = sol_render_impl(s, f, k, seed, false);
        if (rc != SOL_OK) return rc;
        f += k; left -= k;
      }
      return SOL_OK;
    }
  }
  // Background blocks - the tail of the work order - are not traced: their sums are written by sol_fill_background_kernel. Counted
  // renders trace everything (their counters describe the whole algorithm; the creation probes are counted renders).
  P.n_traced_blocks = P.n_local_blocks;
  if (version == 1 && (!count || s->background_in_counted) && s->background_enabled && s->S.block_order && s->n_background_local <= P.n_local_blocks)
    P.n_traced_blocks = P.n_local_blocks - s->n_background_local;
  const uint64_t items = (uint64_t)P.n_chunks * P.n_traced_blocks * 64u;
  // the 32-bit work counter keeps counting after the items run out (every wave adds 64 per refused fetch until all its
  // lanes have left): 16 M of headroom is > 100 times what 5120 resident waves can add
  if ((uint64_t)P.n_chunks * P.n_local_blocks * 64u > SOL_MAX_ITEMS) return sol_fail(SOL_EINVAL, "too many work items in one call (%llu): split the sample range", (unsigned long long)items);
  P.n_items = (uint32_t)items;
  if (P.n_local_blocks == 0) return SOL_OK;
  P.switch_below = s->switch_below;
  // Kernel choice. The product library carries ONE render kernel family, the one-path-per-lane kernel (version 1); the two
  // wavefront variants (2: wave-private pool, 3: two-kernel wavefront; bit-identical images) exist in -DSOL_AB_KERNELS builds for
  // A/B runs: they raise the search's lane occupancy (0.45 -> 0.67-0.73) but pay for it in state traffic, refill stalls and
  // per-round tails (MI355X, C3, 128 spp: v1 997, v2 905, v3 684 Msamples/s when they were last compared).
  if (version != 1 && version != 4 && s->strict_triangles) return sol_fail(SOL_EINVAL, "kernel variant %d does not implement the consistency rule of scenes with needle triangles", version);
#ifndef SOL_AB_KERNELS
  if (version != 1) return sol_fail(SOL_EINVAL, "kernel variant %d exists only in -DSOL_AB_KERNELS builds of the library", version);
  int bpc = sol_render_blocks_per_cu(version, count, s->has_medium, s->strict_triangles);
#else
  int bpc = version == 3 ? sol_wf_trace_blocks_per_cu(count, s->has_medium) : sol_render_blocks_per_cu(version, count, s->has_medium, s->strict_triangles);
#endif
  if (s->max_bpc > 0) bpc = std::max(1, std::min(bpc, s->max_bpc));  // SOL_OPT_MAX_BLOCKS_PER_CU
  uint32_t grid = (uint32_t)(s->n_cu * bpc);
  const uint32_t need_blocks = std::max(1u, (P.n_items + SOL_WG - 1) / SOL_WG);
  if (grid > need_blocks) grid = need_blocks;
  P.total_threads = grid * SOL_WG;
  uint32_t lds_depth = (uint32_t)SOL_LDS_STACK;
  uint32_t stack_need = s->tree_depth;  // dwords of traversal stack a search of this scene can use
#ifdef SOL_AB_KERNELS
  // (swap_min: lanes waiting before an exchange pass of the search loop; 64 = only when the wave would otherwise leave for the service block, the best setting measured)
  if (version == 4) { lds_depth = (uint32_t)sol_pool4_lds_stack_depth(); stack_need = s->packed_depth; P.swap_min = s->pool_swap_min ? s->pool_swap_min : 64u; }
  if (version == 3) lds_depth = (uint32_t)sol_wf_lds_stack_depth();
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
#endif
  // spill stack only when the tree can out-grow the LDS stack
  size_t spill_words = stack_need > lds_depth ? (size_t)P.total_threads * (stack_need - lds_depth) : 16;
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
  P.stage_at = P.n_chunks * P.n_local_blocks * 64u;  // right behind the chunk sums (of every block, traced or not)
  size_t stage_floats = 0;
  const int fine_tail = s->fine_tail >= 0 ? s->fine_tail : s->fine_tail_auto;
  if (version == 4) {  // every pair (block, chunk) is handed out sample by sample: the staging area holds the whole launch
    const uint64_t pairs = (uint64_t)P.n_chunks * P.n_traced_blocks;
    const uint64_t total = pairs * 64u * SOL_CHUNK;
    if (total > SOL_MAX_ITEMS || (uint64_t)P.stage_at + total > 0xFFFFFFFFull) return sol_fail(SOL_EINVAL, "too many work items in one call (%llu): split the sample range", (unsigned long long)total);
    stage_floats = (size_t)total * 3u;
    P.n_coarse = 0;
    P.n_items = (uint32_t)total;
    // items per reservation of a wave's reservoir: a whole pair (1024 samples) when every resident wave gets at least 16 of them, else 64
    P.pool_slots = total >= (uint64_t)(P.total_threads / 64u) * 1024u * 16u ? 1024u : 64u;
  } else if (version == 1 && !count && fine_tail > 0) {
    const uint32_t rest = P.n_traced_blocks - std::min(P.n_traced_blocks, s->S.n_first);
    const uint32_t pairs = std::min<uint32_t>(rest, (uint32_t)(((uint64_t)fine_tail * (P.total_threads / 64u) + 3u) / 4u));
    const uint64_t total = items - (uint64_t)pairs * 64u + (uint64_t)pairs * 64u * SOL_CHUNK;
    if (pairs > 0 && total <= SOL_MAX_ITEMS && (uint64_t)P.stage_at + (uint64_t)pairs * 64u * SOL_CHUNK <= 0xFFFFFFFFull) {
      stage_floats = (size_t)pairs * 64u * SOL_CHUNK * 3u;
      P.n_coarse = (uint32_t)(items - (uint64_t)pairs * 64u);
      P.n_items = (uint32_t)total;
    }
  }
  // v1 writes every chunk sum into `partial` (also when the call has a single chunk); v2 / v3 add a single chunk straight
  // into the accumulator
  const bool via_partial = version == 1 || version == 4 || P.n_chunks > 1;
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
#ifdef SOL_AB_KERNELS
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
        if (rounds > 4000000u) return sol_fail(SOL_EDEVICE, "wavefront did not drain");
      }
    }
    s->last_rounds = rounds;
  } else
#endif
  {
    if (!s->dscene) HIP_TRY(hipMalloc((void**)&s->dscene, sizeof(DevScene)));
    if (!s->dscene_valid || std::memcmp(&s->S, &s->S_uploaded, sizeof(DevScene)) != 0) {
      // rare (scene creation, tree probe, auxiliary renders): launches already queued may still read the old copy
      HIP_TRY(hipStreamSynchronize(s->stream));
      HIP_TRY(hipMemcpy(s->dscene, &s->S, sizeof(DevScene), hipMemcpyHostToDevice));
      std::memcpy(&s->S_uploaded, &s->S, sizeof(DevScene));
      s->dscene_valid = true;
    }
    if (P.n_items > 0)
      HIP_TRY(sol_launch_render(version, s->S, s->dscene, P, s->acc, s->partial, s->work, s->spill, s->pool, s->counters, grid, count,
                                s->has_medium, stack_need > lds_depth, s->stream));
    if (P.n_traced_blocks != P.n_local_blocks) HIP_TRY(sol_launch_fill_background(s->dscene, P, s->partial, s->stream));
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
    s->path_stats.size = sizeof(SolPathStats);
    s->path_stats.samples = c.samples; s->path_stats.primary_hits = c.primary_hits;
    for (int k = 0; k < 6; ++k) s->path_stats.path_len[k] = c.path_len[k];
  }
  return SOL_OK;
}

extern "C" {

int sol_render(SolScene* s, uint32_t first, uint32_t n, uint64_t seed) { return sol_render_impl(s, first, n, seed, false); }
int sol_render_counted(SolScene* s, uint32_t first, uint32_t n, uint64_t seed) { return sol_render_impl(s, first, n, seed, true); }

// Auxiliary albedo / normal buffers (src/renderer/mod.rs:175-204): at depth 0 the reference evaluates AlbedoShader and
// NormalShader on the hit of the primary ray (background / zero on a miss) and accumulates them beside the pixel colour.
// Those are exactly the single-hit shaders of this library evaluated on the same primary ray - same (seed, pixel, sample)
// key, hence the same jitter and camera ray - so the two planes are two primary-ray-only renders into their own
// accumulators; no path-tracing kernel variant is needed. (For Blend materials the reference's extra scatter call draws its
// branch independently of the path's, as the separate render does.)
int sol_render_aux(SolScene* s, uint32_t first, uint32_t n, uint64_t seed) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
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
    rc = sol_render_impl(s, first, n, seed, false);
  }
  s->acc = acc;
  s->S.shader = shader;
  s->S.bgx = bg[0]; s->S.bgy = bg[1]; s->S.bgz = bg[2];
  s->S.env = env;
  return rc;
}

int sol_clear_aux(SolScene* s) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  for (int k = 0; k < 2; ++k)
    if (s->aux[k] && s->aux_floats == s->acc_floats) HIP_TRY(hipMemsetAsync(s->aux[k], 0, s->aux_floats * sizeof(float), s->stream));
  return SOL_OK;
}

int sol_read_aux(SolScene* s, float* albedo_sum, float* normal_sum) {
  if (!s || (!albedo_sum && !normal_sum)) return sol_fail(SOL_EINVAL, "null argument");
  if (!s->aux[0] || s->aux_floats != s->acc_floats) return sol_fail(SOL_EINVAL, "no auxiliary buffers: call sol_render_aux first");
  HIP_TRY(hipSetDevice(s->device));
  float* outs[2] = {albedo_sum, normal_sum};
  for (int k = 0; k < 2; ++k) {
    if (!outs[k]) continue;
    HIP_TRY(sol_launch_unpermute(s->aux[k], s->image, s->S.width, s->S.height, s->blocks_x, (uint32_t)s->world, (uint32_t)s->rank,
                                 s->acc_floats, s->slot_of_block, s->stream));
    HIP_TRY(hipMemcpyAsync(outs[k], s->image, (size_t)s->S.width * s->S.height * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  return SOL_OK;
}

int sol_eval(int device, uint32_t fn, const float* in, uint32_t n, uint32_t in_stride, float* out, uint32_t out_stride) {
  if (!in || !out || !in_stride || !out_stride) return sol_fail(SOL_EINVAL, "bad argument");
  // floats a row of sol_eval_kernel (sol_aux.hip) reads and writes, per function: a narrower stride would read and write past the device copies
  static const uint32_t row_in[9] = {3, 3, 5, 7, 13, 24, 17, 12, 4}, row_out[9] = {7, 5, 2, 18, 2, 4, 4, 2, 7};
  if (fn > 8u) return sol_fail(SOL_EINVAL, "sol_eval: unknown function %u", fn);
  if (in_stride < row_in[fn] || out_stride < row_out[fn])
    return sol_fail(SOL_EINVAL, "sol_eval: function %u reads %u and writes %u floats per row (strides %u / %u)", fn, row_in[fn], row_out[fn], in_stride, out_stride);
  if ((uint64_t)n * std::max(in_stride, out_stride) > (1ull << 32)) return sol_fail(SOL_EINVAL, "sol_eval: more than 2^32 floats");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sol_fail(SOL_EDEVICE, "no HIP device available");
  HIP_TRY(hipSetDevice(device));
  float *din = nullptr, *dout = nullptr;
  const size_t ib = (size_t)n * in_stride * sizeof(float), ob = (size_t)n * out_stride * sizeof(float);
  HIP_TRY(hipMalloc((void**)&din, std::max<size_t>(ib, 64)));
  if (hipMalloc((void**)&dout, std::max<size_t>(ob, 64)) != hipSuccess) { hipFree(din); return sol_fail(SOL_ENOMEM, "hipMalloc failed"); }
  hipError_t e = hipMemcpy(din, in, ib, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(dout, 0, std::max<size_t>(ob, 64));
  if (e == hipSuccess) e = sol_launch_eval(fn, din, n, in_stride, dout, out_stride, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, dout, ob, hipMemcpyDeviceToHost);
  hipFree(din);
  hipFree(dout);
  if (e != hipSuccess) return sol_fail(SOL_EDEVICE, "sol_eval: %s", hipGetErrorString(e));
  return SOL_OK;
}

int sol_debug_path(SolScene* s, uint32_t x, uint32_t y, uint32_t sample, uint64_t seed, float* rows, uint32_t max_rows) {
  if (!s || !rows || max_rows < 2 || x >= s->S.width || y >= s->S.height) return sol_fail(SOL_EINVAL, "bad argument");
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
  if (e != hipSuccess) return sol_fail(SOL_EDEVICE, "sol_debug_path: %s", hipGetErrorString(e));
  for (uint32_t r = 0; r < max_rows && rows[r * 12 + 3] != -1.0f; ++r) {  // hit references: device order -> the caller's indices
    uint32_t ref;
    std::memcpy(&ref, &rows[r * 12 + 7], 4);
    const uint32_t kind = SOL_REF_KIND(ref);
    const int a = kind == SOL_REF_TRIANGLE ? 0 : kind == SOL_REF_SPHERE ? 1 : kind == SOL_REF_QUAD ? 2 : -1;  // (SolScene::old_index)
    if (a >= 0 && SOL_REF_INDEX(ref) < s->old_index[a].size()) {
      ref = SOL_MAKE_REF(SOL_REF_KIND(ref), s->old_index[a][SOL_REF_INDEX(ref)]);
      std::memcpy(&rows[r * 12 + 7], &ref, 4);
    }
  }
  return SOL_OK;
}

}  // extern "C"
