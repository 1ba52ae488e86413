// sol_render.hip -- the gfx950 render kernels of the path-tracing hot path (wave64, no MFMA: branchy traversal + fp32
// shading). Replaces src/renderer/mod.rs:241-291 (row tasks) and everything under ray_color (:164-206).
//
//  sol_render_kernel -- ONE persistent kernel, one path per lane, state in registers. A wave alternates between the search loop
//    (one resumable BVH step per turn, per-lane stack in LDS) and the service block (shade the closest hit, fetch work, generate
//    the next camera ray); it leaves the search loop as soon as too few of its lanes are still searching while others wait
//    (RenderParams::switch_below), unfinished searches keep their state across the service block. DESIGN.md 3.
//  (The measured alternatives - a wave-private wavefront over a pool of path slots and a two-kernel wavefront, bit-identical
//  frames, slower - live in sol_wavefront.hip, which only the -DSOL_AB_KERNELS build of the library carries.)
//
// A work item's 16 samples are summed in sample order by the lane that owns the item and written once; no float atomic touches
// the accumulator, so the image is a pure function of (scene, seed), bit-identical for any tile partition.
#include <hip/hip_runtime.h>

#include "sol_launch.h"
#include "sol_path.h"

// SPILL = false: built for scenes whose searches fit the LDS stack (sol_api.cpp picks it by the tree's depth): every stack access
// is a plain LDS access, the spill branches and their waits fold away.
// STRICT = true: built for scenes with needle triangles (include/solstrale_hip.h, sol_scene_has_needles): the closest hit of a finished
// search must pass the triangle consistency rule before it is shaded, or the lane searches the same ray again for what lies behind it.
template <bool COUNT, bool MEDIUM, bool SPILL, bool STRICT>
__global__ void __launch_bounds__(SOL_WG, SOL_V1_MIN_WAVES)  // 4 waves per SIMD: the 32 KiB LDS stack allows 5 workgroups per CU, 128 VGPRs 4
sol_render_kernel(const DevScene* __restrict__ Sp, const RenderParams P, float* __restrict__ acc, float* __restrict__ partial,
                  uint32_t* __restrict__ work_counter, uint32_t* __restrict__ spill, DevCounters* __restrict__ dcnt) {
  // The scene record (pointers, camera, ... ~70 dwords) is read through a pointer to constant device memory: passed by value it
  // stays in SGPRs for the whole kernel and 100 of them spill into VGPR lanes around the service block; through the pointer
  // the compiler re-loads fields with scalar loads where it needs them (2 spills; C1 +13 %, C2 +5 %, C3 +1.5 %).
  const DevScene& S = *Sp;
  __shared__ uint32_t lds_stack[SOL_LDS_STACK * SOL_WG];
  const uint32_t tid = threadIdx.x;
  const uint32_t gtid = blockIdx.x * SOL_WG + tid;
  const uint32_t lane = tid & 63u;
  Stack st;
  st.lds = (lds_u32*)lds_stack + tid;
  st.spill = (SOL_AS1 uint32_t*)spill + gtid;
  st.stride = P.total_threads;
  st.depth = SPILL ? SOL_LDS_STACK : SOL_NO_SPILL;
  sol_search_context<true>(st, S);
  __shared__ uint8_t oct_table[SOL_OCT_TABLE_BYTES];
  sol_fill_oct_table((lds_u8*)oct_table, tid, SOL_WG);
  st.oct_table = (const lds_u8*)oct_table;
  st.oct_table_on = true;
  __syncthreads();
  Counters cnt = {};
  const float inf = __builtin_huge_valf();

  bool have_item = false, alive = false, in_flight = false;
  Item it = {0, 0, 0, 0};
  uint32_t s = 0, s_end = 0, out_at = 0;  // out_at: where the item's sum goes, in RGB triples from `partial`
  f3 sum = mk3(0.f, 0.f, 0.f);
  Path p = {};
  Trav t;
  t.cur = REF_DONE;
  uint32_t item_rays0 = 0;  // (counted builds) ray count when the lane took its item
  __shared__ uint32_t reservoir[SOL_WG / 64][2];  // per wave: next reserved item, end of the reservation
  __shared__ uint32_t from_lds[MEDIUM ? SOL_WG : 1];  // (MEDIUM kernels: Path::from of the ray being searched, below)
  if (lane == 0) { reservoir[tid >> 6][0] = 0u; reservoir[tid >> 6][1] = 0u; }

  for (;;) {
    // ---- lanes whose search is over: shade the vertex, then start the next ray of the path / sample / item ----
    // (STRICT) a finished search whose closest hit is a triangle the consistency rule refuses: the lane searches again, behind that hit
    // (rule 8) a finished search whose closest hit is the flat primitive the ray had just left is not shaded: the same ray is searched again behind
    // that hit (through the one trav_begin at the end of this block: a second call site cost the MEDIUM kernels 28 spilled registers)
    // (MEDIUM kernels sit at the 128-register limit: Path::from waits in LDS while the ray is searched - one dword per lane, written when a ray is made, read
    // when its search is over - instead of costing 26 spilled registers)
    const bool again = t.cur == REF_DONE && in_flight && sol_self_hit(MEDIUM ? from_lds[tid] : p.from, t.h);
    if (STRICT && !again && t.cur == REF_DONE && in_flight) trav_accept_or_restart(S, t, st);
    if (t.cur == REF_DONE) {
      float tmin = RAY_MIN_F;
      if (again) {
        p.o = t.o; p.d = t.d;
        tmin = sol_behind(t.h.t);
      } else if (in_flight) {
        in_flight = false;
        if (COUNT) cnt.rays++;
        p.o = t.o; p.d = t.d;  // (the ray lives in the search state while it is traced)
        f3 c;
        if (COUNT && p.depth == 0u && SOL_REF_KIND(t.h.ref) != SOL_REF_NONE) cnt.primary_hits++;
        if (shade_vertex<COUNT, STRICT>(S, p, t.h, c, cnt)) {
          if (COUNT) count_path(cnt, p.depth + 1u);  // (depth counts the scatterings before this vertex)
          sum = sum + c;  // add_row_data (src/renderer/mod.rs:361-365): sums, not means
          alive = false;
          s++;
          if (s == s_end) {
            // the item's sum: a chunk sum for sol_resolve_kernel, or one sample of the fine tail for sol_stage_resolve_kernel -
            // where it goes was worked out when the item was taken
            float* a = partial + (size_t)out_at * 3;
            a[0] = sum.x; a[1] = sum.y; a[2] = sum.z;
            if (COUNT && S.block_cost) {
              atomicMax(S.block_cost + (it.slot >> 6), cnt.rays - item_rays0);  // work order: the longest item decides
              atomicAdd(S.block_work + (it.slot >> 6), cnt.rays - item_rays0);  // balanced partition: the block's rays
            }
            have_item = false;
          }
        }
      }
      // work fetch. The wave keeps a reservoir [next, end) of reserved items in LDS and refills it 64 items at a time (one
      // aligned 8x8 pixel block of one chunk) with ONE returning atomic; lanes take consecutive items from it by popcount
      // prefix. The wave thus waits for the global counter (1-3 us under load, ~88 dequeues/us per address) once per 64
      // items instead of once per service pass.
      if (!again && !have_item) {
        const unsigned long long need = sol_ballot(true);
        const uint32_t leader = (uint32_t)__ffsll((long long)need) - 1u;
        const uint32_t n_need = (uint32_t)__popcll(need), my = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
        // volatile: the leader's stores and every lane's loads must stay real LDS accesses in program order (one wave's LDS
        // operations execute in order; without it the compiler may forward a stale value to the non-leader lanes)
        volatile lds_u32* res = (volatile lds_u32*)reservoir[tid >> 6];
        const uint32_t next = res[0], left = res[1] - next;
        uint32_t fresh = 0;
        if (n_need > left) {  // take what is left, then continue in a fresh block of 64
          if (lane == leader) fresh = atomicAdd(work_counter, 64u);
          fresh = __shfl(fresh, (int)leader);
        }
        if (lane == leader) {
          res[0] = n_need > left ? fresh + (n_need - left) : next + n_need;
          if (n_need > left) res[1] = fresh + 64u;
        }
        const uint32_t item = my < left ? next + my : fresh + (my - left);
        if (item >= P.n_items) break;  // no work left for this lane
        // The fine tail: the last pairs of the work order are handed out sample by sample, so that the launch does not end with
        // every lane inside a 16-sample item of its own (~2.5 ms on C3, 4 ms of a launch whatever its length) but inside a
        // single sample. Which lane computes a sample does not change it; the samples are added up in order afterwards.
        uint32_t citem = item, sub = 0u;
        const bool fine = item >= P.n_coarse;
        if (fine) {
          const uint32_t f = item - P.n_coarse, g = f >> 6;
          citem = P.n_coarse + ((g >> 4) << 6) + (f & 63u);
          sub = g & 15u;
          if (sub >= P.fine_count) continue;
        }
        if (!decode_item_ordered(S, P, citem, it)) continue;
        s = P.first_sample + it.chunk * SOL_CHUNK + sub;
        s_end = fine ? s + 1u : min(s + SOL_CHUNK, P.first_sample + P.n_samples);
        out_at = fine ? P.stage_at + (citem - P.n_coarse) * SOL_CHUNK + sub : it.chunk * (P.n_local_blocks * 64u) + it.slot;
        sum = mk3(0.f, 0.f, 0.f);
        have_item = true;
        alive = false;
        if (COUNT) item_rays0 = cnt.rays;
      }
      if (!again && !alive) {
        phase_tick<COUNT>(cnt, 2);
        generate_path<COUNT>(S, P.seed_lo, P.seed_hi, it.px, it.py, s, p, cnt);
        alive = true;
      }
      // world.hit(ray, RAY_INTERVAL) (src/renderer/mod.rs:165)
      if (MEDIUM && !again) from_lds[tid] = p.from;  // (of the ray made above: scattered by shade_vertex, or a camera ray)
      const float bt = t.bt;
      const uint32_t bdfs = t.bdfs;
      trav_begin<true>(t, p.o, p.d, tmin, inf, S.wroot, S.rxmin, S.rxmax, S.rymin, S.rymax, S.rzmin, S.rzmax, 0);
      if (STRICT && again) { t.bt = bt; t.bdfs = bdfs; }  // (a bound the needle rule had set for this ray stays)
      in_flight = true;
    }
    // ---- search: one step per turn for every lane that has one. The wave leaves for the shading block when too few of
    // its lanes are still searching while others wait (P.switch_below 64ths of the live lanes; 0: when none is searching).
#if SOL_LOOP_PRIO
    __builtin_amdgcn_s_setprio(SOL_LOOP_PRIO);
#endif
    for (;;) {
      const bool act = t.cur != REF_DONE;
      const unsigned long long am = sol_ballot(act);
      if (am == 0ull) break;
      const unsigned long long live = sol_ballot(true);
      if (am != live && (uint32_t)__popcll(am) * 64u < P.switch_below * (uint32_t)__popcll(live)) break;
      trav_step_wave<COUNT, MEDIUM, STRICT>(S, t, act, st, p.rng, p.depth, cnt);
    }
#if SOL_LOOP_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  }
  if (COUNT) flush_counters(cnt, dcnt);
}

// The fine tail's second half: one thread per (pair, pixel) of the tail adds the pair's samples in order - the sum a lane would
// have formed had it taken the whole chunk - and writes it where that lane would have.
__global__ void __launch_bounds__(256) sol_stage_resolve_kernel(const DevScene* __restrict__ Sp, const RenderParams P, float* __restrict__ partial) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= P.n_items - P.n_coarse) return;  // (the launcher passes n_items = n_coarse + pixels of the tail here)
  Item it;
  if (!decode_item_ordered(*Sp, P, P.n_coarse + i, it)) return;
  f3 sum = mk3(0.f, 0.f, 0.f);
  const float* a = partial + ((size_t)P.stage_at + (size_t)i * SOL_CHUNK) * 3;
  // (the fine tail lies in the last chunk; the pool kernel hands out EVERY chunk sample by sample: whole chunks but the last)
  const uint32_t count = it.chunk + 1u == P.n_chunks ? P.fine_count : (uint32_t)SOL_CHUNK;
  for (uint32_t k = 0; k < count; ++k) sum = sum + mk3(a[3 * k], a[3 * k + 1], a[3 * k + 2]);
  float* o = partial + ((size_t)it.chunk * (P.n_local_blocks * 64u) + it.slot) * 3;
  o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
}
hipError_t sol_launch_stage_resolve(const DevScene* dS, const RenderParams& P, float* partial, hipStream_t stream) {
  RenderParams Q = P;
  const uint32_t pixels = (P.n_items - P.n_coarse) / SOL_CHUNK;
  if (pixels == 0) return hipSuccess;
  Q.n_items = P.n_coarse + pixels;
  hipLaunchKernelGGL(sol_stage_resolve_kernel, dim3((pixels + 255u) / 256u), dim3(256), 0, stream, dS, Q, partial);
  return hipGetLastError();
}

// Background blocks (include/solstrale_hip.h, SolSceneInfo::background_blocks): the blocks behind the first n_traced_blocks of the work
// order were proved at scene creation to see nothing but the constant background - every sample of their pixels is a camera ray that
// comes near no primitive's box, colour = the background (shade_vertex's miss: x = background, c = A * x with A = 1). One thread per
// (background block, pixel, chunk) writes the chunk sum the render kernel's lane would have written: the same additions in the same order.
__global__ void __launch_bounds__(256) sol_fill_background_kernel(const DevScene* __restrict__ Sp, const RenderParams P, float* __restrict__ partial) {
  const DevScene& S = *Sp;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t per_chunk = (P.n_local_blocks - P.n_traced_blocks) * 64u;
  if (i >= per_chunk * P.n_chunks) return;
  const uint32_t chunk = i / per_chunk, r = i - chunk * per_chunk;
  const uint32_t k = P.n_traced_blocks + (r >> 6), pin = r & 63u;
  const uint32_t lb = ldg_u32(S.block_order + k);
  const uint32_t b = S.block_of_local ? ldg_u32(S.block_of_local + lb) : lb * P.world + P.rank;
  const uint32_t by = b / P.blocks_x, bx = b - by * P.blocks_x;
  if (bx * SOL_TILE + (pin & 7u) >= S.width || by * SOL_TILE + (pin >> 3) >= S.height) return;  // padding pixel of an edge block
  f3 c = mk3(S.bgx, S.bgy, S.bgz);
  if (S.shader == SOL_SHADER_PATH_TRACING) c = mk3(1.f, 1.f, 1.f) * c;
  const uint32_t count = min((uint32_t)SOL_CHUNK, P.n_samples - chunk * SOL_CHUNK);
  f3 sum = mk3(0.f, 0.f, 0.f);
  for (uint32_t j = 0; j < count; ++j) sum = sum + c;
  float* o = partial + ((size_t)chunk * (P.n_local_blocks * 64u) + lb * 64u + pin) * 3;
  o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
}
hipError_t sol_launch_fill_background(const DevScene* dS, const RenderParams& P, float* partial, hipStream_t stream) {
  const uint32_t n = (P.n_local_blocks - P.n_traced_blocks) * 64u * P.n_chunks;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(sol_fill_background_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, dS, P, partial);
  return hipGetLastError();
}

// Diagnostic: one path (pixel, sample) on one lane, every ray and its closest hit recorded: 12 floats per ray
// (origin, direction, t, ref bits, dfs bits, depth, 0, 0), then a terminator row (colour in the first 3 floats, -1 in the 4th).
template <bool MEDIUM, bool STRICT>
__global__ void __launch_bounds__(SOL_WG)
sol_debug_path_kernel(const DevScene S, const RenderParams P, uint32_t px, uint32_t py, uint32_t s, uint32_t* __restrict__ spill,
                      float* __restrict__ out, uint32_t max_rows) {
  __shared__ uint32_t lds_stack[SOL_LDS_STACK * SOL_WG];
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Stack st;
  st.lds = (lds_u32*)lds_stack;
  st.spill = (SOL_AS1 uint32_t*)spill;
  st.stride = P.total_threads;
  st.depth = SOL_LDS_STACK;
  sol_search_context<false>(st, S);
  Counters cnt = {};
  const float inf = __builtin_huge_valf();
  Path p = {};
  generate_path<false>(S, P.seed_lo, P.seed_hi, px, py, s, p, cnt);
  uint32_t row = 0;
  for (;;) {
    Hit h;
    closest_hit<false, MEDIUM>(S, p.o, p.d, RAY_MIN_F, inf, h, st, 0, p.rng, p.depth, cnt, p.from);
    if (row + 1 < max_rows) {
      float* o = out + (size_t)row * 12;
      o[0] = p.o.x; o[1] = p.o.y; o[2] = p.o.z; o[3] = p.d.x; o[4] = p.d.y; o[5] = p.d.z; o[6] = h.t;
      o[7] = __uint_as_float(h.ref); o[8] = __uint_as_float(h.dfs); o[9] = (float)p.depth; o[10] = 0.f; o[11] = 0.f;
      row++;
    }
    f3 c;
    if (shade_vertex<false, STRICT>(S, p, h, c, cnt)) {
      float* o = out + (size_t)row * 12;
      o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = -1.0f;
      return;
    }
  }
}
hipError_t sol_launch_debug_path(const DevScene& S, const RenderParams& P, uint32_t px, uint32_t py, uint32_t s, uint32_t* spill,
                                 float* out, uint32_t max_rows, bool medium, hipStream_t stream) {
#define DEBUG_PATH(M, ST) hipLaunchKernelGGL((sol_debug_path_kernel<M, ST>), dim3(1), dim3(SOL_WG), 0, stream, S, P, px, py, s, spill, out, max_rows)
  if (S.tri_delta > 0.0f) { if (medium) DEBUG_PATH(true, true); else DEBUG_PATH(false, true); }
  else { if (medium) DEBUG_PATH(true, false); else DEBUG_PATH(false, false); }
#undef DEBUG_PATH
  return hipGetLastError();
}

// ---- launch wrappers (called from sol_launch.cpp) ----
template <bool COUNT, bool MEDIUM, bool SPILL, bool STRICT>
static hipError_t launch_v1(const DevScene* dS, const RenderParams& P, float* acc, float* partial, uint32_t* work,
                            uint32_t* spill, DevCounters* cnt, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL((sol_render_kernel<COUNT, MEDIUM, SPILL, STRICT>), dim3(grid), dim3(SOL_WG), 0, stream, dS, P, acc, partial, work, spill, cnt);
  return hipGetLastError();
}
template <bool STRICT>
static hipError_t launch_v1_any(const DevScene* dS, const RenderParams& P, float* acc, float* partial, uint32_t* work, uint32_t* spill, DevCounters* cnt,
                                uint32_t grid, bool count, bool medium, bool may_spill, hipStream_t stream) {
  if (count) return medium ? launch_v1<true, true, true, STRICT>(dS, P, acc, partial, work, spill, cnt, grid, stream)
                           : launch_v1<true, false, true, STRICT>(dS, P, acc, partial, work, spill, cnt, grid, stream);
  if (may_spill) return medium ? launch_v1<false, true, true, STRICT>(dS, P, acc, partial, work, spill, cnt, grid, stream)
                               : launch_v1<false, false, true, STRICT>(dS, P, acc, partial, work, spill, cnt, grid, stream);
  return medium ? launch_v1<false, true, false, STRICT>(dS, P, acc, partial, work, spill, cnt, grid, stream)
                : launch_v1<false, false, false, STRICT>(dS, P, acc, partial, work, spill, cnt, grid, stream);
}

hipError_t sol_launch_render(int version, const DevScene& S, const DevScene* dS, const RenderParams& P, float* acc, float* partial, uint32_t* work,
                             uint32_t* spill, void* pool, DevCounters* cnt, uint32_t grid, bool count, bool medium, bool may_spill,
                             hipStream_t stream) {
#ifdef SOL_AB_KERNELS
  if (version == 4) return sol_launch_pool4(S, dS, P, partial, work, spill, grid, medium, may_spill, count ? cnt : nullptr, stream);  // (sol_pool.hip)
#endif
  if (version == 1)
    return S.tri_delta > 0.0f ? launch_v1_any<true>(dS, P, acc, partial, work, spill, cnt, grid, count, medium, may_spill, stream)
                              : launch_v1_any<false>(dS, P, acc, partial, work, spill, cnt, grid, count, medium, may_spill, stream);
#ifdef SOL_AB_KERNELS
  return sol_launch_pool(S, P, acc, partial, work, spill, pool, cnt, grid, count, medium, stream);  // (sol_wavefront.hip)
#else
  return hipErrorInvalidValue;  // (the wavefront variants exist in -DSOL_AB_KERNELS builds only)
#endif
}

template <typename K>
static int blocks_per_cu(K kernel) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, SOL_WG, 0) != hipSuccess || n < 1) n = 1;
  return n;
}
int sol_render_blocks_per_cu(int version, bool count, bool medium, bool strict) {
#ifdef SOL_AB_KERNELS
  if (version == 4) return sol_pool4_blocks_per_cu(medium, strict);
#endif
  if (version == 1) {  // (the SPILL = false builds need no more registers or LDS than these)
    if (strict) {
      if (count) return medium ? blocks_per_cu(sol_render_kernel<true, true, true, true>) : blocks_per_cu(sol_render_kernel<true, false, true, true>);
      return medium ? blocks_per_cu(sol_render_kernel<false, true, true, true>) : blocks_per_cu(sol_render_kernel<false, false, true, true>);
    }
    if (count) return medium ? blocks_per_cu(sol_render_kernel<true, true, true, false>) : blocks_per_cu(sol_render_kernel<true, false, true, false>);
    return medium ? blocks_per_cu(sol_render_kernel<false, true, true, false>) : blocks_per_cu(sol_render_kernel<false, false, true, false>);
  }
#ifdef SOL_AB_KERNELS
  return sol_pool_blocks_per_cu(count, medium);
#else
  return 1;
#endif
}
