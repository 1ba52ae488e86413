// sol_pool.hip -- the POOL render kernel (kernel version 4): the one-path-per-lane kernel of sol_render.hip with a second path context per
// lane kept in LDS and handed out WAVE-WIDE. Same device functions (sol_path.h, sol_shade.h, sol_trace.h), same frames bit for bit.
// MEASURED SLOWER than the product kernel (round 5, profiles/r05_pool_kernel_ab.txt: C3 -7 % at its best setting, C2 -1 %; it raises the
// search loop's occupancy - 41.7 -> 50.7 of 64 lanes - and pays more than that in exchange passes and thinner shading passes) and kept, like
// the wavefront variants of sol_wavefront.hip, in the -DSOL_AB_KERNELS build of the library only (SOL_KERNEL=v4 / SOL_OPT_KERNEL 4).
//
// Why. sol_render_kernel is bound by vector-instruction issue at 0.49 lane utilisation; half of the idle lane-slots of its search loop
// are lanes whose search is over and that wait for the wave's switch to the service block. Here such a lane does not wait:
// it parks its finished context - path state, ray, closest hit: 22 dwords - in a slot of the wave's POOL and takes a context whose ray is
// ready from another slot (any slot: the hand-out is by ballot + popcount prefix over two small queues, not lane-private), so the search
// loop keeps running while the pool holds ready rays. The service block shades the finished contexts the lanes hold, then exchanges the
// fresh rays against parked hits and shades those, pass by pass, until the pool holds only ready rays again.
//
// What pays for it. (1) Node groups go on the traversal stack as ONE dword (sol_trace.h, wide_visit<PACK>: 17-bit base, 7-bit inner mask,
// 8 ordered hit bits; trees below 2^17 wide nodes): 16 KiB instead of 32 per workgroup, 16 levels either way. (2) A context is ONE SAMPLE:
// work items are single samples (the fine tail of sol_render.hip made general - its cost is nil, profiles/r05_sample_granular_items.txt),
// every sample's colour goes to the staging area and sol_stage_resolve_kernel adds each pixel's colours in sample order, the sums a lane
// would have formed. No lane carries an item, a sample counter or a running sum, and a context can finish in any lane.
// LDS per workgroup: 16 KiB stack + 22 KiB pool (4 waves x 64 slots x 88 B) + 1 KiB octant table + queues = 39.5 KiB: four workgroups per
// CU, as before. Counted renders (creation probes, statistics) stay with sol_render_kernel.
#ifdef SOL_AB_KERNELS  // an A/B variant: not part of the product library (build.py)
#include <hip/hip_runtime.h>

#include "sol_launch.h"
#include "sol_path.h"

// (RenderParams::pool_slots = items per reservation of the wave's reservoir: 1024 - 16 samples x 64 pixels of one (block, chunk) pair - in a
// long launch, 64 in a short one, whose tail would otherwise be a thousand samples per wave)
#define POOL_LDS_STACK 16  // one-dword node groups per lane in LDS (= the 16 levels of sol_render_kernel's 32 dwords)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));  // (native vectors: HIP's uint4 class cannot be assigned through an LDS-qualified pointer)
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v4u lds_u4;
typedef __attribute__((address_space(3))) v2u lds_u2;

// A context's records (LDS, [record][slot of the workgroup]):
//   0: o.xyz, accumulated ray length     1: d.xyz, depth | pdf_seen << 31     2: A.xyz, rng.k0     3: C.xyz, rng.k1
//   4: hit t, reference, u, v (a parked HIT); Path::from in .x (a parked RAY: rule 8)   5 (two dwords): rng.ctr, out_at
struct PoolCtx {
  v4u r0, r1, r2, r3, r4;
  v2u r5;
};


// The next sample of the work order for every lane that calls (the callers are the wave's lanes without a context): the reservoir of
// sol_render.hip - 64 consecutive items, one sample index of one 8x8 pixel block, per returning atomic -, item -> (pair (block, chunk),
// sample of the chunk, pixel of the block), camera ray. False: the item names no sample (a padding pixel of an edge block, a sample beyond
// a ragged last chunk) or the work has run out (no_work).
DEV bool pool_fetch_sample(const DevScene& S, const RenderParams& P, volatile lds_u32* res, uint32_t* __restrict__ work_counter, uint32_t lane,
                           unsigned long long below, bool& no_work, Path& p, uint32_t& out_at, Counters& cnt) {
  const unsigned long long need = sol_ballot(true);
  const uint32_t leader = (uint32_t)__ffsll((long long)need) - 1u;
  const uint32_t n_need = (uint32_t)__popcll(need), my = (uint32_t)__popcll(need & below);
  // A reservation is P.pool_slots items - in a long launch one pair (block, chunk) with all its samples -, so that the global counter sees as many atomics as
  // under whole items (one per 1024 samples; at one per 64 the counter's ~88 dequeues per microsecond bound C1 and weighed on the rest).
  // res[2]: the wave has been refused once - it stops asking (every refused reservation moves the 32-bit counter on by a reservation).
  const uint32_t next = res[0], left = res[1] - next;
  const bool dry = res[2] != 0u;
  uint32_t fresh = 0;
  if (n_need > left && !dry) {
    if (lane == leader) fresh = atomicAdd(work_counter, P.pool_slots);
    fresh = __shfl(fresh, (int)leader);
  }
  if (lane == leader) {
    if (n_need <= left) res[0] = next + n_need;
    else if (dry) res[0] = res[1];  // (what was left is handed out below; nothing more comes)
    else { res[0] = fresh + (n_need - left); res[1] = fresh + P.pool_slots; if (fresh >= P.n_items) res[2] = 1u; }
  }
  if (my >= left && dry) { no_work = true; return false; }
  const uint32_t item = my < left ? next + my : fresh + (my - left);
  if (item >= P.n_items) { no_work = true; return false; }
  const uint32_t g = item >> 6, pair = g >> 4, sub = g & 15u, citem = (pair << 6) + (item & 63u);
  Item it;
  if (!decode_item_ordered(S, P, citem, it)) return false;
  const uint32_t count = it.chunk + 1u == P.n_chunks ? P.fine_count : (uint32_t)SOL_CHUNK;
  if (sub >= count) return false;
  out_at = P.stage_at + citem * SOL_CHUNK + sub;
  generate_path<false>(S, P.seed_lo, P.seed_hi, it.px, it.py, P.first_sample + it.chunk * SOL_CHUNK + sub, p, cnt);
  return true;
}

template <bool MEDIUM, bool SPILL, bool STRICT, bool COUNT = false>
__global__ void __launch_bounds__(SOL_WG, SOL_V1_MIN_WAVES)
sol_render_pool4_kernel(const DevScene* __restrict__ Sp, const RenderParams P, float* __restrict__ partial, uint32_t* __restrict__ work_counter,
                        uint32_t* __restrict__ spill, DevCounters* __restrict__ dcnt = nullptr) {
  const DevScene& S = *Sp;
  __shared__ uint32_t lds_stack[POOL_LDS_STACK * SOL_WG];
  __shared__ v4u pool4[5 * SOL_WG];
  __shared__ v2u pool2[SOL_WG];
  __shared__ uint8_t oct_table[SOL_OCT_TABLE_BYTES];
  __shared__ uint8_t ray_queue[SOL_WG], hit_queue[SOL_WG];
  __shared__ uint32_t reservoir[SOL_WG / 64][4];
  const uint32_t tid = threadIdx.x;
  const uint32_t gtid = blockIdx.x * SOL_WG + tid;
  const uint32_t lane = tid & 63u;
  const uint32_t wbase = tid & ~63u;  // this wave's first slot / queue entry
  const unsigned long long below = (1ull << lane) - 1ull;
  Stack st;
  st.lds = (lds_u32*)lds_stack + tid;
  st.spill = (SOL_AS1 uint32_t*)spill + gtid;
  st.stride = P.total_threads;
  st.depth = SPILL ? POOL_LDS_STACK : SOL_NO_SPILL;
  sol_search_context<true>(st, S);
  sol_fill_oct_table((lds_u8*)oct_table, tid, SOL_WG);
  st.oct_table = (const lds_u8*)oct_table;
  st.oct_table_on = true;
  if (lane == 0) { reservoir[tid >> 6][0] = 0u; reservoir[tid >> 6][1] = 0u; reservoir[tid >> 6][2] = 0u; }
  __syncthreads();
  lds_u4* const rec4 = (lds_u4*)pool4 + wbase;
  lds_u2* const rec2 = (lds_u2*)pool2 + wbase;
  volatile lds_u8* const rq = (volatile lds_u8*)ray_queue + wbase;
  volatile lds_u8* const hq = (volatile lds_u8*)hit_queue + wbase;
  Counters cnt = {};
  const float inf = __builtin_huge_valf();

  // lane state: `have` - the lane holds a context; `in_flight` - its ray has been handed to the search (running, or over: t.cur == REF_DONE);
  // a context held but not in flight is FRESH: a ray the service block has just made
  bool have = false, in_flight = false, no_work = false;
  Path p = {};
  uint32_t out_at = 0;
  Trav t;
  t.cur = REF_DONE;
  uint32_t n_ray = 0, n_hit = 0;  // wave-uniform: entries of the ray queue (slots holding a ready ray) and of the hit queue

#define POOL_STORE_RAY(slot)                                                                                                            \
  {                                                                                                                                     \
    rec4[0 * SOL_WG + (slot)] = (v4u){__float_as_uint(p.o.x), __float_as_uint(p.o.y), __float_as_uint(p.o.z), __float_as_uint(p.acc_len)}; \
    rec4[1 * SOL_WG + (slot)] = (v4u){__float_as_uint(p.d.x), __float_as_uint(p.d.y), __float_as_uint(p.d.z), p.depth | (p.pdf_seen ? 0x80000000u : 0u)}; \
    rec4[2 * SOL_WG + (slot)] = (v4u){__float_as_uint(p.A.x), __float_as_uint(p.A.y), __float_as_uint(p.A.z), p.rng.k0};           \
    rec4[3 * SOL_WG + (slot)] = (v4u){__float_as_uint(p.C.x), __float_as_uint(p.C.y), __float_as_uint(p.C.z), p.rng.k1};           \
    rec4[4 * SOL_WG + (slot)] = (v4u){p.from, 0u, 0u, 0u}; /* (a parked hit overwrites it: its search is over) */                   \
    rec2[(slot)] = (v2u){p.rng.ctr, out_at};                                                                                       \
  }
#define POOL_LOAD(c, slot, with_hit)                                                                                                    \
  {                                                                                                                                     \
    c.r0 = rec4[0 * SOL_WG + (slot)]; c.r1 = rec4[1 * SOL_WG + (slot)]; c.r2 = rec4[2 * SOL_WG + (slot)]; c.r3 = rec4[3 * SOL_WG + (slot)];  \
    c.r4 = rec4[4 * SOL_WG + (slot)];                                                                                                   \
    c.r5 = rec2[(slot)];                                                                                                                \
  }
#define POOL_ADOPT(c)                                                                                                                   \
  {                                                                                                                                     \
    p.o = mk3(__uint_as_float(c.r0.x), __uint_as_float(c.r0.y), __uint_as_float(c.r0.z)); p.acc_len = __uint_as_float(c.r0.w);          \
    p.d = mk3(__uint_as_float(c.r1.x), __uint_as_float(c.r1.y), __uint_as_float(c.r1.z));                                               \
    p.depth = c.r1.w & 0x7FFFFFFFu; p.pdf_seen = (c.r1.w >> 31) != 0u;                                                                  \
    p.A = mk3(__uint_as_float(c.r2.x), __uint_as_float(c.r2.y), __uint_as_float(c.r2.z)); p.rng.k0 = c.r2.w;                            \
    p.C = mk3(__uint_as_float(c.r3.x), __uint_as_float(c.r3.y), __uint_as_float(c.r3.z)); p.rng.k1 = c.r3.w;                            \
    p.rng.ctr = c.r5.x; out_at = c.r5.y; p.from = c.r4.x; /* (of a ray; a hit's is set when it is shaded) */                          \
  }

  // The pool starts full: every lane makes one context and parks it as a ready ray in its own slot.
  if (pool_fetch_sample(S, P, (volatile lds_u32*)reservoir[tid >> 6], work_counter, lane, below, no_work, p, out_at, cnt)) { have = true; if (COUNT) cnt.samples++; }
  {
    const unsigned long long fm = sol_ballot(have);
    if (have) {
      POOL_STORE_RAY(lane)
      rq[(uint32_t)__popcll(fm & below)] = (uint8_t)lane;
    }
    n_ray = (uint32_t)__popcll(fm);
    have = false;
  }

  for (;;) {
    // ================= service: shade what is finished, make new rays, until the pool holds no parked hit =================
    for (;;) {
      // (a) a finished search this lane holds: shade its vertex; the path goes on with a new ray, or the sample is done
      if (have && in_flight) {  // (a lane enters the service block either searching - and is left alone - or with its search over)
        if (t.cur == REF_DONE) {
          in_flight = false;
          p.o = t.o; p.d = t.d;  // (the ray lives in the search state while it is traced)
          f3 c;
          if (shade_vertex<COUNT, STRICT>(S, p, t.h, c, cnt)) {
            float* a = partial + (size_t)out_at * 3;  // one colour per sample; sol_stage_resolve_kernel adds them up in sample order
            a[0] = c.x; a[1] = c.y; a[2] = c.z;
            have = false;
          }
        }
      }
      // (b) a lane without a context takes the next sample of the work order
      if (!have && !no_work) {
        have = pool_fetch_sample(S, P, (volatile lds_u32*)reservoir[tid >> 6], work_counter, lane, below, no_work, p, out_at, cnt);
        if (COUNT && have) cnt.samples++;
      }
      // (c) parked hits: the lanes that are not searching (a fresh ray, or nothing) take one each - the fresh ray goes into the slot the
      // hit came from - and go round again to shade it
      if (n_hit == 0u) break;
      const bool elig = !in_flight;
      const unsigned long long em = sol_ballot(elig);
      const uint32_t m = min((uint32_t)__popcll(em), n_hit);
      if (m == 0u) break;
      // a pass that would shade a handful of hits costs what a full one costs: the hits wait in the pool while the wave has other work
      // (a search running, a fresh ray to start); with nothing else to do they are shaded whatever their number
      if (m < P.swap_min && sol_ballot(have) != 0ull) break;
      const uint32_t r = (uint32_t)__popcll(em & below);
      const bool take = elig && r < m;
      const bool put = take && have;
      const unsigned long long wm = sol_ballot(put);
      if (take) {
        const uint32_t slot = hq[n_hit - 1u - r];
        PoolCtx c;
        POOL_LOAD(c, slot, true)
        if (put) {
          POOL_STORE_RAY(slot)
          rq[n_ray + (uint32_t)__popcll(wm & below)] = (uint8_t)slot;
        }
        POOL_ADOPT(c)
        t.o = p.o; t.d = p.d;
        t.h.t = __uint_as_float(c.r4.x); t.h.ref = c.r4.y; t.h.dfs = SOL_DFS_UNKNOWN; t.h.u = __uint_as_float(c.r4.z); t.h.v = __uint_as_float(c.r4.w);
        t.cur = REF_DONE;
        have = true;
        in_flight = true;
      }
      n_hit -= m;
      n_ray += (uint32_t)__popcll(wm);
    }
    // fresh rays start their search: world.hit(ray, RAY_INTERVAL) (src/renderer/mod.rs:165)
    if (have && !in_flight) {
      trav_begin<true>(t, p.o, p.d, RAY_MIN_F, inf, S.wroot, S.rxmin, S.rxmax, S.rymin, S.rymax, S.rzmin, S.rzmax, 0);
      in_flight = true;
    }
    if (sol_ballot(have || !no_work) == 0ull && n_ray == 0u) break;  // no context left in the wave and none to fetch (no parked hit either: the service block drained them)

    // ================= search: one step per turn; a lane whose search is over exchanges it for a ready ray of the pool =================
#if SOL_LOOP_PRIO
    __builtin_amdgcn_s_setprio(SOL_LOOP_PRIO);
#endif
    for (;;) {
      const bool act = in_flight && t.cur != REF_DONE;
      const unsigned long long am = sol_ballot(act);
      const uint32_t n_act = (uint32_t)__popcll(am);
      // lanes the wave could still feed: holding a context, or able to fetch one
      const uint32_t n_live = (uint32_t)__popcll(sol_ballot(have || !no_work));
      const bool few = n_act * 64u < P.switch_below * n_live || n_act == 0u;
      if (n_ray != 0u) {
        const unsigned long long im = ~am;  // idle: search over (a hit to park), or no context
        const uint32_t n_idle = 64u - n_act;
        if ((n_idle >= P.swap_min && n_idle != 0u) || few) {
          const uint32_t m = min(n_idle, n_ray);
          const uint32_t r = (uint32_t)__popcll(im & below);
          const bool take = !act && r < m;
          const bool put = take && have;
          const unsigned long long wm = sol_ballot(put);
          if (take) {
            phase_tick<COUNT>(cnt, 2);  // (instrumented builds: the "generate" phase counts the swap passes of the search loop and their lanes)
            const uint32_t slot = rq[n_ray - 1u - r];
            PoolCtx c;
            POOL_LOAD(c, slot, false)
            if (put) {  // park the finished context: path state, ray and closest hit
              p.o = t.o; p.d = t.d;
              POOL_STORE_RAY(slot)
              rec4[4 * SOL_WG + slot] = (v4u){__float_as_uint(t.h.t), t.h.ref, __float_as_uint(t.h.u), __float_as_uint(t.h.v)};
              hq[n_hit + (uint32_t)__popcll(wm & below)] = (uint8_t)slot;
            }
            POOL_ADOPT(c)
            trav_begin<true>(t, p.o, p.d, RAY_MIN_F, inf, S.wroot, S.rxmin, S.rxmax, S.rymin, S.rymax, S.rzmin, S.rzmax, 0);
            have = true;
            in_flight = true;
          }
          n_ray -= m;
          n_hit += (uint32_t)__popcll(wm);
          continue;
        }
      }
      if (few) {
        // leave for the service block if it has something to do: a finished search, a parked hit, a lane that can fetch
        const bool fin = have && in_flight && t.cur == REF_DONE;
        // (parked hits count only if the service block will take them: enough of them for a pass, or nothing else to do - its own rule)
        if ((n_hit != 0u && (min(64u - n_act, n_hit) >= P.swap_min || n_act == 0u)) || sol_ballot(fin || (!have && !no_work)) != 0ull) break;
        if (n_act == 0u) break;  // (nothing runs and nothing to service: the wave is done - the outer loop's test ends it)
      }
      trav_step_wave<COUNT, MEDIUM, STRICT, true>(S, t, act, st, p.rng, p.depth, cnt);
      // (rule 8, sol_path.h) a finished search whose closest hit is the flat primitive the ray left: searched again behind that hit
      if (act && t.cur == REF_DONE && sol_self_hit(p.from, t.h)) {
        const float bt = t.bt;
        const uint32_t bdfs = t.bdfs;
        trav_begin<true>(t, t.o, t.d, sol_behind(t.h.t), inf, S.wroot, S.rxmin, S.rxmax, S.rymin, S.rymax, S.rzmin, S.rzmax, 0);
        t.bt = bt; t.bdfs = bdfs;
      }
      if (COUNT && act && t.cur == REF_DONE) cnt.rays++;
      // (STRICT) a finished search whose closest hit is a triangle the consistency rule refuses searches again, behind that hit
      if (STRICT && act && t.cur == REF_DONE) trav_accept_or_restart(S, t, st);
    }
#if SOL_LOOP_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  }
  if (COUNT) flush_counters(cnt, dcnt);
}

template <bool MEDIUM, bool SPILL, bool STRICT>
static hipError_t launch_pool4(const DevScene* dS, const RenderParams& P, float* partial, uint32_t* work, uint32_t* spill, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL((sol_render_pool4_kernel<MEDIUM, SPILL, STRICT>), dim3(grid), dim3(SOL_WG), 0, stream, dS, P, partial, work, spill, (DevCounters*)nullptr);
  return hipGetLastError();
}
hipError_t sol_launch_pool4(const DevScene& S, const DevScene* dS, const RenderParams& P, float* partial, uint32_t* work, uint32_t* spill, uint32_t grid,
                            bool medium, bool may_spill, DevCounters* cnt, hipStream_t stream) {
  const bool strict = S.tri_delta > 0.0f;
  if (cnt) {  // instrumented (SOL_POOL_COUNT=1: phase statistics of the pool kernel; two variants only)
    if (strict) return hipErrorInvalidValue;
    if (medium) hipLaunchKernelGGL((sol_render_pool4_kernel<true, true, false, true>), dim3(grid), dim3(SOL_WG), 0, stream, dS, P, partial, work, spill, cnt);
    else hipLaunchKernelGGL((sol_render_pool4_kernel<false, true, false, true>), dim3(grid), dim3(SOL_WG), 0, stream, dS, P, partial, work, spill, cnt);
    return hipGetLastError();
  }
#define POOL4(M, SP, ST) launch_pool4<M, SP, ST>(dS, P, partial, work, spill, grid, stream)
  if (strict) {
    if (may_spill) return medium ? POOL4(true, true, true) : POOL4(false, true, true);
    return medium ? POOL4(true, false, true) : POOL4(false, false, true);
  }
  if (may_spill) return medium ? POOL4(true, true, false) : POOL4(false, true, false);
  return medium ? POOL4(true, false, false) : POOL4(false, false, false);
#undef POOL4
}
template <typename K>
static int pool4_blocks(K kernel) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, SOL_WG, 0) != hipSuccess || n < 1) n = 1;
  return n;
}
int sol_pool4_blocks_per_cu(bool medium, bool strict) {  // (the SPILL = false builds need no more registers or LDS than these)
  if (strict) return medium ? pool4_blocks(sol_render_pool4_kernel<true, true, true>) : pool4_blocks(sol_render_pool4_kernel<false, true, true>);
  return medium ? pool4_blocks(sol_render_pool4_kernel<true, true, false>) : pool4_blocks(sol_render_pool4_kernel<false, true, false>);
}
int sol_pool4_lds_stack_depth() { return POOL_LDS_STACK; }
#endif  // SOL_AB_KERNELS
