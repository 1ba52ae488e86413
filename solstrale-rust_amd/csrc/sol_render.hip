// sol_render.hip -- the gfx950 render kernels of the path-tracing hot path (wave64, no MFMA: branchy traversal + fp32
// shading). Replaces src/renderer/mod.rs:241-291 (row tasks) and everything under ray_color (:164-206).
//
// Two persistent kernels with bit-identical results (same device functions, sol_path.h):
//
//  sol_render_kernel (the product path, SOL_KERNEL=v1) -- one path per lane, state in registers. A wave alternates between
//    the search loop (one resumable BVH step per turn, per-lane stack in LDS) and the service block (shade the closest hit,
//    fetch work, generate the next camera ray); it leaves the search loop as soon as too few of its lanes are still
//    searching while others wait (RenderParams::switch_below), unfinished searches keep their state across the service
//    block. DESIGN.md 3.
//  sol_render_pool_kernel (SOL_KERNEL=v2, A/B only) -- wave-private WAVEFRONT over a pool of path slots in global memory
//    (112 B of state per slot): stage A shades / regenerates 64 slots at a time and compacts live rays into an LDS queue by
//    ballot + popcount prefix; stage B searches with refill of idle lanes from the queue. Higher search occupancy, but the
//    state traffic and the lower residency (3 waves/SIMD) cost more than they buy (measurements in sol_api.cpp).
//  (The two-kernel wavefront, SOL_KERNEL=v3, lives in sol_wavefront.hip.)
//
// In all of them a work item's 16 samples are summed in sample order by whoever owns the item and written once; no float atomic
// touches the accumulator, so the image is a pure function of (scene, seed), bit-identical for any tile partition.
#include <hip/hip_runtime.h>

#include "sol_launch.h"
#include "sol_path.h"

// ---------------------------------------------------------------------------------------------------------------------------
// v1: one path per lane
// ---------------------------------------------------------------------------------------------------------------------------
// SPILL = false: built for scenes whose searches fit the LDS stack (sol_api.cpp picks it by the tree's depth): every stack access
// is a plain LDS access, the spill branches and their waits fold away.
template <bool COUNT, bool MEDIUM, bool SPILL>
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
#ifndef SOL_NO_OCT_TABLE
  __shared__ uint8_t oct_table[SOL_OCT_TABLE_BYTES];
  sol_fill_oct_table((lds_u8*)oct_table, tid, SOL_WG);
  st.oct_table = (const lds_u8*)oct_table;
  st.oct_table_on = true;
  __syncthreads();
#endif
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
  if (lane == 0) { reservoir[tid >> 6][0] = 0u; reservoir[tid >> 6][1] = 0u; }
#if SOL_PARK_PATH
  __shared__ float park[6][SOL_WG];
#endif
#if SOL_DONATE
  __shared__ uint32_t donate_pairs[SOL_WG / 64][64];  // per wave: the lanes that give a node group in a hand-out (A/B build)
  uint32_t turn = 0;
#endif
#if SOL_COOP_TRIANGLES
  __shared__ uint32_t coop_queue[SOL_WG / 64][64];  // per wave: the pending triangle tests of a cooperative primitive part (A/B build)
#else
  lds_u32* const coop_queue[SOL_WG / 64] = {};
  (void)coop_queue;
#endif

  for (;;) {
    // ---- lanes whose search is over: shade the vertex, then start the next ray of the path / sample / item ----
    if (t.cur == REF_DONE) {
      if (in_flight) {
        in_flight = false;
        if (COUNT) cnt.rays++;
        p.o = t.o; p.d = t.d;  // (the ray lives in the search state while it is traced)
#if SOL_PARK_PATH
        p.A = mk3(park[0][tid], park[1][tid], park[2][tid]);
        p.C = mk3(park[3][tid], park[4][tid], park[5][tid]);
#endif
        f3 c;
        if (COUNT && p.depth == 0u && SOL_REF_KIND(t.h.ref) != SOL_REF_NONE) cnt.primary_hits++;
        if (shade_vertex<COUNT>(S, p, t.h, c, cnt)) {
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
      if (!have_item) {
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
      if (!alive) {
        phase_tick<COUNT>(cnt, 2);
        generate_path<COUNT>(S, P.seed_lo, P.seed_hi, it.px, it.py, s, p, cnt);
        alive = true;
      }
#if SOL_PARK_PATH
      // The path's throughput (A, C: six registers) is dead during the search: parked in LDS until the next shading, so that the
      // search loop's live set stays below the point where edits of this block perturb its register allocation
      park[0][tid] = p.A.x; park[1][tid] = p.A.y; park[2][tid] = p.A.z; park[3][tid] = p.C.x; park[4][tid] = p.C.y; park[5][tid] = p.C.z;
#endif
      // world.hit(ray, RAY_INTERVAL) (src/renderer/mod.rs:165)
      trav_begin<!SOL_WORLD_BINARY>(t, p.o, p.d, RAY_MIN_F, inf, SOL_WORLD_ROOT(S), S.rxmin, S.rxmax, S.rymin, S.rymax, S.rzmin, S.rzmax, 0);
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
#if SOL_WAVE_STEP && !SOL_WORLD_BINARY
      trav_step_wave<COUNT, MEDIUM>(S, t, act, st, (volatile lds_u32*)coop_queue[tid >> 6], p.rng, p.depth, cnt);
#if SOL_DONATE
      if (!COUNT && !MEDIUM && !SPILL && P.donate) trav_donate(t, st, (volatile lds_u32*)donate_pairs[tid >> 6], turn++);
#endif
#else
      if (act) trav_step<COUNT, MEDIUM, SOL_WORLD_BINARY>(S, t, st, p.rng, p.depth, cnt);
#endif
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
  for (uint32_t k = 0; k < P.fine_count; ++k) sum = sum + mk3(a[3 * k], a[3 * k + 1], a[3 * k + 2]);
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

#ifdef SOL_AB_KERNELS
// ---------------------------------------------------------------------------------------------------------------------------
// v2: wave-private wavefront over a pool of path slots
// ---------------------------------------------------------------------------------------------------------------------------
// Pool record k of slot s of wave w: pool[(w * POOL_RECORDS + k) * slots + s]  (float4; consecutive slots are contiguous)
//   0: o.xyz, acc_len        1: d.xyz, flags | depth << 8      2: A.xyz, rng.k0       3: C.xyz, rng.k1
//   4: sum.xyz, rng.ctr      5: px | py << 16, out slot, chunk, sample               6: hit t, ref, u, v
#define POOL_RECORDS 7
#define PF_ITEM 1u   // the slot holds a work item
#define PF_ALIVE 2u  // its path is in flight: a ray is queued, or its hit waits to be shaded
#define PF_PDF 4u    // Path.pdf_seen

#ifndef SOL_REFILL_MIN
#define SOL_REFILL_MIN 12  // refill idle lanes once at least this many are idle (or all are)
#endif
#ifndef SOL_TRAV_BURST
#define SOL_TRAV_BURST 6   // traversal steps between two refill checks
#endif

#ifndef SOL_V2_MIN_WAVES
#define SOL_V2_MIN_WAVES 1
#endif
template <bool COUNT, bool MEDIUM>
__global__ void __launch_bounds__(SOL_WG, SOL_V2_MIN_WAVES)
sol_render_pool_kernel(const DevScene S, const RenderParams P, float* __restrict__ acc, float* __restrict__ partial,
                       uint32_t* __restrict__ work_counter, uint32_t* __restrict__ spill, float4* __restrict__ pool,
                       DevCounters* __restrict__ dcnt) {
  __shared__ uint32_t lds_stack[SOL_LDS_STACK * SOL_WG];
  __shared__ uint16_t lds_queue[(SOL_WG / 64) * SOL_POOL_MAX];
  const uint32_t tid = threadIdx.x;
  const uint32_t gtid = blockIdx.x * SOL_WG + tid;
  const uint32_t lane = tid & 63u;
  // wave-uniform by construction; readfirstlane tells the compiler, so pool / queue bases live in SGPRs
  const uint32_t wave_in_wg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint32_t wave = blockIdx.x * (SOL_WG / 64) + wave_in_wg;
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  Stack st;
  st.lds = (lds_u32*)lds_stack + tid;
  st.spill = (SOL_AS1 uint32_t*)spill + gtid;
  st.stride = P.total_threads;
  st.depth = SOL_LDS_STACK;
  sol_search_context<false>(st, S);
  uint16_t* queue = lds_queue + wave_in_wg * SOL_POOL_MAX;
  const uint32_t NS = P.pool_slots;  // slots of this wave, a multiple of 64
  float4* const rec = pool + (size_t)wave * POOL_RECORDS * NS;
  Counters cnt = {};
  const float inf = __builtin_huge_valf();

  for (uint32_t sl = lane; sl < NS; sl += 64) rec[1 * NS + sl] = make_float4(0.f, 0.f, 0.f, 0.f);  // all slots empty
  bool exhausted = false;  // wave-uniform: the global item counter has run out

  for (;;) {
    // ======== stage A: shade finished searches, regenerate, compact live rays into the queue ========
    uint32_t qn = 0;  // wave-uniform
    for (uint32_t base = 0; base < NS; base += 64) {
      const uint32_t sl = base + lane;
      float4 r1 = rec[1 * NS + sl];
      uint32_t flags = __float_as_uint(r1.w);
      // wave-uniform shortcut: nothing in these 64 slots and nothing left to fetch
      if (exhausted && sol_ballot((flags & PF_ITEM) != 0) == 0ull) continue;
      Path p = {};
      Item it = {0, 0, 0, 0};
      uint32_t s = 0;
      f3 sum = mk3(0.f, 0.f, 0.f);
      if (flags & PF_ITEM) {
        const float4 r0 = rec[0 * NS + sl], r2 = rec[2 * NS + sl], r3 = rec[3 * NS + sl], r4 = rec[4 * NS + sl];
        const float4 r5 = rec[5 * NS + sl];
        p.o = mk3(r0.x, r0.y, r0.z); p.acc_len = r0.w;
        p.d = mk3(r1.x, r1.y, r1.z); p.depth = flags >> 8; p.pdf_seen = (flags & PF_PDF) != 0;
        p.A = mk3(r2.x, r2.y, r2.z); p.rng.k0 = __float_as_uint(r2.w);
        p.C = mk3(r3.x, r3.y, r3.z); p.rng.k1 = __float_as_uint(r3.w);
        sum = mk3(r4.x, r4.y, r4.z); p.rng.ctr = __float_as_uint(r4.w);
        const uint32_t pix = __float_as_uint(r5.x);
        it.px = pix & 0xFFFFu; it.py = pix >> 16; it.slot = __float_as_uint(r5.y); it.chunk = __float_as_uint(r5.z);
        s = __float_as_uint(r5.w);
      }
      bool has_item = (flags & PF_ITEM) != 0, alive = (flags & PF_ALIVE) != 0;
      if (has_item && alive) {  // the search of this path's ray has finished: shade it
        const float4 r6 = rec[6 * NS + sl];
        Hit h;
        h.t = r6.x; h.ref = __float_as_uint(r6.y); h.dfs = 0; h.u = r6.z; h.v = r6.w;
        f3 c;
        if (shade_vertex<COUNT>(S, p, h, c, cnt)) {
          sum = sum + c;  // add_row_data (src/renderer/mod.rs:361-365): sums, not means, in sample order
          alive = false;
          s++;
          const uint32_t s_end = min(P.first_sample + (it.chunk + 1u) * SOL_CHUNK, P.first_sample + P.n_samples);
          if (s == s_end) {
            write_chunk(P, acc, partial, it.slot, it.chunk, sum);
            has_item = false;
          }
        }
      }
      bool refused = false;
      if (!has_item && !exhausted) {  // take the next work item: one atomic for the wave, popcount prefix per lane
        const unsigned long long need = sol_ballot(true);
        const uint32_t leader = (uint32_t)__ffsll((long long)need) - 1u;
        uint32_t b0 = 0;
        if (lane == leader) b0 = atomicAdd(work_counter, (uint32_t)__popcll(need));
        b0 = __shfl(b0, (int)leader);
        const uint32_t item = b0 + (uint32_t)__popcll(need & lanes_below);
        refused = item >= P.n_items;
        if (!refused && decode_item(S, P, item, it)) {
          s = P.first_sample + it.chunk * SOL_CHUNK;
          sum = mk3(0.f, 0.f, 0.f);
          has_item = true;
          alive = false;
        }
      }
      // the counter is monotone: once any lane was refused, every later fetch of this wave would be refused too
      if (sol_ballot(refused) != 0ull) exhausted = true;
      if (has_item && !alive) {
        phase_tick<COUNT>(cnt, 2);
        generate_path<COUNT>(S, P.seed_lo, P.seed_hi, it.px, it.py, s, p, cnt);
        alive = true;
      }
      // store the slot
      const uint32_t nflags = (has_item ? PF_ITEM : 0u) | (alive ? PF_ALIVE : 0u) | (p.pdf_seen ? PF_PDF : 0u) | (p.depth << 8);
      if (has_item) {
        rec[0 * NS + sl] = make_float4(p.o.x, p.o.y, p.o.z, p.acc_len);
        rec[1 * NS + sl] = make_float4(p.d.x, p.d.y, p.d.z, __uint_as_float(nflags));
        rec[2 * NS + sl] = make_float4(p.A.x, p.A.y, p.A.z, __uint_as_float(p.rng.k0));
        rec[3 * NS + sl] = make_float4(p.C.x, p.C.y, p.C.z, __uint_as_float(p.rng.k1));
        rec[4 * NS + sl] = make_float4(sum.x, sum.y, sum.z, __uint_as_float(p.rng.ctr));
        rec[5 * NS + sl] = make_float4(__uint_as_float(it.px | (it.py << 16)), __uint_as_float(it.slot), __uint_as_float(it.chunk),
                                       __uint_as_float(s));
      } else if (flags & PF_ITEM) {
        rec[1 * NS + sl] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      // compaction: live rays of these 64 slots go to the queue in slot order
      const unsigned long long live = sol_ballot(has_item);
      if (has_item) queue[qn + (uint32_t)__popcll(live & lanes_below)] = (uint16_t)sl;
      qn += (uint32_t)__popcll(live);
    }
    if (qn == 0) break;  // no ray in flight and no work left

    // ======== stage B: intersect the queued rays; idle lanes are refilled from the queue ========
    uint32_t head = 0;  // wave-uniform
    bool have = false;
    uint32_t my_slot = 0;
    Trav t;
    t.cur = REF_DONE;
    Rng rng_medium = {0, 0, 0};
    uint32_t depth_medium = 0;
    for (;;) {
      const unsigned long long idle = sol_ballot(!have);
      const uint32_t n_idle = (uint32_t)__popcll(idle);
      if (head < qn && (n_idle >= SOL_REFILL_MIN || n_idle == 64u)) {
        if (!have) {
          const uint32_t q = head + (uint32_t)__popcll(idle & lanes_below);
          if (q < qn) {
            my_slot = queue[q];
            const float4 r0 = rec[0 * NS + my_slot], r1 = rec[1 * NS + my_slot];
            if (MEDIUM) {  // the medium's sub-stream needs the path's generator and depth
              const float4 r2 = rec[2 * NS + my_slot], r3 = rec[3 * NS + my_slot];
              rng_medium.k0 = __float_as_uint(r2.w); rng_medium.k1 = __float_as_uint(r3.w);
              depth_medium = __float_as_uint(r1.w) >> 8;
            }
            trav_begin<!SOL_WORLD_BINARY>(t, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), RAY_MIN_F, inf, SOL_WORLD_ROOT(S), S.rxmin, S.rxmax, S.rymin,
                       S.rymax, S.rzmin, S.rzmax, 0);
            have = true;
          }
        }
        head = min(qn, head + n_idle);
      }
      if (sol_ballot(have) == 0ull) break;  // every queued ray has been searched
      if (have) {
        for (int k = 0; k < SOL_TRAV_BURST && t.cur != REF_DONE; ++k) trav_step<COUNT, MEDIUM, SOL_WORLD_BINARY>(S, t, st, rng_medium, depth_medium, cnt);
        if (t.cur == REF_DONE) {
          rec[6 * NS + my_slot] = make_float4(t.h.t, __uint_as_float(t.h.ref), t.h.u, t.h.v);
          if (COUNT) cnt.rays++;
          have = false;
        }
      }
    }
  }
  if (COUNT) flush_counters(cnt, dcnt);
}

#endif  // SOL_AB_KERNELS

// Diagnostic: one path (pixel, sample) on one lane, every ray and its closest hit recorded: 12 floats per ray
// (origin, direction, t, ref bits, dfs bits, depth, 0, 0), then a terminator row (colour in the first 3 floats, -1 in the 4th).
template <bool MEDIUM>
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
    closest_hit<false, MEDIUM, SOL_WORLD_BINARY>(S, p.o, p.d, RAY_MIN_F, inf, SOL_WORLD_ROOT(S), S.rxmin, S.rxmax, S.rymin, S.rymax,
                                                 S.rzmin, S.rzmax, h, st, 0, p.rng, p.depth, cnt);
    if (row + 1 < max_rows) {
      float* o = out + (size_t)row * 12;
      o[0] = p.o.x; o[1] = p.o.y; o[2] = p.o.z; o[3] = p.d.x; o[4] = p.d.y; o[5] = p.d.z; o[6] = h.t;
      o[7] = __uint_as_float(h.ref); o[8] = __uint_as_float(h.dfs); o[9] = (float)p.depth; o[10] = 0.f; o[11] = 0.f;
      row++;
    }
    f3 c;
    if (shade_vertex<false>(S, p, h, c, cnt)) {
      float* o = out + (size_t)row * 12;
      o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = -1.0f;
      return;
    }
  }
}
hipError_t sol_launch_debug_path(const DevScene& S, const RenderParams& P, uint32_t px, uint32_t py, uint32_t s, uint32_t* spill,
                                 float* out, uint32_t max_rows, bool medium, hipStream_t stream) {
  if (medium) hipLaunchKernelGGL((sol_debug_path_kernel<true>), dim3(1), dim3(SOL_WG), 0, stream, S, P, px, py, s, spill, out, max_rows);
  else hipLaunchKernelGGL((sol_debug_path_kernel<false>), dim3(1), dim3(SOL_WG), 0, stream, S, P, px, py, s, spill, out, max_rows);
  return hipGetLastError();
}

// ---- launch wrappers (called from sol_api.cpp) ----
template <bool COUNT, bool MEDIUM, bool SPILL>
static hipError_t launch_v1(const DevScene* dS, const RenderParams& P, float* acc, float* partial, uint32_t* work,
                            uint32_t* spill, DevCounters* cnt, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL((sol_render_kernel<COUNT, MEDIUM, SPILL>), dim3(grid), dim3(SOL_WG), 0, stream, dS, P, acc, partial, work, spill, cnt);
  return hipGetLastError();
}
#ifdef SOL_AB_KERNELS
template <bool COUNT, bool MEDIUM>
static hipError_t launch_v2(const DevScene& S, const RenderParams& P, float* acc, float* partial, uint32_t* work,
                            uint32_t* spill, float4* pool, DevCounters* cnt, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL((sol_render_pool_kernel<COUNT, MEDIUM>), dim3(grid), dim3(SOL_WG), 0, stream, S, P, acc, partial, work, spill,
                     pool, cnt);
  return hipGetLastError();
}
#endif

hipError_t sol_launch_render(int version, const DevScene& S, const DevScene* dS, const RenderParams& P, float* acc, float* partial, uint32_t* work,
                             uint32_t* spill, void* pool, DevCounters* cnt, uint32_t grid, bool count, bool medium, bool may_spill,
                             hipStream_t stream) {
  if (version == 1) {
    if (count) return medium ? launch_v1<true, true, true>(dS, P, acc, partial, work, spill, cnt, grid, stream)
                             : launch_v1<true, false, true>(dS, P, acc, partial, work, spill, cnt, grid, stream);
    if (may_spill) return medium ? launch_v1<false, true, true>(dS, P, acc, partial, work, spill, cnt, grid, stream)
                                 : launch_v1<false, false, true>(dS, P, acc, partial, work, spill, cnt, grid, stream);
    return medium ? launch_v1<false, true, false>(dS, P, acc, partial, work, spill, cnt, grid, stream)
                  : launch_v1<false, false, false>(dS, P, acc, partial, work, spill, cnt, grid, stream);
  }
#ifdef SOL_AB_KERNELS
  float4* pl = (float4*)pool;
  if (count) return medium ? launch_v2<true, true>(S, P, acc, partial, work, spill, pl, cnt, grid, stream)
                           : launch_v2<true, false>(S, P, acc, partial, work, spill, pl, cnt, grid, stream);
  return medium ? launch_v2<false, true>(S, P, acc, partial, work, spill, pl, cnt, grid, stream)
                : launch_v2<false, false>(S, P, acc, partial, work, spill, pl, cnt, grid, stream);
#else
  return hipErrorInvalidValue;  // (the wavefront variants are A/B builds: -DSOL_AB_KERNELS)
#endif
}

template <typename K>
static int blocks_per_cu(K kernel) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, SOL_WG, 0) != hipSuccess || n < 1) n = 1;
  return n;
}
int sol_render_blocks_per_cu(int version, bool count, bool medium) {
  if (version == 1) {  // (the SPILL = false builds need no more registers or LDS than these)
    if (count) return medium ? blocks_per_cu(sol_render_kernel<true, true, true>) : blocks_per_cu(sol_render_kernel<true, false, true>);
    return medium ? blocks_per_cu(sol_render_kernel<false, true, true>) : blocks_per_cu(sol_render_kernel<false, false, true>);
  }
#ifdef SOL_AB_KERNELS
  if (count) return medium ? blocks_per_cu(sol_render_pool_kernel<true, true>) : blocks_per_cu(sol_render_pool_kernel<true, false>);
  return medium ? blocks_per_cu(sol_render_pool_kernel<false, true>) : blocks_per_cu(sol_render_pool_kernel<false, false>);
#else
  return 1;
#endif
}
#ifdef SOL_AB_KERNELS
size_t sol_pool_bytes_per_wave(uint32_t slots) { return (size_t)POOL_RECORDS * slots * sizeof(float4); }
#endif
