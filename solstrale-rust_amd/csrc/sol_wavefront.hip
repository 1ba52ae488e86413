// sol_wavefront.hip -- the two WAVEFRONT variants of the render kernel (generate / intersect / shade stages over queues of
// active rays compacted by wave64 ballot + popcount prefix), bit-identical to sol_render_kernel (same device functions). Kept as
// the measured alternatives to the product kernel, in the -DSOL_AB_KERNELS build of the library only (`_build_ab/`;
// SOL_KERNEL=v2|v3): they reach a higher lane occupancy in the search (0.67 - 0.73) but pay 112 B of state traffic per vertex and
// run at 3 waves per SIMD (C3, 64 spp, round 3: product 2049, v2 1308, v3 935 Msamples/s).
//
// v2, sol_render_pool_kernel -- wave-private wavefront over a pool of path slots in global memory: stage A shades / regenerates
//   64 slots at a time and compacts live rays into an LDS queue by ballot + popcount prefix; stage B searches with refill of idle
//   lanes from the queue.
// v3 -- two kernels per round over ONE global pool of path slots in HBM (112 B of state per slot). Per round the host launches
//   sol_wf_shade_kernel : one thread per slot. Shades the finished search of the slot's path (scatter, light/BSDF mixture
//                         pdf, throughput update or termination), starts the next sample of its work item or takes a new
//                         (pixel, 16-sample chunk) item from the wave's item reservoir (refilled 64 items per global atomic).
//   sol_wf_trace_kernel : persistent waves. Wave w owns the 64-slot groups w, w + W, w + 2W, .. of the pool (static round
//                         robin, no atomics); it reads a group's flags and COMPACTS the live slots onto its idle lanes with
//                         ballot + popcount + n-th-set-bit selection - no ray queue exists in memory. Lanes search the BVH
//                         (trav_step, LDS stack + global spill); whenever enough lanes have finished they are refilled the
//                         same way. Only (origin, direction, best hit) live in registers, so the kernel is compiled for 5
//                         waves per SIMD (SOL_WF_MIN_WAVES) against the product kernel's 4.
// The first version of this design issued one global atomic per 16 rays on one address and ran at the chip's single-address
// atomic rate (about 88 M/s) instead of the traversal rate - the lesson behind the product kernel's item reservoir.
#ifdef SOL_AB_KERNELS  // an A/B variant: not part of the product library (build.py)
#include <hip/hip_runtime.h>

#include "sol_launch.h"
#include "sol_path.h"

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
#define WF_STRETCH 1024u   // slots a trace wave takes per global atomic
#define WF_RESERVOIR 64u   // work items a shade wave takes per global atomic

// ---------------------------------------------------------------------------------------------------------------------------
// v2: wave-private wavefront over a pool of path slots
// ---------------------------------------------------------------------------------------------------------------------------
// Pool record k of slot s of wave w: pool[(w * POOL_RECORDS + k) * slots + s]  (float4; consecutive slots are contiguous)
//   0: o.xyz, acc_len        1: d.xyz, flags | depth << 8      2: A.xyz, rng.k0       3: C.xyz, rng.k1
//   4: sum.xyz, rng.ctr      5: px | py << 16, out slot, chunk, sample               6: hit t, ref, u, v

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
        h.t = r6.x; h.ref = __float_as_uint(r6.y); h.dfs = SOL_DFS_UNKNOWN; h.u = r6.z; h.v = r6.w;
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
        rec[6 * NS + sl] = make_float4(0.f, __uint_as_float(p.from), 0.f, 0.f);  // (rule 8: Path::from rides in the hit record until the ray is traced)
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
    uint32_t my_slot = 0, my_from = 0;
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
            trav_begin<true>(t, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), RAY_MIN_F, inf, S.wroot, S.rxmin, S.rxmax, S.rymin,
                       S.rymax, S.rzmin, S.rzmax, 0);
            my_from = __float_as_uint(rec[6 * NS + my_slot].y);
            have = true;
          }
        }
        head = min(qn, head + n_idle);
      }
      if (sol_ballot(have) == 0ull) break;  // every queued ray has been searched
      if (have) {
        for (int k = 0; k < SOL_TRAV_BURST && t.cur != REF_DONE; ++k) trav_step<COUNT, MEDIUM>(S, t, st, rng_medium, depth_medium, cnt);
        if (t.cur == REF_DONE && sol_self_hit(my_from, t.h)) {  // (rule 8, sol_path.h: searched again behind the primitive the ray left)
          trav_begin<true>(t, t.o, t.d, sol_behind(t.h.t), inf, S.wroot, S.rxmin, S.rxmax, S.rymin, S.rymax, S.rzmin, S.rzmax, 0);
        } else if (t.cur == REF_DONE) {
          rec[6 * NS + my_slot] = make_float4(t.h.t, __uint_as_float(t.h.ref), t.h.u, t.h.v);
          if (COUNT) cnt.rays++;
          have = false;
        }
      }
    }
  }
  if (COUNT) flush_counters(cnt, dcnt);
}


template <bool COUNT, bool MEDIUM>
static hipError_t launch_v2(const DevScene& S, const RenderParams& P, float* acc, float* partial, uint32_t* work,
                            uint32_t* spill, float4* pool, DevCounters* cnt, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL((sol_render_pool_kernel<COUNT, MEDIUM>), dim3(grid), dim3(SOL_WG), 0, stream, S, P, acc, partial, work, spill,
                     pool, cnt);
  return hipGetLastError();
}
hipError_t sol_launch_pool(const DevScene& S, const RenderParams& P, float* acc, float* partial, uint32_t* work, uint32_t* spill, void* pool,
                           DevCounters* cnt, uint32_t grid, bool count, bool medium, hipStream_t stream) {
  float4* pl = (float4*)pool;
  if (count) return medium ? launch_v2<true, true>(S, P, acc, partial, work, spill, pl, cnt, grid, stream)
                           : launch_v2<true, false>(S, P, acc, partial, work, spill, pl, cnt, grid, stream);
  return medium ? launch_v2<false, true>(S, P, acc, partial, work, spill, pl, cnt, grid, stream)
                : launch_v2<false, false>(S, P, acc, partial, work, spill, pl, cnt, grid, stream);
}
int sol_pool_blocks_per_cu(bool count, bool medium) {
  int n = 0;
  hipError_t e;
  if (count) e = medium ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sol_render_pool_kernel<true, true>, SOL_WG, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sol_render_pool_kernel<true, false>, SOL_WG, 0);
  else e = medium ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sol_render_pool_kernel<false, true>, SOL_WG, 0)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sol_render_pool_kernel<false, false>, SOL_WG, 0);
  if (e != hipSuccess || n < 1) n = 1;
  return n;
}
size_t sol_pool_bytes_per_wave(uint32_t slots) { return (size_t)POOL_RECORDS * slots * sizeof(float4); }

// ---------------------------------------------------------------------------------------------------------------------------
// v3: two-kernel wavefront
// ---------------------------------------------------------------------------------------------------------------------------
struct WfCounters {
  uint32_t work_next;    // next work item (monotone, saturating)
  uint32_t slot_cursor;  // next pool stretch for the trace kernel of this round
  uint32_t live;         // set by the shade kernel of this round when any slot holds work
  uint32_t pad;
};

template <bool COUNT>
__global__ void __launch_bounds__(SOL_WG)
sol_wf_shade_kernel(const DevScene S, const RenderParams P, float* __restrict__ acc, float* __restrict__ partial,
                    WfCounters* __restrict__ ctr, float4* __restrict__ rec, uint2* __restrict__ reservoir,
                    DevCounters* __restrict__ dcnt) {
  const size_t NS = P.pool_slots;  // total slots, a multiple of SOL_WG
  const uint32_t sl = blockIdx.x * SOL_WG + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(sl >> 6));
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  Counters cnt = {};
  const float4 r1 = rec[1 * NS + sl];
  const uint32_t flags = __float_as_uint(r1.w);
  Path p = {};
  Item it = {0, 0, 0, 0};
  uint32_t s = 0;
  f3 sum = mk3(0.f, 0.f, 0.f);
  bool has_item = (flags & PF_ITEM) != 0, alive = (flags & PF_ALIVE) != 0;
  if (has_item) {
    const float4 r0 = rec[0 * NS + sl], r2 = rec[2 * NS + sl], r3 = rec[3 * NS + sl], r4 = rec[4 * NS + sl], r5 = rec[5 * NS + sl];
    p.o = mk3(r0.x, r0.y, r0.z); p.acc_len = r0.w;
    p.d = mk3(r1.x, r1.y, r1.z); p.depth = flags >> 8; p.pdf_seen = (flags & PF_PDF) != 0;
    p.A = mk3(r2.x, r2.y, r2.z); p.rng.k0 = __float_as_uint(r2.w);
    p.C = mk3(r3.x, r3.y, r3.z); p.rng.k1 = __float_as_uint(r3.w);
    sum = mk3(r4.x, r4.y, r4.z); p.rng.ctr = __float_as_uint(r4.w);
    const uint32_t pix = __float_as_uint(r5.x);
    it.px = pix & 0xFFFFu; it.py = pix >> 16; it.slot = __float_as_uint(r5.y); it.chunk = __float_as_uint(r5.z);
    s = __float_as_uint(r5.w);
  }
  if (has_item && alive) {  // the search of this path's ray has finished: shade it
    const float4 r6 = rec[6 * NS + sl];
    Hit h;
    h.t = r6.x; h.ref = __float_as_uint(r6.y); h.dfs = SOL_DFS_UNKNOWN; h.u = r6.z; h.v = r6.w;
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
  // ---- new work items from the wave's reservoir [x, y); one global atomic refills it with WF_RESERVOIR items ----
  const unsigned long long need = __ballot(!has_item);
  if (need != 0ull) {
    const uint32_t k = (uint32_t)__popcll(need);
    const uint32_t leader = (uint32_t)__ffsll((long long)need) - 1u;
    uint2 rv = reservoir[wave];  // wave-private: only this wave touches it, in this kernel
    uint32_t x = rv.x, y = rv.y;
    const uint32_t old_avail = y - x;
    uint32_t nb = 0, ne = 0;  // fresh range, when the reservoir cannot serve every lane
    if (k > old_avail) {
      uint32_t b = 0xFFFFFFFFu;
      if (lane == leader) {
        const uint32_t seen = *(volatile uint32_t*)&ctr->work_next;  // saturating: never add once the items are gone
        if (seen < P.n_items) b = atomicAdd(&ctr->work_next, WF_RESERVOIR);
      }
      b = __shfl(b, (int)leader);
      if (b < P.n_items) { nb = b; ne = min(b + WF_RESERVOIR, P.n_items); }
    }
    if (!has_item) {
      const uint32_t r = (uint32_t)__popcll(need & lanes_below);
      uint32_t item = 0xFFFFFFFFu;
      if (r < old_avail) item = x + r;
      else if (nb + (r - old_avail) < ne) item = nb + (r - old_avail);
      if (item != 0xFFFFFFFFu && decode_item(S, P, item, it)) {
        s = P.first_sample + it.chunk * SOL_CHUNK;
        sum = mk3(0.f, 0.f, 0.f);
        has_item = true;
        alive = false;
      }
    }
    if (k <= old_avail) { x += k; }
    else { x = min(nb + (k - old_avail), ne); y = ne; }
    if (lane == leader) reservoir[wave] = make_uint2(x, y);
  }
  if (has_item && !alive) {
    phase_tick<COUNT>(cnt, 2);
    generate_path<COUNT>(S, P.seed_lo, P.seed_hi, it.px, it.py, s, p, cnt);
    alive = true;
  }
  const uint32_t nflags = (has_item ? PF_ITEM : 0u) | (alive ? PF_ALIVE : 0u) | (p.pdf_seen ? PF_PDF : 0u) | (p.depth << 8);
  if (has_item) {
    rec[0 * NS + sl] = make_float4(p.o.x, p.o.y, p.o.z, p.acc_len);
    rec[1 * NS + sl] = make_float4(p.d.x, p.d.y, p.d.z, __uint_as_float(nflags));
    rec[2 * NS + sl] = make_float4(p.A.x, p.A.y, p.A.z, __uint_as_float(p.rng.k0));
    rec[3 * NS + sl] = make_float4(p.C.x, p.C.y, p.C.z, __uint_as_float(p.rng.k1));
    rec[4 * NS + sl] = make_float4(sum.x, sum.y, sum.z, __uint_as_float(p.rng.ctr));
    rec[5 * NS + sl] = make_float4(__uint_as_float(it.px | (it.py << 16)), __uint_as_float(it.slot), __uint_as_float(it.chunk),
                                   __uint_as_float(s));
    rec[6 * NS + sl] = make_float4(0.f, __uint_as_float(p.from), 0.f, 0.f);  // (rule 8: Path::from rides in the hit record until the ray is traced)
  } else if (flags & PF_ITEM) {
    rec[1 * NS + sl] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // a wave that still holds work, or items in its reservoir, keeps the job alive (plain store, idempotent)
  {
    const uint2 rv = reservoir[wave];
    const bool more = __ballot(has_item) != 0ull || rv.y > rv.x || *(volatile uint32_t*)&ctr->work_next < P.n_items;
    if (lane == 0 && more) ctr->live = 1u;
  }
  if (COUNT) flush_counters(cnt, dcnt);
}

// position of the r-th (0-based) set bit of m; requires popcount(m) > r
DEV uint32_t nth_set_bit(unsigned long long m, uint32_t r) {
  uint32_t pos = 0;
#pragma unroll
  for (int w = 32; w > 0; w >>= 1) {
    const uint32_t c = (uint32_t)__popcll((m >> pos) & ((1ull << w) - 1ull));
    if (r >= c) { r -= c; pos += (uint32_t)w; }
  }
  return pos;
}

#ifndef SOL_LDS_STACK_TRACE
#define SOL_LDS_STACK_TRACE 32  // the 8-wide tree pushes up to 7 entries per level; 16 entries spilled to global memory
#endif
#ifndef SOL_WF_MIN_WAVES
#define SOL_WF_MIN_WAVES 5      // 96 VGPRs, no spills; 32 KiB of LDS per workgroup -> 5 workgroups per CU
#endif
template <bool COUNT, bool MEDIUM>
__global__ void __launch_bounds__(SOL_WG, (MEDIUM ? 4 : SOL_WF_MIN_WAVES))  // the nested boundary search needs registers
sol_wf_trace_kernel(const DevScene S, const RenderParams P, WfCounters* __restrict__ ctr, float4* __restrict__ rec,
                    uint32_t* __restrict__ spill, DevCounters* __restrict__ dcnt) {
  __shared__ uint32_t lds_stack[SOL_LDS_STACK_TRACE * SOL_WG];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  Stack st;
  st.lds = (lds_u32*)lds_stack + tid;
  st.spill = (SOL_AS1 uint32_t*)spill + (blockIdx.x * SOL_WG + tid);
  sol_search_context<false>(st, S);
  st.stride = P.total_threads;
  st.depth = SOL_LDS_STACK_TRACE;
  const size_t NS = P.pool_slots;
  const uint32_t* const flagw = reinterpret_cast<const uint32_t*>(rec + NS) + 3;  // .w of record 1, stride 4 words
  const float inf = __builtin_huge_valf();
  Counters cnt = {};
  // wave-uniform scan state: the stretch [cur, end) being read, and the live slots of the last 64 read that are not yet taken
  uint32_t cur = 0, end = 0, pend_base = 0;
  unsigned long long pend = 0ull;
  const uint32_t n_waves = gridDim.x * (SOL_WG / 64), n_groups = (uint32_t)(NS / 64);
  uint32_t next_group = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (SOL_WG / 64) + (tid >> 6)));
  bool dry = false;  // the whole pool has been handed out
  bool have = false;
  uint32_t my_slot = 0, my_from = 0;
  Trav t;
  t.cur = REF_DONE;
  Rng rng_medium = {0, 0, 0};
  uint32_t depth_medium = 0;
  for (;;) {
    const unsigned long long idle = __ballot(!have);
    uint32_t n_need = (uint32_t)__popcll(idle);
    const bool source_empty = dry && pend == 0ull && cur >= end;
    if (!source_empty && (n_need >= SOL_REFILL_MIN || n_need == 64u)) {
      uint32_t my_rank = (uint32_t)__popcll(idle & lanes_below);
      bool need = !have;
      for (;;) {  // wave-uniform loop: hand live slots to idle lanes, reading further flags as needed
        if (pend == 0ull) {
          if (cur >= end) {
            if (dry) break;
            // static round-robin assignment of 64-slot groups to waves: no atomics (a shared cursor - one atomic per
            // stretch on one address - cost 1.7 ms per launch at any fill level), neighbouring groups (= neighbouring
            // pixels, similar cost) go to different waves, and with ~800 slots per wave the totals balance.
            if (next_group >= n_groups) { dry = true; break; }
            cur = next_group * 64u;
            end = cur + 64u;
            next_group += n_waves;
          }
          const uint32_t sl = cur + lane;
          const uint32_t f = sl < end ? flagw[(size_t)sl * 4] : 0u;
          pend = __ballot((f & PF_ALIVE) != 0u);
          pend_base = cur;
          cur += 64u;
          continue;
        }
        const uint32_t avail = (uint32_t)__popcll(pend);
        const uint32_t take = min(avail, n_need);
        if (need && my_rank < take) {
          my_slot = pend_base + nth_set_bit(pend, my_rank);
          const float4 r0 = rec[0 * NS + my_slot], r1 = rec[1 * NS + my_slot];
          if (MEDIUM) {  // the medium's sub-stream needs the path's generator and depth
            const float4 r2 = rec[2 * NS + my_slot], r3 = rec[3 * NS + my_slot];
            rng_medium.k0 = __float_as_uint(r2.w); rng_medium.k1 = __float_as_uint(r3.w);
            depth_medium = __float_as_uint(r1.w) >> 8;
          }
          trav_begin<true>(t, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), RAY_MIN_F, inf, S.wroot, S.rxmin, S.rxmax, S.rymin, S.rymax,
                     S.rzmin, S.rzmax, 0);
          my_from = __float_as_uint(rec[6 * NS + my_slot].y);
          have = true;
          need = false;
        }
        if (take == avail) pend = 0ull;
        else pend &= ~((1ull << nth_set_bit(pend, take)) - 1ull);  // drop the `take` lowest set bits
        my_rank -= take;  // meaningful only for lanes still in need (their rank was >= take)
        n_need -= take;
        if (n_need == 0u) break;
      }
    }
    if (__ballot(have) == 0ull) {
      if (dry && pend == 0ull && cur >= end) break;
      continue;
    }
    if (have) {
      for (int k = 0; k < SOL_TRAV_BURST && t.cur != REF_DONE; ++k) trav_step<COUNT, MEDIUM>(S, t, st, rng_medium, depth_medium, cnt);
      if (t.cur == REF_DONE && sol_self_hit(my_from, t.h)) {  // (rule 8, sol_path.h: searched again behind the primitive the ray left)
        trav_begin<true>(t, t.o, t.d, sol_behind(t.h.t), inf, S.wroot, S.rxmin, S.rxmax, S.rymin, S.rymax, S.rzmin, S.rzmax, 0);
      } else if (t.cur == REF_DONE) {
        rec[6 * NS + my_slot] = make_float4(t.h.t, __uint_as_float(t.h.ref), t.h.u, t.h.v);
        if (COUNT) cnt.rays++;
        have = false;
      }
    }
  }
  if (COUNT) flush_counters(cnt, dcnt);
}

// ---- launch wrappers (called from sol_api.cpp) ---------------------------------------------------------------
hipError_t sol_launch_wf_shade(const DevScene& S, const RenderParams& P, float* acc, float* partial, void* ctr, void* rec,
                               void* reservoir, DevCounters* cnt, bool count, hipStream_t stream) {
  const uint32_t grid = P.pool_slots / SOL_WG;
  if (count) hipLaunchKernelGGL((sol_wf_shade_kernel<true>), dim3(grid), dim3(SOL_WG), 0, stream, S, P, acc, partial, (WfCounters*)ctr, (float4*)rec, (uint2*)reservoir, cnt);
  else hipLaunchKernelGGL((sol_wf_shade_kernel<false>), dim3(grid), dim3(SOL_WG), 0, stream, S, P, acc, partial, (WfCounters*)ctr, (float4*)rec, (uint2*)reservoir, cnt);
  return hipGetLastError();
}
hipError_t sol_launch_wf_trace(const DevScene& S, const RenderParams& P, void* ctr, void* rec, uint32_t* spill, DevCounters* cnt,
                               uint32_t grid, bool count, bool medium, hipStream_t stream) {
#define WF_TRACE(C, M) hipLaunchKernelGGL((sol_wf_trace_kernel<C, M>), dim3(grid), dim3(SOL_WG), 0, stream, S, P, (WfCounters*)ctr, (float4*)rec, spill, cnt)
  if (count) { if (medium) WF_TRACE(true, true); else WF_TRACE(true, false); }
  else { if (medium) WF_TRACE(false, true); else WF_TRACE(false, false); }
#undef WF_TRACE
  return hipGetLastError();
}
int sol_wf_trace_blocks_per_cu(bool count, bool medium) {
  int n = 0;
  hipError_t e;
  if (count) e = medium ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sol_wf_trace_kernel<true, true>, SOL_WG, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sol_wf_trace_kernel<true, false>, SOL_WG, 0);
  else e = medium ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sol_wf_trace_kernel<false, true>, SOL_WG, 0)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sol_wf_trace_kernel<false, false>, SOL_WG, 0);
  if (e != hipSuccess || n < 1) n = 1;
  return n;
}
size_t sol_wf_pool_bytes(uint32_t slots) { return (size_t)POOL_RECORDS * slots * sizeof(float4); }
int sol_wf_lds_stack_depth() { return SOL_LDS_STACK_TRACE; }
#endif  // SOL_AB_KERNELS
