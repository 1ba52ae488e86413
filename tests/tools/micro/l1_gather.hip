// l1_gather.hip -- microbenchmark (not product code): cost of one global_load_dwordx4 wave-instruction on a gfx950 CU as a
// function of how its 64 lane addresses spread over cache lines, for an L1-resident (16 KiB), an L2-resident (2 MiB) and an
// Infinity-Cache-resident (64 MiB) table. Answers: what does a BVH node fetch cost when every lane visits a different node, and
// what would it cost if groups of lanes fetched one node together ("node fetch").
//   hipcc --offload-arch=gfx950 -O3 -w tests/tools/micro/l1_gather.hip -o gpurun_out/l1_gather && gpurun_out/l1_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-result"

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define AS1 __attribute__((address_space(1)))

// GROUP lanes share one random 16*GROUP-byte aligned segment and read its consecutive 16-byte pieces (GROUP = 1: every lane its
// own segment; 64: one contiguous KiB per instruction). PIECES loads per iteration from consecutive 16*GROUP*k offsets? No: each of
// the PIECES loads goes to a fresh random segment (independent), so the figure is per load instruction.
template <int GROUP>
__global__ void __launch_bounds__(256) gather(const v4u* table, unsigned n_pieces_mask, int iters, unsigned* out) {
  const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  unsigned state = wave * 9781u + (lane / GROUP) * 6271u + 12345u;
  const unsigned sub = lane % GROUP;
  v4u acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      state = state * 1664525u + 1013904223u;
      const unsigned seg = (state >> 8) & n_pieces_mask & ~(unsigned)(GROUP - 1);
      const v4u v = ((const AS1 v4u*)table)[seg + sub];
      acc ^= v;
    }
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// The same with 8-byte accesses (global_load_dwordx2) of every lane's own random 16-byte segment: what a node fetch split into
// per-axis near / far halves would cost per instruction.
__global__ void __launch_bounds__(256) gather_x2(const v4u* table, unsigned n_pieces_mask, int iters, unsigned* out) {
  typedef unsigned int v2u __attribute__((ext_vector_type(2)));
  const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  unsigned state = wave * 9781u + lane * 6271u + 12345u;
  v2u acc = {0, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      state = state * 1664525u + 1013904223u;
      const unsigned seg = (state >> 8) & n_pieces_mask;
      const v2u v = ((const AS1 v2u*)table)[2u * seg + ((state >> 7) & 1u)];
      acc ^= v;
    }
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc.x ^ acc.y;
}
static void run_x2(const v4u* table, size_t bytes, const char* where, int wg_per_cu) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount, grid = n_cu * wg_per_cu, iters = 2000;
  unsigned* out;
  hipMalloc(&out, (size_t)grid * 256 * 4);
  const unsigned mask = (unsigned)(bytes / 16 - 1);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(gather_x2, dim3(grid), dim3(256), 0, 0, table, mask, 50, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(gather_x2, dim3(grid), dim3(256), 0, 0, table, mask, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = (double)wg_per_cu * 4 * iters * 8;
  const double ns_per_instr = ms * 1e6 / instr_per_cu;
  std::printf("%-14s dwordx2, every lane its own segment  waves/CU %2d  %7.3f ms  %6.1f ns per wave-load per CU (~%5.1f clk @2.1GHz)\n", where,
              wg_per_cu * 4, ms, ns_per_instr, ns_per_instr * 2.1);
  hipFree(out);
}

template <int GROUP>
static void run(const v4u* table, size_t bytes, const char* where, int wg_per_cu) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount, grid = n_cu * wg_per_cu, iters = 2000;
  unsigned* out;
  hipMalloc(&out, (size_t)grid * 256 * 4);
  const unsigned mask = (unsigned)(bytes / 16 - 1);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(gather<GROUP>, dim3(grid), dim3(256), 0, 0, table, mask, 50, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(gather<GROUP>, dim3(grid), dim3(256), 0, 0, table, mask, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = (double)wg_per_cu * 4 * iters * 8;  // wave-level load instructions per CU
  const double ns_per_instr = ms * 1e6 / instr_per_cu;
  std::printf("%-14s lanes/segment %2d  waves/CU %2d  %7.3f ms  %6.1f ns per wave-load per CU (~%5.1f clk @2.1GHz)  %6.2f TB/s chip\n", where, GROUP,
              wg_per_cu * 4, ms, ns_per_instr, ns_per_instr * 2.1, 1024.0 * instr_per_cu * n_cu / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  const size_t sizes[3] = {16u << 10, 2u << 20, 64u << 20};
  const char* names[3] = {"L1 16KiB", "L2 2MiB", "MALL 64MiB"};
  for (int t = 0; t < 3; ++t) {
    v4u* table;
    hipMalloc(&table, sizes[t]);
    hipMemset(table, 1, sizes[t]);
    for (int wg : {4}) {
      run<1>(table, sizes[t], names[t], wg);
      run<2>(table, sizes[t], names[t], wg);
      run<4>(table, sizes[t], names[t], wg);
      run<8>(table, sizes[t], names[t], wg);
      run<16>(table, sizes[t], names[t], wg);
      run<64>(table, sizes[t], names[t], wg);
      run_x2(table, sizes[t], names[t], wg);
    }
    hipFree(table);
  }
  return 0;
}
