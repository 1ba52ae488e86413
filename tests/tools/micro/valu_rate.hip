// valu_rate.hip -- microbenchmark (not product code): how many wave64 VALU instructions per cycle one gfx950 SIMD issues
// with 1..8 resident waves, for independent v_fma_f32 streams. Settles the peak the "valu_issue" roofline of bench.py is
// priced against (MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles on the SIMD-32, one wave alone issues every 4).
//   hipcc --offload-arch=gfx950 -O3 tests/tools/micro/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <vector>

template <int ACTIVE>
__global__ void __launch_bounds__(256) fma_loop(float* out, int iters, long long* cycles) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0000001f, c = 0.5f;
  const bool on = (threadIdx.x & 63) < ACTIVE;  // lane mask: partial waves cost the same issue slots
  long long t0 = __builtin_amdgcn_s_memtime();
  if (on) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        // inline asm: the compiler would pack pairs into v_pk_fma_f32
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int ACTIVE>
static void run(int wg_per_cu, int threads) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount, iters = 20000, grid = n_cu * wg_per_cu;
  float* out; long long* cyc;
  hipMalloc(&out, (size_t)grid * threads * sizeof(float));
  hipMalloc(&cyc, grid * sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(fma_loop<ACTIVE>, dim3(grid), dim3(threads), 0, 0, out, 100, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(fma_loop<ACTIVE>, dim3(grid), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid);
  hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= grid;
  const double waves_per_simd = (double)wg_per_cu * (threads / 64) / 4.0;
  const double inst_per_wave = (double)iters * 64.0;
  // instructions issued per SIMD = waves_per_simd * inst_per_wave, over `mean` shader cycles
  std::printf("active_lanes %2d  waves/SIMD %.2f  kernel %.3f ms  cycles/wave-loop %.0f  VALU inst per SIMD-cycle %.3f  (clock %.2f GHz)\n", ACTIVE,
              waves_per_simd, ms, mean, waves_per_simd * inst_per_wave / mean, mean / (ms * 1e6));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int wg : {1, 2, 3, 4, 6, 8}) run<64>(wg, 256);
  run<64>(1, 64);  // one wave per CU: one SIMD of four
  for (int wg : {1, 2, 4}) run<29>(wg, 256);
  return 0;
}
