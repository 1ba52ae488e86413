// fetch_gather.hip -- microbenchmark (not product code) behind two counter questions of DESIGN.md 3 / VERDICT r03 #5:
//   (i)  does rocprofv3's FETCH_SIZE need the gfx950 "x2" correction for DIVERGENT 16-byte gathers as it does for wide coalesced
//        reads? Every lane reads 16 bytes at a random 128-byte-aligned line of a table far larger than the 256 MiB Infinity
//        Cache (default 4 GiB), so every access is one line miss all the way to HBM and the byte count is known: the kernel
//        prints its access count, and the access RATE bounds the bytes each access can have moved (HBM delivers <= ~6.3 TB/s).
//   (ii) what does TD_TD_BUSY mean for such a kernel (no vector ALU work to speak of)? Run under --pmc TD_TD_BUSY_sum TA_BUSY_avr
//        GRBM_GUI_ACTIVE and compare with the render kernel's 0.99.
// Modes: gather (random 16 B per lane), gather_l2 (the same over a 2 MiB table: hits), stream (coalesced 16 B per lane, whole table).
//   hipcc --offload-arch=gfx950 -O3 -w tests/tools/micro/fetch_gather.hip -o gpurun_out/fetch_gather
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/fg_fetch -- gpurun_out/fetch_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define AS1 __attribute__((address_space(1)))

__global__ void __launch_bounds__(256) gather_lines(const v4u* table, unsigned long long n_lines_mask, int iters, unsigned* out) {
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  unsigned long long state = (unsigned long long)gid * 0x9E3779B97F4A7C15ull + 12345ull;
  v4u acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      const unsigned long long line = (state >> 20) & n_lines_mask;
      const v4u v = ((const AS1 v4u*)table)[line * 8ull + ((state >> 17) & 7ull)];  // one 16-byte piece of a random 128-byte line
      acc ^= v;
    }
  }
  out[gid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
__global__ void __launch_bounds__(256) stream_lines(const v4u* table, unsigned long long n_pieces, unsigned* out) {
  const unsigned long long gid = blockIdx.x * 256ull + threadIdx.x, stride = (unsigned long long)gridDim.x * 256ull;
  v4u acc = {0, 0, 0, 0};
  for (unsigned long long p = gid; p < n_pieces; p += stride) acc ^= ((const AS1 v4u*)table)[p];
  out[gid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 4;
  const size_t bytes = gib << 30;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount, grid = n_cu * 8;
  v4u* table;
  unsigned* out;
  if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&out, (size_t)grid * 256 * 4) != hipSuccess) { std::printf("allocation failed\n"); return 1; }
  hipMemset(table, 1, bytes);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  struct { const char* name; size_t table_bytes; int iters; } runs[2] = {{"gather_hbm", bytes, 400}, {"gather_l2", (size_t)2 << 20, 400}};
  for (auto& r : runs) {
    const unsigned long long mask = r.table_bytes / 128 - 1;
    hipLaunchKernelGGL(gather_lines, dim3(grid), dim3(256), 0, 0, table, mask, 8, out);  // warm-up (not counted below: 1/50 of the main launch)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(gather_lines, dim3(grid), dim3(256), 0, 0, table, mask, r.iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double accesses = (double)grid * 256.0 * r.iters * 8.0;
    std::printf("%-11s table %6zu MiB  lane accesses %.4g (16 B each = %.4g bytes requested, %.4g bytes if every access moves a 128-B line, %.4g if a 64-B half)  %8.3f ms  "
                "%.3g accesses/s = %.2f TB/s at 128 B, %.2f TB/s at 64 B, %.2f TB/s at 32 B\n",
                r.name, r.table_bytes >> 20, accesses, accesses * 16, accesses * 128, accesses * 64, ms, accesses / (ms * 1e-3), accesses * 128 / (ms * 1e-3) / 1e12,
                accesses * 64 / (ms * 1e-3) / 1e12, accesses * 32 / (ms * 1e-3) / 1e12);
  }
  hipEventRecord(e0);
  hipLaunchKernelGGL(stream_lines, dim3(grid), dim3(256), 0, 0, table, (unsigned long long)(bytes / 16), out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  std::printf("%-11s table %6zu MiB  %.4g bytes read once, coalesced  %8.3f ms  %.2f TB/s\n", "stream_hbm", bytes >> 20, (double)bytes, ms, (double)bytes / (ms * 1e-3) / 1e12);
  return 0;
}
