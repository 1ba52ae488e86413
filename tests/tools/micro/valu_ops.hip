// valu_ops.hip -- microbenchmark (not product code): issue cost of the vector instructions the 7-wide node test is made of,
// relative to v_fma_f32, on one gfx950 SIMD with 4 and 8 resident waves (8 independent chains per wave).
//   hipcc --offload-arch=gfx950 -O3 tests/tools/micro/valu_ops.hip -o gpurun_out/valu_ops && gpurun_out/valu_ops
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <vector>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define OPERANDS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)

#define FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define MUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define ADD(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define MAXF(k) "v_max_f32 %" #k ", %" #k ", %8\n"
#define MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
#define CVTUB(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n"
#define CVTU(k) "v_cvt_f32_u32 %" #k ", %" #k "\n"
#define ANDB(k) "v_and_b32 %" #k ", %" #k ", %8\n"
#define ADDU(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 1, %9\n"
#define BFE(k) "v_bfe_u32 %" #k ", %" #k ", 3, 9\n"
#define PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define CNDM(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define CMPLE(k) "v_cmp_le_f32 vcc, %" #k ", %8\n"
#define CMPLES(k) "v_cmp_le_f32 s[20:21], %" #k ", %8\n"
#define PKFMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define MOV(k) "v_mov_b32 %" #k ", %8\n"
#define MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n"
#define RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define OR3(k) "v_or3_b32 %" #k ", %" #k ", %8, %9\n"
#define BCNT(k) "v_bcnt_u32_b32 %" #k ", %" #k ", %8\n"
#define MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define SDWA(k) "v_or_b32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define CNDS(k) "v_cndmask_b32 %" #k ", %" #k ", %8, s[20:21]\n"
#define CNDMIX(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define CNDMIX3(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n v_fma_f32 %" #k ", %" #k ", %8, %9\n v_fma_f32 %" #k ", %" #k ", %8, %9\n v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define ADDC(k) "v_addc_co_u32 %" #k ", vcc, %" #k ", %" #k ", vcc\n"
#define CMPADDC(k) "v_cmp_le_f32 vcc, %9, %8\n v_addc_co_u32 %" #k ", vcc, %" #k ", %" #k ", vcc\n"
#define CMPCND(k) "v_cmp_le_f32 vcc, %9, %8\n v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define LSHR(k) "v_lshrrev_b32 %" #k ", 8, %" #k "\n"
#define XORB(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define SUBF(k) "v_sub_f32 %" #k ", %" #k ", %8\n"
#define FMAC(k) "v_fmac_f32 %" #k ", %8, %9\n"
#define FMAMIX(k) "v_fma_mix_f32 %" #k ", %" #k ", %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define FMAMIXLO(k) "v_fma_mix_f32 %" #k ", %" #k ", %8, %9 op_sel_hi:[1,0,0]\n"
#define ALIGNBIT(k) "v_alignbit_b32 %" #k ", %" #k ", %8, 31\n"
#define MIN3(k) "v_min3_f32 %" #k ", %" #k ", %8, %9\n"
#define CVTF16(k) "v_cvt_f32_f16 %" #k ", %" #k "\n"
#define ORB(k) "v_or_b32 %" #k ", %" #k ", %8\n"
#define LSHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define MULU24(k) "v_mul_u32_u24 %" #k ", %" #k ", %8\n"
#define MADU24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n"
#define ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %9\n"
#define MINU(k) "v_min_u32 %" #k ", %" #k ", %8\n"
#define FFBL(k) "v_ffbl_b32 %" #k ", %" #k "\n"
#define LDEXP(k) "v_ldexp_f32 %" #k ", %" #k ", %8\n"
#define BITOP3(k) "v_bitop3_b32 %" #k ", %" #k ", %8, %9 bitop3:0x30\n"
#define SNOP(k) "s_nop 0\n"
#define SMOV(k) "s_mov_b32 s20, s21\n"

enum Op { kFma, kMul, kAdd, kMax, kMax3, kCvtUb, kCvtU, kAnd, kAddU, kLshlOr, kBfe, kPerm, kCndmask, kCmpVcc, kCmpS, kPkFma, kMov, kMed3, kRcp, kOr3, kBcnt, kMulLo, kSdwa, kSnop, kSmov, kCndS, kCndMix, kCndMix3, kAddc, kCmpAddc, kCmpCnd, kLshr, kXor, kSubF, kFmac, kFmaMix, kFmaMixLo, kAlignbit, kMin3, kCvtF16, kOr, kLshl, kMulU24, kMadU24, kAdd3, kLshlAdd, kMinU, kFfbl, kLdexp, kBitop3, kOps };
static const char* kNames[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_max3_f32", "v_cvt_f32_ubyte1", "v_cvt_f32_u32", "v_and_b32", "v_add_u32", "v_lshl_or_b32", "v_bfe_u32", "v_perm_b32", "v_cndmask_b32", "v_cmp_le_f32 vcc", "v_cmp_le_f32 sgpr", "v_pk_fma_f32", "v_mov_b32", "v_med3_f32", "v_rcp_f32", "v_or3_b32", "v_bcnt_u32_b32", "v_mul_lo_u32", "v_or_b32_sdwa", "s_nop 0", "s_mov_b32", "v_cndmask_b32 sgpr", "cndmask+fma (2 instr)", "cndmask+3fma (4 instr)", "v_addc_co_u32", "cmp+addc (2 instr)", "cmp+cndmask (2 instr)", "v_lshrrev_b32", "v_xor_b32", "v_sub_f32", "v_fmac_f32", "v_fma_mix_f32 (f16 hi)", "v_fma_mix_f32 (f16 lo)", "v_alignbit_b32", "v_min3_f32", "v_cvt_f32_f16", "v_or_b32", "v_lshlrev_b32", "v_mul_u32_u24", "v_mad_u32_u24", "v_add3_u32", "v_lshl_add_u32", "v_min_u32", "v_ffbl_b32", "v_ldexp_f32", "v_bitop3_b32"};

template <int OP>
__global__ void __launch_bounds__(256) op_loop(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0000001f, c = 0.5f;
  if (OP == kPkFma) {  // register pairs
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6}, pb = {b, b}, pc = {c, c};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        asm volatile(REP8(PKFMA) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    return;
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (OP == kFma) asm volatile(REP8(FMA) OPERANDS);
      if (OP == kMul) asm volatile(REP8(MUL) OPERANDS);
      if (OP == kAdd) asm volatile(REP8(ADD) OPERANDS);
      if (OP == kMax) asm volatile(REP8(MAXF) OPERANDS);
      if (OP == kMax3) asm volatile(REP8(MAX3) OPERANDS);
      if (OP == kCvtUb) asm volatile(REP8(CVTUB) OPERANDS);
      if (OP == kCvtU) asm volatile(REP8(CVTU) OPERANDS);
      if (OP == kAnd) asm volatile(REP8(ANDB) OPERANDS);
      if (OP == kAddU) asm volatile(REP8(ADDU) OPERANDS);
      if (OP == kLshlOr) asm volatile(REP8(LSHLOR) OPERANDS);
      if (OP == kBfe) asm volatile(REP8(BFE) OPERANDS);
      if (OP == kPerm) asm volatile(REP8(PERM) OPERANDS);
      if (OP == kCndmask) asm volatile(REP8(CNDM) OPERANDS : "vcc");
      if (OP == kCmpVcc) asm volatile(REP8(CMPLE) OPERANDS : "vcc");
      if (OP == kCmpS) asm volatile(REP8(CMPLES) OPERANDS : "s20", "s21");
      if (OP == kMov) asm volatile(REP8(MOV) OPERANDS);
      if (OP == kMed3) asm volatile(REP8(MED3) OPERANDS);
      if (OP == kRcp) asm volatile(REP8(RCP) OPERANDS);
      if (OP == kOr3) asm volatile(REP8(OR3) OPERANDS);
      if (OP == kBcnt) asm volatile(REP8(BCNT) OPERANDS);
      if (OP == kMulLo) asm volatile(REP8(MULLO) OPERANDS);
      if (OP == kSdwa) asm volatile(REP8(SDWA) OPERANDS);
      if (OP == kSnop) asm volatile(REP8(SNOP) OPERANDS);
      if (OP == kCndS) asm volatile(REP8(CNDS) OPERANDS);
      if (OP == kCndMix) asm volatile(REP8(CNDMIX) OPERANDS : "vcc");
      if (OP == kCndMix3) asm volatile(REP8(CNDMIX3) OPERANDS : "vcc");
      if (OP == kAddc) asm volatile(REP8(ADDC) OPERANDS : "vcc");
      if (OP == kCmpAddc) asm volatile(REP8(CMPADDC) OPERANDS : "vcc");
      if (OP == kCmpCnd) asm volatile(REP8(CMPCND) OPERANDS : "vcc");
      if (OP == kLshr) asm volatile(REP8(LSHR) OPERANDS);
      if (OP == kXor) asm volatile(REP8(XORB) OPERANDS);
      if (OP == kSubF) asm volatile(REP8(SUBF) OPERANDS);
      if (OP == kFmac) asm volatile(REP8(FMAC) OPERANDS);
      if (OP == kFmaMix) asm volatile(REP8(FMAMIX) OPERANDS);
      if (OP == kFmaMixLo) asm volatile(REP8(FMAMIXLO) OPERANDS);
      if (OP == kAlignbit) asm volatile(REP8(ALIGNBIT) OPERANDS);
      if (OP == kMin3) asm volatile(REP8(MIN3) OPERANDS);
      if (OP == kCvtF16) asm volatile(REP8(CVTF16) OPERANDS);
      if (OP == kOr) asm volatile(REP8(ORB) OPERANDS);
      if (OP == kLshl) asm volatile(REP8(LSHL) OPERANDS);
      if (OP == kMulU24) asm volatile(REP8(MULU24) OPERANDS);
      if (OP == kMadU24) asm volatile(REP8(MADU24) OPERANDS);
      if (OP == kAdd3) asm volatile(REP8(ADD3) OPERANDS);
      if (OP == kLshlAdd) asm volatile(REP8(LSHLADD) OPERANDS);
      if (OP == kMinU) asm volatile(REP8(MINU) OPERANDS);
      if (OP == kFfbl) asm volatile(REP8(FFBL) OPERANDS);
      if (OP == kLdexp) asm volatile(REP8(LDEXP) OPERANDS);
      if (OP == kBitop3) asm volatile(REP8(BITOP3) OPERANDS);
      if (OP == kSmov) asm volatile(REP8(SMOV) OPERANDS : "s20");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// 64-bit address arithmetic (register pairs)
template <int WHICH>
__global__ void __launch_bounds__(256) op64_loop(unsigned long long* out, int iters) {
  unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned long long b = 48;
  unsigned int c = 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#define L64(k) "v_lshl_add_u64 %" #k ", %" #k ", 0, %8\n"
#define M64(k) "v_mad_u64_u32 %" #k ", s[20:21], %9, 48, %" #k "\n"
      if (WHICH == 0) asm volatile(REP8(L64) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      if (WHICH == 1) asm volatile(REP8(M64) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int WHICH>
static double run64(int wg_per_cu) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount, iters = 8000, grid = n_cu * wg_per_cu, threads = 256;
  unsigned long long* out;
  hipMalloc(&out, (size_t)grid * threads * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(op64_loop<WHICH>, dim3(grid), dim3(threads), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(op64_loop<WHICH>, dim3(grid), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return (double)wg_per_cu * iters * 64.0 / (ms * 1e6);
}

template <int OP>
static double run(int wg_per_cu) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount, iters = 8000, grid = n_cu * wg_per_cu, threads = 256;
  float* out;
  hipMalloc(&out, (size_t)grid * threads * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(op_loop<OP>, dim3(grid), dim3(threads), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(op_loop<OP>, dim3(grid), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return (double)wg_per_cu * iters * 64.0 / (ms * 1e6);  // G instructions per second per SIMD
}

template <int OP>
static void all(double* fma) {
  const double r4 = run<OP>(4), r8 = run<OP>(8);
  if (OP == kFma) { fma[0] = r4; fma[1] = r8; }
  std::printf("%-20s 4 waves/SIMD %6.3f G/s/SIMD (%.2f x fma)   8 waves/SIMD %6.3f G/s/SIMD (%.2f x fma)\n", kNames[OP], r4, fma[0] / r4, r8, fma[1] / r8);
  if constexpr (OP + 1 < kOps) all<OP + 1>(fma);
}

int main() {
  double fma[2] = {1, 1};
  all<0>(fma);
  std::printf("%-20s 4 waves/SIMD %6.3f G/s/SIMD   8 waves/SIMD %6.3f G/s/SIMD\n", "v_lshl_add_u64", run64<0>(4), run64<0>(8));
  std::printf("%-20s 4 waves/SIMD %6.3f G/s/SIMD   8 waves/SIMD %6.3f G/s/SIMD\n", "v_mad_u64_u32", run64<1>(4), run64<1>(8));
  return 0;
}
