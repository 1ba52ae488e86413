// sol_math.h -- device fp32 vector math, the counter-based RNG and the fp32 elementary-function contract.
//
// Compiled with -ffp-contract=off: every expression below is the plain IEEE mul/add sequence it spells, in the
// association the reference writes (src/geo/vec3.rs), so discrete decisions (hit / miss, which primitive, which
// branch of a material) agree with the fp32 restatement the tests compare against. Division and sqrt are the
// correctly rounded forms (hipcc default for fp32 without fast-math).
//
// Elementary functions (DESIGN.md "fp32 arithmetic contract"): sin/cos of 2*pi*r, acos, atan2 and ln are fixed
// polynomials (Cephes single-precision coefficients, A&S 4.4.46 for acos) evaluated in the written order; libm /
// OCML are not used for them, since their last-bit behaviour is not specified.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEV __device__ __forceinline__

// Scene pointers reach the device code through a DevScene record read from memory, so the compiler sees them as generic
// ("flat") pointers: every access becomes flat_load, which counts against BOTH vmcnt and lgkmcnt - each wait for a node or a
// triangle then also drains the queue of the LDS traversal stack - and pays the aperture check. They are all global
// (hipMalloc). Neither an address-space round trip nor an is_shared/is_private assumption survives to the backend; what does
// is a load THROUGH an address-space-1 pointer, so every scene record is fetched with these helpers (global_load_dword*).
// Lane mask of a condition. HIP's __ballot(int) compares a 0 / 1 VALUE with zero: when the condition is a lane mask already
// (a compare result) the compiler materialises it with v_cndmask and compares again, two 4-cycle instructions per ballot in
// the search loop; the builtin takes the condition as it is.
#define sol_ballot(cond) __builtin_amdgcn_ballot_w64(cond)
#define SOL_AS1 __attribute__((address_space(1)))
typedef float sol_v4f __attribute__((ext_vector_type(4)));
typedef uint32_t sol_v4u __attribute__((ext_vector_type(4)));
DEV float4 ldg_f4(const void* p) {
  const sol_v4f v = *(const SOL_AS1 sol_v4f*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
DEV uint4 ldg_u4(const void* p) {
  const sol_v4u v = *(const SOL_AS1 sol_v4u*)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
DEV uint32_t ldg_u32(const void* p) { return *(const SOL_AS1 uint32_t*)p; }
DEV int32_t ldg_i32(const void* p) { return *(const SOL_AS1 int32_t*)p; }
DEV uint8_t ldg_u8(const void* p) { return *(const SOL_AS1 uint8_t*)p; }
// a whole 16-byte-aligned record (sizeof a multiple of 16) by value
template <typename T>
DEV T ldg_rec(const T* p) {
  static_assert(sizeof(T) % 16 == 0 && alignof(T) >= 16, "device records are 16-byte multiples");
  union { T rec; sol_v4u q[sizeof(T) / 16]; } u;
#pragma unroll
  for (unsigned k = 0; k < sizeof(T) / 16; ++k) u.q[k] = ((const SOL_AS1 sol_v4u*)p)[k];
  return u.rec;
}

// Correctly rounded fp32 square root. `sqrtf` lowers to v_sqrt_f32 plus the fix-up sequence (16 instructions);
// `__fsqrt_rn` on ROCm 7.2 is the bare 1-ulp v_sqrt_f32 despite its name (measured: 15 % of results differ from IEEE).
DEV float sol_sqrt(float x) { return sqrtf(x); }

struct f3 {
  float x, y, z;
};
DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
DEV f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
DEV f3 operator*(f3 a, float t) { return f3{a.x * t, a.y * t, a.z * t}; }
DEV f3 operator/(f3 a, float t) { return f3{a.x / t, a.y / t, a.z / t}; }
DEV f3 neg3(f3 a) { return f3{-a.x, -a.y, -a.z}; }
DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                                  // vec3.rs:227
DEV f3 cross3(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }  // vec3.rs:238
DEV float len2(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
DEV float len3(f3 a) { return sol_sqrt(len2(a)); }
DEV f3 unit3(f3 a) { return a / len3(a); }                                                                // vec3.rs:287
DEV f3 reflect3(f3 v, f3 n) { return v - n * (dot3(v, n) * 2.0f); }                                       // vec3.rs:329
DEV f3 refract3(f3 v, f3 n, float ior) {                                                                  // vec3.rs:341-346
  float cos_theta = fminf(dot3(neg3(v), n), 1.0f);
  f3 perp = (n * cos_theta + v) * ior;
  f3 par = n * (-sol_sqrt(fabsf(1.0f - len2(perp))));
  return perp + par;
}

struct Onb {
  f3 tangent, bi_tangent, normal;
};
DEV Onb onb_new(f3 w) {  // geo/mod.rs:245-257
  f3 uw = unit3(w);
  f3 a = fabsf(uw.x) > 0.9f ? mk3(0.f, 1.f, 0.f) : mk3(1.f, 0.f, 0.f);
  f3 v = unit3(cross3(uw, a));
  f3 u = cross3(uw, v);
  return Onb{u, v, uw};
}
DEV f3 onb_local(const Onb& o, f3 a) { return o.tangent * a.x + o.bi_tangent * a.y + o.normal * a.z; }  // geo/mod.rs:260-262

// ---- counter-based RNG (replaces src/random.rs; DESIGN.md "RNG") ---------------------------------------------
DEV uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
  return x;
}
struct Rng {
  uint32_t k0, k1, ctr;
};
DEV void rng_init(Rng& r, uint32_t seed_lo, uint32_t seed_hi, uint32_t pixel, uint32_t sample) {
  r.k0 = mix32(mix32(pixel ^ seed_lo) + sample);
  r.k1 = mix32(mix32(sample ^ seed_hi ^ 0x9E3779B9u) + pixel);
  r.ctr = 0;
}
DEV uint32_t rng_bits(const Rng& r, uint32_t c) { return mix32(mix32(r.k0 + c * 0x9E3779B9u) ^ r.k1); }
DEV float u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
DEV float rnd(Rng& r) { return u32_to_unit(rng_bits(r, r.ctr++)); }                  // random_normal_float
DEV float rnd_range(Rng& r, float mn, float mx) { return rnd(r) * (mx - mn) + mn; }  // random_float
DEV uint32_t rnd_index(Rng& r, uint32_t n) { return __umulhi(rng_bits(r, r.ctr++), n); }  // random_element_index

// ---- fp32 elementary functions ------------------------------------------------------------------------------
#define SOL_PI 3.14159265358979323846f
DEV void sincos2pi(float r, float& c, float& s) {
  float t = r * 4.0f;
  float j = floorf(t + 0.5f);
  float f = t - j;
  float x = f * 1.57079632679489661923f;
  float z = x * x;
  float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
  float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  int q = ((int)j) & 3;
  c = (q == 0) ? cp : (q == 1) ? -sp : (q == 2) ? -cp : sp;
  s = (q == 0) ? sp : (q == 1) ? cp : (q == 2) ? -sp : -cp;
}
DEV float acos_r(float x) {
  float a = fabsf(x);
  float p = -0.0012624911f;
  p = p * a + 0.0066700901f;
  p = p * a - 0.0170881256f;
  p = p * a + 0.0308918810f;
  p = p * a - 0.0501743046f;
  p = p * a + 0.0889789874f;
  p = p * a - 0.2145988016f;
  p = p * a + 1.5707963050f;
  float r = sol_sqrt(1.0f - a) * p;
  return x < 0.0f ? SOL_PI - r : r;
}
DEV float atan2_r(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  if (mx == 0.0f) return 0.0f;
  float a = mn / mx;
  float off = 0.0f;
  if (a > 0.4142135623730950f) { off = 0.78539816339744831f; a = (a - 1.0f) / (a + 1.0f); }
  float z = a * a;
  float r = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * a + a;
  r = off + r;
  if (ay > ax) r = 1.57079632679489661923f - r;
  if (x < 0.0f) r = SOL_PI - r;
  return y < 0.0f ? -r : r;
}
DEV float log_r(float x) {
  if (x <= 0.0f) return -__builtin_huge_valf();
  uint32_t b = __float_as_uint(x);
  int e = (int)(b >> 23) - 126;
  float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F000000u);
  if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float y = 7.0376836292e-2f;
  y = y * m - 1.1514610310e-1f;
  y = y * m + 1.1676998740e-1f;
  y = y * m - 1.2420140846e-1f;
  y = y * m + 1.4249322787e-1f;
  y = y * m - 1.6668057665e-1f;
  y = y * m + 2.0000714765e-1f;
  y = y * m - 2.4999993993e-1f;
  y = y * m + 3.3333331174e-1f;
  y = y * m * z;
  float fe = (float)e;
  y = y + fe * -2.12194440e-4f;
  y = y - 0.5f * z;
  float r = m + y;
  r = r + fe * 0.693359375f;
  return r;
}
