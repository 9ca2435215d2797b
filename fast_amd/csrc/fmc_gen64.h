// fmc_gen64.h -- the device generator at the reference's precision (GPU_RNG_PRECISION 'f64'), fast form.
//
// The reference draws 53-bit normals and colours them in float64 (fast/funcs.py:352-356, fast/fast.py:593-594).  Our float64
// generator is defined in fmc_kernels.h (box_muller_f64) and restated in oracle/devrng.py (box_muller_f64):
//     u = (a 2^21 + (a2 >> 11) + 1/2) 2^-53,   t = ((b >> 9) 2^30 + (b2 >> 2)) 2^-53,   sqrt(-2 ln u) exp(2 pi i t)
// with (a, b) the words of the float32 draw and (a2, b2) those of the second stream.  Round 3 evaluated it with libm's log /
// sqrt / sincospi (~600 VALU instructions per coefficient) in a staging kernel; this header is the SAME definition in ~90
// instructions, cheap enough to be fused into the row kernels (fmc_kernels.h: MODE 2), so that no coefficient ever passes
// through HBM:
//   * y = -2 ln u: u = 2^K m with m in [0.75, 1.5) by integer arithmetic on the hi word; the top seven mantissa bits of the
//     reduced hi word index a 128-entry table (-2 c_j, 2 ln c_j), c_j ~ 1 / centre of interval j (exactly 1 for the two
//     intervals that touch m = 1, so that u -> 1 keeps full RELATIVE accuracy); r' = fma(m, -2 c_j, 2) = -2 (m c_j - 1) is
//     exact up to one rounding, |r'| <= 2^-6, and  -2 ln(1 + r) = r' + r'^2 Q(r')  with a degree-5 near-minimax Q
//     (tools/gen64_design.py: relative error of y < 2e-18 before rounding);
//   * sqrt y: v_rsq_f32 seed (2^-22) and ONE cubic correction step in float64 (five instructions): ~1 ulp;
//   * exp(2 pi i t): quadrant q = round(4 t) and x = (pi / 2)(4 t - q) in [-pi/4, pi/4] from the integer bits (one float64
//     subtraction) -- the same x, bit for bit, as the restatement's 2 pi (t - q / 4) -- then the fdlibm kernel polynomials
//     (each < 1 ulp) and the quadrant's swap / signs by three integer instructions on the angle word.
// Everything but the seed is plain IEEE float64 arithmetic with FMAs, so the host emulation (emu_gen64.cpp, tests/
// test_emu_gen64.py) executes the kernels' arithmetic exactly; draws agree with the libm restatement to ~4e-16 relative
// (bar in the tests: 2e-14 absolute).
#pragma once
#include "fmc_core.h"
#include <string.h>
#include <math.h>

namespace fmc {

constexpr int GEN64_LOG_ENTRIES = 128;
struct Gen64Entry {
  double c2;   // -2 c_j
  double T;    // 2 ln c_j
};
constexpr size_t GEN64_TABLE_BYTES = GEN64_LOG_ENTRIES * sizeof(Gen64Entry);   // 2 KB

FMC_HD uint32_t g64_hi(double x) { uint64_t b; memcpy(&b, &x, 8); return (uint32_t)(b >> 32); }
FMC_HD uint32_t g64_lo(double x) { uint64_t b; memcpy(&b, &x, 8); return (uint32_t)b; }
FMC_HD double g64_mk(uint32_t hi, uint32_t lo) { const uint64_t b = ((uint64_t)hi << 32) | lo; double x; memcpy(&x, &b, 8); return x; }
FMC_HD double g64_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// c_j of the table: the reciprocal of the centre of interval j of the reduced hi word, rounded to float32 (any value near it
// serves: the table carries ln of the value actually stored)
inline double gen64_table_c(int j) {
  if (j == 63 || j == 64) return 1.0;          // m in [1 - 2^-8, 1) and [1, 1 + 2^-7): r = m - 1 exactly
  const double lo = g64_mk((uint32_t)((j << 13) + 0x3FE80000), 0u), hi = g64_mk((uint32_t)(((j + 1) << 13) + 0x3FE80000), 0u);
  return (double)(float)(1.0 / (0.5 * (lo + hi)));
}
// the table (host side; uploaded once per device by fastmc.hip, built again by the emulator)
inline void gen64_build_table(Gen64Entry* t) {
  for (int j = 0; j < GEN64_LOG_ENTRIES; ++j) {
    const double c = gen64_table_c(j);
    t[j].c2 = -2.0 * c;
    t[j].T = (double)(2.0L * logl((long double)c));
  }
}

// 1 / sqrt(x) to ~2^-22: the hardware's float32 estimate (host emulation: a correctly rounded one; the two Newton steps
// that follow forget the difference)
FMC_HD float g64_rsq_seed(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / sqrtf(x);
#endif
}

// y = -2 ln((A + 1/2) 2^-53) for the 53-bit integer A = a 2^21 + (a2 >> 11); `tab` = the 128-entry table (LDS or global)
template <class TabPtr>
FMC_HD double g64_neg2log(uint32_t a, uint32_t a2, TabPtr tab) {
  // 2 A + 1 = a 2^22 + ((a2 >> 10) | 1) in ONE FMA: a single rounding of the exact 54-bit odd integer, i.e. twice the
  // restatement's float64(A) + 0.5 (A >= 2^52 rounds to even there as here)
  const double v = g64_fma((double)a, 0x1p22, (double)((a2 >> 10) | 1u));
  const uint32_t hx = g64_hi(v) + 0x00080000u;            // mantissas >= 1.5 carry into the exponent: m in [0.75, 1.5)
  const int K = (int)(hx >> 20) - (1023 + 54);              // u = v 2^-54 = 2^K m
  const double m = g64_mk((hx & 0x000FFFFFu) + 0x3FE80000u, g64_lo(v));
  const Gen64Entry e = tab[(hx >> 13) & 0x7Fu];
  const double r = g64_fma(m, e.c2, 2.0);                   // r' = -2 (m c_j - 1)
  double q = g64_fma(r, 0x1.2199e38a9b961p-9, 0x1.5554fa10ca076p-8);
  q = g64_fma(r, q, 0x1.99999dcf7fe94p-7);
  q = g64_fma(r, q, 0x1.00000000a1389p-5);
  q = g64_fma(r, q, 0x1.5555555554aa9p-4);
  q = g64_fma(r, q, 0x1.fffffffffffffp-3);
  const double p = g64_fma(r * r, q, r);                    // -2 ln(1 + r)
  return g64_fma((double)K, -0x1.62e42fefa39efp+0, e.T) + p;      // K (-2 ln 2) + 2 ln c_j + p
}

// sqrt(y), y >= 0 (y = 0 when u rounds to 1: probability 2^-53)
FMC_HD double g64_sqrt(double y) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(FMC_G64_RSQ64)     // A/B: the float64 estimate instruction instead of cvt + v_rsq_f32 + cvt
  const double s = __builtin_amdgcn_rsq(y > 1.0e-300 ? y : 1.0e-300);
#else
  const float yf = (float)y;
  const double s = (double)g64_rsq_seed(yf > 1.0e-30f ? yf : 1.0e-30f);
#endif
  // ONE cubic step from the seed: g = y s = sqrt(y) (1 + d), e = 1 - g s = -(2 d + d^2) exactly enough (|d| < 2^-21), and
  // sqrt(y) = g / sqrt(1 - e) = g (1 + e / 2 + 3 e^2 / 8 + O(e^3)), O(e^3) < 2^-62
  const double g = y * s;
  const double e = g64_fma(-g, s, 1.0);
  return g64_fma(g, e * g64_fma(e, 0.375, 0.5), g);
}

FMC_HD uint32_t g64_alignbit(uint32_t hi, uint32_t lo, int sh) {      // ({hi, lo} >> sh) & 0xFFFFFFFF, 0 < sh < 32
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
  return (hi << (32 - sh)) | (lo >> sh);
#endif
}
// a ^ (b & c) in one v_bitop3_b32 (truth table 0xF0 ^ (0xCC & 0xAA))
FMC_HD uint32_t g64_xor_and(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FMC_NO_BITOP3)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x78);
#else
  return a ^ (b & c);
#endif
}

// (cos, sin)(2 pi t) scaled by R, t = B 2^-53, B = (b >> 9) 2^30 + (b2 >> 2).  In quarter turns 4 t = e / 2 + ..., e = b >> 29:
// the quadrant q = (e + 1) >> 1 (round to nearest) and the signed remainder f = 4 t - q in [-1/2, 1/2) come from the integer
// bits -- G = B mod 2^51 with bit 50 flipped, placed in the mantissa of 2^52, minus (2^52 + 2^50) is f 2^51 EXACTLY: one
// float64 subtraction instead of two conversions, an ldexp, an FMA, a rint and a conversion back -- and
//     swap sin / cos  <=>  q odd       <=>  bit 30 of b + 2^29
//     sin negative    <=>  q in {2, 3} <=>  bit 31 of b + 2^29
//     cos negative    <=>  q in {1, 2} <=>  bit 31 of b + 3 2^29      (all mod 2^32: e = 7 rounds up to q = 4 = 0)
// x = (pi / 2) f is the restatement's 2 pi (t - rint(4 t) / 4) bit for bit, except on the ties of the quadrant rounding
// (probability 2^-51: rint rounds them to even, this rounds them up; either is the same angle).
FMC_HD void g64_sincos_scaled(uint32_t b, uint32_t b2, double R, double& re, double& im) {
  // f 2^51 = G - (bit 50 of G) 2^51 = (G xor 2^50) - 2^50: the xor and the exponent of 2^52 are ONE constant on the hi word
  const uint32_t glo = g64_alignbit(b >> 9, b2, 2);                              // ((b >> 9) << 30) | (b2 >> 2)
  const double d = g64_mk(((b >> 11) & 0x0007FFFFu) ^ 0x43340000u, glo);         // 2^52 + (G xor 2^50)
  const double x = (d - 0x1.4p+52) * 0x1.921fb54442d18p-51;                      // (pi / 2) 2^-51 (f 2^51)
  const double z = x * x;
  // fdlibm __kernel_sin / __kernel_cos on [-pi/4, pi/4]
  double ps = g64_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = g64_fma(z, ps, 2.75573137070700676789e-06);
  ps = g64_fma(z, ps, -1.98412698298579493134e-04);
  ps = g64_fma(z, ps, 8.33333333332248946124e-03);
  ps = g64_fma(z, ps, -1.66666666666666324348e-01);
  const double sn = g64_fma(z * x, ps, x);
  double pc = g64_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = g64_fma(z, pc, -2.75573143513906633035e-07);
  pc = g64_fma(z, pc, 2.48015872894767294178e-05);
  pc = g64_fma(z, pc, -1.38888888888741095749e-03);
  pc = g64_fma(z, pc, 4.16666666666666019037e-02);
  const double cs = g64_fma(z * z, pc, g64_fma(z, -0.5, 1.0));
  const uint32_t q1 = b + 0x20000000u, q3 = b + 0x60000000u;
  const bool swap = (q1 & 0x40000000u) != 0u;
  const double cv = swap ? sn : cs, sv = swap ? cs : sn;
  // cos(2 pi t) = {cs, -sn, -cs, sn}[q & 3],  sin(2 pi t) = {sn, cs, -sn, -cs}[q & 3]: the signs go onto R's hi word
  const uint32_t rh = g64_hi(R), rl = g64_lo(R);
  re = g64_mk(g64_xor_and(rh, q3, 0x80000000u), rl) * cv;
  im = g64_mk(g64_xor_and(rh, q1, 0x80000000u), rl) * sv;
}

// One coloured coefficient: sqrt(-2 ln u) exp(2 pi i t) amp
template <class TabPtr>
FMC_HD void box_muller_f64_fast(uint32_t a, uint32_t b, uint32_t a2, uint32_t b2, double amp, TabPtr tab, double& re, double& im) {
  const double R = g64_sqrt(g64_neg2log(a, a2, tab)) * amp;
  g64_sincos_scaled(b, b2, R, re, im);
}

}  // namespace fmc
