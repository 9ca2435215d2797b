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
//   * exp(2 pi i t): the top seven bits of t index a 128-entry table (cos, sin)(2 pi (j + 1/2) / 128) (2 KB more); the
//     remainder x in [-pi/128, pi/128) comes EXACTLY out of the integer bits (one float64 subtraction), sin x and cos x - 1 from
//     three-term polynomials, and the table entry (scaled by the radius) is rotated by x: seventeen float64 instructions.
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
constexpr int GEN64_TRIG_ENTRIES = 128;
// entries [0, 128): the log table; [128, 256): (cos, sin)(2 pi (j + 1/2) / 128) in the same two-double layout
constexpr size_t GEN64_TABLE_BYTES = (GEN64_LOG_ENTRIES + GEN64_TRIG_ENTRIES) * sizeof(Gen64Entry);   // 4 KB

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
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int j = 0; j < GEN64_TRIG_ENTRIES; ++j) {
    // exact symmetries first (the table is symmetric about the octants), then long double
    const long double a = two_pi * ((long double)(2 * j + 1) / (long double)(2 * GEN64_TRIG_ENTRIES));
    t[GEN64_LOG_ENTRIES + j].c2 = (double)cosl(a);
    t[GEN64_LOG_ENTRIES + j].T = (double)sinl(a);
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

// (cos, sin)(2 pi t) scaled by R, t = B 2^-53, B = (b >> 9) 2^30 + (b2 >> 2).  The top seven bits of B pick the table angle
// theta_j = 2 pi (j + 1/2) / 128; the low 46 bits G, placed in the mantissa of 2^52 and reduced by (2^52 + 2^45), are
// (t - (j + 1/2) / 128) 2^53 EXACTLY (one float64 subtraction, no conversion), so x = 2 pi (t - theta_j / 2 pi) lies in
// [-pi/128, pi/128) and  sin x = x + x^3 (S1 + S2 x^2 + S3 x^4),  cos x - 1 = x^2 (C1 + C2 x^2 + C3 x^4)  to < 1e-17; then the
// rotation of the table entry (scaled by R first).  Seventeen float64 instructions and one 16-byte table read: no quadrant
// selects, no sign logic (round 4b; the quadrant form with the two fdlibm polynomials cost twenty and eleven integer ones).
template <class TabPtr>
FMC_HD void g64_sincos_scaled(uint32_t b, uint32_t b2, double R, TabPtr tab, double& re, double& im) {
  const uint32_t glo = g64_alignbit(b >> 9, b2, 2);                              // ((b >> 9) << 30) | (b2 >> 2)
  const double d = g64_mk(0x43300000u | ((b >> 11) & 0x00003FFFu), glo);         // 2^52 + G,  G = B mod 2^46
  const double x = (d - 0x1.02p+52) * 0x1.921fb54442d18p-51;                    // ((G - 2^45) 2^-53) 2 pi   (2 pi 2^-53 = (pi / 2) 2^-51)
  const Gen64Entry e = tab[GEN64_LOG_ENTRIES + (b >> 25)];
  const double z = x * x;
  double p = g64_fma(z, -1.0 / 5040.0, 1.0 / 120.0);
  p = g64_fma(z, p, -1.0 / 6.0);
  const double sn = g64_fma(x * z, p, x);
  double q = g64_fma(z, -1.0 / 720.0, 1.0 / 24.0);
  q = g64_fma(z, q, -0.5);
  const double cm1 = z * q;                                                       // cos x - 1
  const double tc = R * e.c2, ts = R * e.T;                                      // R (cos, sin) theta_j
  re = g64_fma(-ts, sn, g64_fma(tc, cm1, tc));
  im = g64_fma(tc, sn, g64_fma(ts, cm1, ts));
}

// One coloured coefficient: sqrt(-2 ln u) exp(2 pi i t) amp
template <class TabPtr>
FMC_HD void box_muller_f64_fast(uint32_t a, uint32_t b, uint32_t a2, uint32_t b2, double amp, TabPtr tab, double& re, double& im) {
  const double R = g64_sqrt(g64_neg2log(a, a2, tab)) * amp;
  g64_sincos_scaled(b, b2, R, tab, re, im);
}

}  // namespace fmc
