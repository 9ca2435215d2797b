// fmc_gen64.h -- the device generator at the reference's precision (GPU_RNG_PRECISION 'f64', the default), fast form.
//
// The reference draws 53-bit normals and colours them in float64 (fast/funcs.py:352-356, fast/fast.py:593-594).  Our float64
// generator turns FOUR 32-bit words into one complex normal (definition: box_muller_f64 below, libm; restated in
// oracle/devrng.py: box_muller_f64):
//     u = RNE(a 2^32 + (a2 | 1)) 2^-64,     t = ((b >> 8) 2^32 + b2) 2^-56 turns,     sqrt(-2 ln u) exp(2 pi i t)
// -- a uniform with 53 significant bits at every magnitude down to 2^-64 (tails to 9.4 sigma), a 56-bit angle; their leading 32 /
// 24 bits are the words (a, b >> 8) of the float32 draw, so the two precisions see the same normals to ~2^-24.  The coefficient
// streams take the four words from ONE xoshiro128+ state (fmc_core.h: xoshiro128p::next4); the log-amplitude and sub-harmonic
// draws from Philox blocks.  (Round 4's definition spliced 53 + 53 bits out of two streams: 14 more integer instructions per
// coefficient and a second Philox block per lane and row.)
//
// This header is that definition in ~60 instructions per coefficient (30 float64, 5 conversions, ~24 integer, 1 v_rsq_f32),
// cheap enough to be fused into the row kernels (fmc_kernels.h: MODE 2), so that no coefficient ever passes through HBM:
//   * y = -2 ln u: u = 2^K m with m in [0.75, 1.5) by integer arithmetic on the hi word; the top seven mantissa bits of the
//     reduced hi word index a 128-entry table (-2 c_j, 2 ln c_j), c_j ~ 1 / centre of interval j (exactly 1 for the interval
//     that ends at m = 1, so that u -> 1 keeps full RELATIVE accuracy); r' = fma(m, -2 c_j, 2) = -2 (m c_j - 1) is exact up to
//     one rounding, |r'| <= 2^-7, and  -2 ln(1 + r) = r' + r'^2 Q(r')  with a degree-4 near-minimax Q
//     (tools/gen64_design.py: relative error of y < 4e-17 before rounding);
//   * sqrt y: v_rsq_f32 seed (2^-22) and ONE cubic correction step in float64 (five instructions): ~1 ulp;
//   * exp(2 pi i t): the top byte of b indexes a 256-entry table (cos, sin)(2 pi (j + 1/2) / 256); the other 48 bits G = (bits 8 ...
//     23 of b, b2), placed in the mantissa of 2^52 by ONE v_perm_b32 and reduced by (2^52 + 2^47), are the remainder EXACTLY
//     (one float64 subtraction, no conversion): x = 2 pi 2^-56 (G - 2^47) in [-pi/256, pi/256), sin x and cos x - 1 from two- and
//     three-term polynomials in w = G - 2^47 (the scale folded into the coefficients), and the table entry (scaled by the
//     radius) is rotated by x: fourteen float64 instructions, two integer ones.
// Everything but the seed is plain IEEE float64 arithmetic with FMAs, so the host emulation (emu_gen64.cpp, tests/
// test_emu_gen64.py) executes the kernels' arithmetic exactly; draws agree with the libm restatement to ~4e-16 relative
// (bar in the tests: 2e-14 absolute).
#pragma once
#include "fmc_core.h"
#include <string.h>
#include <math.h>

namespace fmc {

struct alignas(16) Gen64Entry {
  double c2;   // log table: -2 c_j          trig table: cos
  double T;    //            2 ln c_j                    sin
};
// entries [0, 256): (cos, sin)(2 pi (j + 1/2) / 256) -- FIRST, so that in the LDS (tables at address 0) the byte offset of a trig
// entry is 16 (b >> 24), one SDWA shift of the word's top byte; [256, 384): the log table
constexpr int GEN64_TRIG_ENTRIES = 256;
constexpr int GEN64_LOG_ENTRIES = 128;
constexpr int GEN64_LOG_BASE = GEN64_TRIG_ENTRIES;
constexpr size_t GEN64_TABLE_BYTES = (GEN64_LOG_ENTRIES + GEN64_TRIG_ENTRIES) * sizeof(Gen64Entry);   // 6 KB

// the two 32-bit halves of a float64 (device: a register pair taken apart / put together, no arithmetic -- hipcc turns the
// 64-bit shift-and-or form into byte-wise masks and ors)
#if defined(__HIPCC__)
typedef uint32_t g64_u32x2 __attribute__((ext_vector_type(2)));
FMC_HD uint32_t g64_hi(double x) { return __builtin_bit_cast(g64_u32x2, x).y; }
FMC_HD uint32_t g64_lo(double x) { return __builtin_bit_cast(g64_u32x2, x).x; }
FMC_HD double g64_mk(uint32_t hi, uint32_t lo) { g64_u32x2 v; v.x = lo; v.y = hi; return __builtin_bit_cast(double, v); }
#else
FMC_HD uint32_t g64_hi(double x) { uint64_t b; memcpy(&b, &x, 8); return (uint32_t)(b >> 32); }
FMC_HD uint32_t g64_lo(double x) { uint64_t b; memcpy(&b, &x, 8); return (uint32_t)b; }
FMC_HD double g64_mk(uint32_t hi, uint32_t lo) { const uint64_t b = ((uint64_t)hi << 32) | lo; double x; memcpy(&x, &b, 8); return x; }
#endif
FMC_HD double g64_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// -2 ln(1 + r) = r' + r'^2 Q(r'),  r' = -2 r,  |r'| <= 2^-7: degree-4 near-minimax Q (tools/gen64_design.py)
FMC_HD double g64_log_poly(double r) {
  double q = g64_fma(r, 0x1.55596cb774b43p-8, 0x1.999b05e405b8dp-7);
  q = g64_fma(r, q, 0x1.fffffffd90478p-6);
  q = g64_fma(r, q, 0x1.5555555527f95p-4);
  q = g64_fma(r, q, 0x1.0000000000001p-2);
  return g64_fma(r * r, q, r);
}

// c_j of the log table: the reciprocal of the centre of interval j of the reduced hi word, rounded to float32 (any value near it
// serves: the table carries ln of the value actually stored)
inline double gen64_table_c(int j) {
  if (j == 63) return 1.0;                     // m in [1 - 2^-8, 1): r = m - 1 exactly, K = 0, T = 0 -- y = p
  const double lo = g64_mk((uint32_t)((j << 13) + 0x3FE80000), 0u), hi = g64_mk((uint32_t)(((j + 1) << 13) + 0x3FE80000), 0u);
  return (double)(float)(1.0 / (0.5 * (lo + hi)));
}
// the tables (host side; uploaded once per device by fastmc.hip, built again by the emulator)
inline void gen64_build_table(Gen64Entry* t) {
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int j = 0; j < GEN64_TRIG_ENTRIES; ++j) {
    const long double a = two_pi * ((long double)(2 * j + 1) / (long double)(2 * GEN64_TRIG_ENTRIES));
    t[j].c2 = (double)cosl(a);
    t[j].T = (double)sinl(a);
  }
  for (int j = 0; j < GEN64_LOG_ENTRIES; ++j) {
    const double c = gen64_table_c(j);
    t[GEN64_LOG_BASE + j].c2 = -2.0 * c;
    t[GEN64_LOG_BASE + j].T = (double)(2.0L * logl((long double)c));
  }
  // m = 1 in interval 64 (the uniform that rounded up to 1): T := -p(m = 1) -- within an ulp of 2 ln c_64 -- makes y exactly 0
  t[GEN64_LOG_BASE + 64].T = -g64_log_poly(g64_fma(1.0, t[GEN64_LOG_BASE + 64].c2, 2.0));
}

// ---- where the tables are.  A plain pointer (global memory, host memory, or a generic pointer into the LDS) ...
FMC_HD Gen64Entry g64_trig_entry(const Gen64Entry* tab, uint32_t b) { return tab[b >> 24]; }
FMC_HD Gen64Entry g64_log_entry(const Gen64Entry* tab, uint32_t hx) { return tab[GEN64_LOG_BASE + ((hx >> 13) & 0x7Fu)]; }
#if defined(__HIPCC__)
// ... or the LDS at address 0 (the row kernels stage them there first thing: gen64_lds0_check): an entry's byte offset IS its
// address, so a look-up is one or two integer instructions and a ds_read_b128 with no base to add.
struct Gen64Lds0 {};
__device__ __forceinline__ Gen64Entry g64_lds_at(uint32_t byte_off) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) const d2* P;
  const d2 v = *(P)(uintptr_t)byte_off;              // one ds_read_b128
  Gen64Entry e;
  e.c2 = v.x; e.T = v.y;
  return e;
}
__device__ __forceinline__ Gen64Entry g64_trig_entry(Gen64Lds0, uint32_t b) {

  uint32_t off;      // 16 (b >> 24): the top byte selected by SDWA, shifted by four
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off) : "v"(4u), "v"(b));
  return g64_lds_at(off);
}
__device__ __forceinline__ Gen64Entry g64_log_entry(Gen64Lds0, uint32_t hx) {
  return g64_lds_at(((hx >> 9) & 0x7F0u) + (uint32_t)(GEN64_LOG_BASE * sizeof(Gen64Entry)));
}
// the kernels that use Gen64Lds0 carve their dynamic LDS with the tables first and have no static LDS: address 0
__device__ __forceinline__ void gen64_lds0_check(const void* s_tab) {
  typedef __attribute__((address_space(3))) const void* P;
  if ((uint32_t)(uintptr_t)(P)s_tab != 0u) __builtin_trap();
}
#endif

// 1 / sqrt(x) to ~2^-22: the hardware's float32 estimate (host emulation: a correctly rounded one; the cubic step
// that follows forgets the difference)
FMC_HD float g64_rsq_seed(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / sqrtf(x);
#endif
}

// y = -2 ln(V 2^-64), V = RNE(a 2^32 + (a2 | 1)) -- odd before rounding, so never 0 -- is evaluated in g64_pre / g64_post below:
// u = 2^K m, table entry by the top mantissa bits, r' = fma(m, -2 c_j, 2), y = K (-2 ln 2) + 2 ln c_j + g64_log_poly(r').
// V = 2^64 (a = 2^32 - 1 and a2 >= 2^32 - 2^10: probability 2^-54) reduces to K = 0, m = 1, interval 64, whose T the table sets
// to MINUS the polynomial's value at m = 1 (gen64_build_table), so that y is exactly 0 there and never negative anywhere.
// hi word of m: (hx & 0xFFFFF) + 0x3FE80000 (hipcc, left alone, splits the mask in two and ORs a byte back in: three instructions)
FMC_HD uint32_t g64_m_hi(uint32_t hx) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
  asm("v_and_b32 %0, 0xfffff, %1\n\tv_add_u32 %0, 0x3fe80000, %0" : "=&v"(r) : "v"(hx));
  return r;
#else
  return (hx & 0x000FFFFFu) + 0x3FE80000u;
#endif
}
// sqrt(y), y >= 0
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
// 0x43300000 | ((b >> 8) & 0xFFFF): the hi word of 2^52 + G, G = ((b >> 8) & 0xFFFF) 2^32 + b2 -- bytes (0x43, 0x30, b.2, b.1)
FMC_HD uint32_t g64_angle_hi(uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;      // (as asm: hipcc takes the builtin apart and puts it together again in four instructions)
  asm("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "s"(0x43300000u), "v"(b), "v"(0x07060201u));      // (one SGPR per VOP3: the selector in a VGPR)
  return r;
#else
  return 0x43300000u | ((b >> 8) & 0xFFFFu);
#endif
}

// (cos, sin)(2 pi t) scaled by R, t = ((b >> 8) 2^32 + b2) 2^-56.  The top byte of b picks the table angle
// theta_j = 2 pi (j + 1/2) / 256; the other 48 bits G give w = G - 2^47 exactly and x = 2 pi 2^-56 w in [-pi/256, pi/256):
//   sin x = w (k1 + z (k3 + z k5)),   cos x - 1 = z (c2 + z (c4 + z c6)),   z = w^2     (truncation < 1e-17; tools/gen64_design.py)
// then the rotation of the table entry (scaled by R first).
FMC_HD void g64_sincos_scaled(uint32_t b, uint32_t b2, double R, Gen64Entry e, double& re, double& im) {      // e = g64_trig_entry(tab, b)
  const double d = g64_mk(g64_angle_hi(b), b2);                                   // 2^52 + G
  const double w = d - 0x1.08p+52;                                                // G - 2^47
  const double z = w * w;
  double p = g64_fma(z, 0x1.466bc6775aae2p-274, -0x1.4abbce625be53p-163);
  p = g64_fma(z, p, 0x1.921fb54442d18p-54);
  const double sn = w * p;
  double q = g64_fma(z, -0x1.55d3c7e3cbffap-330, 0x1.03c1f081b5ac4p-218);
  q = g64_fma(z, q, -0x1.3bd3cc9be45dep-108);
  const double cm1 = z * q;                                                       // cos x - 1
  const double tc = R * e.c2, ts = R * e.T;                                      // R (cos, sin) theta_j
  re = g64_fma(-ts, sn, g64_fma(tc, cm1, tc));
  im = g64_fma(tc, sn, g64_fma(ts, cm1, ts));
}

// One coloured coefficient: sqrt(-2 ln u) exp(2 pi i t) amp, in two halves so that a row kernel can ask for draw j + 1's table
// entries before it does draw j's arithmetic (g64_pre: the words' integer work and the two table reads; g64_post: everything
// else).  `a2_odd`: the word a2 with its lowest bit set (the coefficient streams deliver it so: xoshiro128p::next4).
struct Gen64Pre {
  double v;            // a 2^32 + a2_odd, rounded once
  uint32_t hx;         // its hi word + 2^19
  uint32_t b, b2;
  Gen64Entry el, et;   // log entry, angle entry
};
template <class Tab>
FMC_HD Gen64Pre g64_pre(uint32_t a, uint32_t b, uint32_t a2_odd, uint32_t b2, Tab tab) {
  Gen64Pre p;
  p.et = g64_trig_entry(tab, b);            // asked for first: its address needs one instruction
  p.v = g64_fma((double)a, 0x1p32, (double)a2_odd);
  p.hx = g64_hi(p.v) + 0x00080000u;
  p.el = g64_log_entry(tab, p.hx);
  p.b = b; p.b2 = b2;
  return p;
}
FMC_HD void g64_post(const Gen64Pre& p, double amp, double& re, double& im) {
  const int K = (int)(p.hx >> 20) - (1023 + 64);
  const double m = g64_mk(g64_m_hi(p.hx), g64_lo(p.v));
  const double pl = g64_log_poly(g64_fma(m, p.el.c2, 2.0));
  const double y = g64_fma((double)K, -0x1.62e42fefa39efp+0, p.el.T) + pl;
  const double R = g64_sqrt(y) * amp;
  g64_sincos_scaled(p.b, p.b2, R, p.et, re, im);
}
template <class Tab>
FMC_HD void box_muller_f64_fast(uint32_t a, uint32_t b, uint32_t a2_odd, uint32_t b2, double amp, Tab tab, double& re, double& im) {
  g64_post(g64_pre(a, b, a2_odd, b2, tab), amp, re, im);
}

}  // namespace fmc
