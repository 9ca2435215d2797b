// fmc_kernels.h -- gfx950 kernels of the FAST Monte-Carlo hot path (included by fastmc.hip).
//
// Per realisation g (one complex N x N transform = two Monte-Carlo iterations):
//   rows kernel : draw/colour one spectrum row, N-point DFT along kx pruned to the Np window
//                 columns  ->  V[b][oi][ky]                       (x half of 4)
//   cols kernel : N-point DFT along ky of each window column pruned to the Np window rows,
//                 + sub-harmonics, W * exp(i phi) and the pixel sum  ->  partial[b][xi][4]
//                                                                 (rows 3-5, 5c)
//   finalize    : sum the Np column partials, log-amplitude, |.|^2 (rows 5, 5b)
// Two kernel families: "wave" (N = 64 P, P = 2^k times 1, 3, 5, 7, 9, <= 32, i.e. 128 ... 2048; fmc_wavefft.h)
// and "direct" (any N <= 4096).
#pragma once
#include <hip/hip_runtime.h>
#ifndef FMC_TU
#define FMC_TU 0      // translation unit of the split build (fastmc.hip): the kernels that are not templates live in unit 0 only
#endif
#include "fmc_core.h"
#include "fmc_wavefft.h"
#include "fmc_bluestein.h"
#include "fmc_mrfft.h"
#include "fmc_gen64.h"

namespace fmc {

// ------------------------------------------------------------------ device RNG
// Philox rounds of the block that seeds a coefficient stream.  7 is the smallest count for which Philox4x32 is
// Crush-resistant (Salmon et al., SC'11, table 2; 10 is its safety-margin default, kept for the log-amplitude and
// sub-harmonic draws, which use Philox words directly): here the block only seeds a 16- to 64-step xoshiro128+ stream.
#define FMC_SEED_ROUNDS 7
// Box-Muller on two 32-bit words:  r = sqrt(-2 ln U), U = (x0 + 0.5) 2^-32;  theta = 2 pi (x1 >> 9) 2^-23
// -> (r cos theta, r sin theta): a standard complex normal.  float32 hardware transcendentals
// (v_log_f32 = log2, v_sqrt_f32, v_sin_f32 / v_cos_f32 take turns); the oracle restates the
// same formula in float64 (oracle/devrng.py) and the two agree to ~1e-6 absolute.
// The angle as a float in [1, 2) built from the top 23 bits of the word by ONE v_alignbit (v_sin_f32 / v_cos_f32 take
// turns and have period 1): theta / 2 pi = 1 + (x1 >> 9) 2^-23.
__device__ __forceinline__ float angle_turns(uint32_t x1) {
  return __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, x1, 9));     // (0x7F << 23) | (x1 >> 9)
}
// K_BM = sqrt(2 ln 2): r = K_BM sqrt(-log2 u).  draw_coloured folds K_BM into the colouring table instead.
#define FMC_K_BM 1.1774100225154747f
__device__ __forceinline__ void box_muller(uint32_t x0, uint32_t x1, float& re, float& im) {
  const float u = fmaf((float)x0, 2.3283064365386963e-10f, 1.1641532182693481e-10f);  // (x0 + .5) 2^-32
  const float t = angle_turns(x1);
  const float r = FMC_K_BM * __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(u));            // sqrt(-2 ln u)
  re = r * __builtin_amdgcn_cosf(t);
  im = r * __builtin_amdgcn_sinf(t);
}
// The same draw scaled by `ampk` = amp * K_BM (the constant of the radius folded into the colouring table):
// cvt, fma, log, sqrt, alignbit, cos, sin and three multiplies.
__device__ __forceinline__ void box_muller_scaled(uint32_t x0, uint32_t x1, float ampk, float& re, float& im) {
  const float u = fmaf((float)x0, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
  const float t = angle_turns(x1);
  const float ra = __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(u)) * ampk;
  re = ra * __builtin_amdgcn_cosf(t);
  im = ra * __builtin_amdgcn_sinf(t);
}

struct RngKey {
  uint32_t k0, k1;   // seed
};

// Coefficient stream of (realisation g, row ky, stream L = kx mod SL), SL = 64 * spec_split(N): xoshiro128+
// seeded with one Philox block; its (2j)-th and (2j+1)-th words make coefficient (ky, L + SL j).
__device__ __forceinline__ xoshiro128p row_stream(RngKey key, uint64_t g, int ky, int L, int SL) {
  xoshiro128p s;
  s.seed(philox4x32<FMC_SEED_ROUNDS>((uint32_t)(ky * SL + L), STREAM_SCREEN, (uint32_t)g, (uint32_t)(g >> 32), key.k0, key.k1));
  return s;
}
// ---- the generator at the reference's precision (GPU_RNG_PRECISION 'f64', the default; fast/funcs.py:352-356 draws 53-bit
// normals).  Four words make one complex normal (fmc_gen64.h has the definition and its fast form):
//     u = RNE(a 2^32 + (a2 | 1)) 2^-64,   turns t = ((b >> 8) 2^32 + b2) 2^-56,   (re, im) = sqrt(-2 ln u) (cos, sin)(2 pi t)
// in float64.  (a, b) are the words of the float32 draw, whose u and t are this draw's cut to their first 24 / 23 bits: the two
// precisions see the same normals to ~2^-24.  The coefficient streams take (a2, b2) from the SAME xoshiro128+ state as (a, b)
// (xoshiro128p::next4: a 24-bit multiply-add on both halves of the state) -- round 4 ran a second stream per lane for them.  Tails
// reach 9.4 sigma.
constexpr uint32_t STREAM_SUBHARM_LO = 4;
// the definition, with libm (log-amplitude and sub-harmonic draws: a handful per realisation; the coefficient draws run
// box_muller_f64_fast, which agrees with this to ~4e-16)
__device__ __forceinline__ void box_muller_f64(uint32_t a, uint32_t b, uint32_t a2, uint32_t b2, double& re, double& im) {
  const double u = __builtin_fma((double)a, 0x1p32, (double)(a2 | 1u)) * 0x1p-64;
  const double r = sqrt(-2.0 * log(u));
  // the 56-bit angle reduced EXACTLY to the nearest quarter turn: |rem| <= 2^53 is a float64, x carries one rounding
  const uint64_t T = ((uint64_t)(b >> 8) << 32) | (uint64_t)b2;
  const uint64_t q = (T + (1ull << 53)) >> 54;
  const double x = (double)(int64_t)(T - (q << 54)) * 0x1.921fb54442d18p-54;      // 2 pi 2^-56 rem
  double sn, cs;
  sincos(x, &sn, &cs);
  const int qi = (int)(q & 3);
  const double c = qi == 0 ? cs : (qi == 1 ? -sn : (qi == 2 ? -cs : sn));
  const double si = qi == 0 ? sn : (qi == 1 ? cs : (qi == 2 ? -sn : -cs));
  re = r * c;
  im = r * si;
}

// The same draw in ~60 instructions (fmc_gen64.h): table-driven log, v_rsq_f32-seeded cubic sqrt, table + rotation for the
// angle -- what the row kernels run in MODE 2, the staging kernel and the read-back.  `tab`: the tables as a pointer (global
// memory, or generic into the LDS) or Gen64Lds0 (staged at LDS address 0: the row kernels).
template <class Tab>
__device__ __forceinline__ cpx<double> draw_coloured_f64(xoshiro128p& s, double amp, Tab tab) {
  uint32_t a, b, a2, b2;
  s.next4(a, b, a2, b2);
  cpx<double> c;
  box_muller_f64_fast(a, b, a2, b2, amp, tab, c.x, c.y);
  return c;
}
// stage the tables of the float64 generator into the LDS (4 KB; the caller's barrier follows)
__device__ __forceinline__ void load_gen64_table(Gen64Entry* s_tab, const Gen64Entry* g) {
  double* d = reinterpret_cast<double*>(s_tab);
  const double* sgl = reinterpret_cast<const double*>(g);
  for (int i = threadIdx.x; i < (int)(GEN64_TABLE_BYTES / 8); i += blockDim.x) d[i] = sgl[i];
}

// both Box-Muller words of a coefficient from ONE state advance (9 instead of 16 integer operations)
__device__ __forceinline__ void draw_words(xoshiro128p& s, uint32_t& a, uint32_t& b) { s.next2(a, b); }
template <class R>
__device__ __forceinline__ cpx<R> draw_coeff(xoshiro128p& s) {
  uint32_t a, b;
  draw_words(s, a, b);
  float re, im;
  box_muller(a, b, re, im);
  return mk<R>((R)re, (R)im);
}
// The coloured coefficient c * amp.  The normals are float32 (hardware Box-Muller), so the colouring multiply is done
// in float32 too (amp as a float table; one rounding to 24 bits, like the normals themselves) and the product is
// widened: float32 multiplies instead of 2 v_mul_f64, half the table bytes; the table carries the radius constant
// sqrt(2 ln 2) of Box-Muller, so a coefficient costs three multiplies.  Host-coefficient (parity) mode multiplies
// float64 by float64 as the reference does (fast/fast.py:594).
template <class R>
__device__ __forceinline__ cpx<R> draw_coloured(xoshiro128p& s, float ampk) {   // ampk = amp * K_BM (k_make_amp)
  uint32_t a, b;
  draw_words(s, a, b);
  float re, im;
  box_muller_scaled(a, b, ampk, re, im);
  return mk<R>((R)re, (R)im);
}

__device__ __forceinline__ float draw_logamp_normal(RngKey key, uint64_t iter) {
  const u32x4 x = philox4x32_10(0u, STREAM_LOGAMP, (uint32_t)iter, (uint32_t)(iter >> 32), key.k0, key.k1);
  float a, b;
  box_muller(x.a, x.b, a, b);
  return a;
}
// the same draw with all four words of the block: words a, b lead (as in the float32 draw), c, d supply the low bits
__device__ __forceinline__ double draw_logamp_normal_f64(RngKey key, uint64_t iter) {
  const u32x4 x = philox4x32_10(0u, STREAM_LOGAMP, (uint32_t)iter, (uint32_t)(iter >> 32), key.k0, key.k1);
  double a, b;
  box_muller_f64(x.a, x.b, x.c, x.d, a, b);
  return a;
}

// ------------------------------------------------------------------ shared parameter blocks
// Tables of the chirp-z kernels (fmc_bluestein.h), all in global memory except twf (staged into LDS).
template <class R>
struct BluArgs {
  const cpx<R>* twf;            // [64]   w_64^{l0 b0}
  const cpx<R>* pre;            // [M]    input chirp (zero beyond N)
  const cpx<R>* vhat;           // [SB][M] DFT_M of the chirp kernel (per input block)
  const cpx<R>* post;           // [omS]  output chirp / M
  int SB, B;                    // input blocks per row and their length (SB = 1: the whole row in one transform)
};

template <class R>
struct RowArgs {
  int N, Np, lo, nb;            // grid size, window size, first window index, realisations in this launch
  int tiles = 0;                // k_rows_wave: tiles of the launch when its workgroups walk them (0: one tile per workgroup)
  int rpw = 0;                  // k_rows_wave / _mr / _blu: rows per wave of this launch (0: ROWS_PER_WAVE); small launches take fewer (pick_rpw)
  int S = 0;                    // k_rows_pks with a RUN-TIME sub-row count (template S <= 0: the grids beyond 1792): the count
  const R* amp;                 // [N][N] sqrt(powerspec)*df  (wave family: with (-1)^(ky+kx) folded in); host-coefficient mode
  const float* ampf;            // the same table rounded to float32: colouring of the device generator's float32 normals
  const cpx<R>* tw;             // wave: tw1 [P*64];  direct: w_N^e, e < N
  const cpx<R>* om;             // wave: [8][omS]
  int omS;
  const cpx<R>* cw;             // split rows (S > 1): [S][omS]  w_N^{s (lo + oi)}
  int tw_global;                // direct family: twiddles read from global memory (the LDS holds the row only)
  cpx<R>* V;                    // [nb][Np][N]  (window column major)
  RngKey key;
  uint64_t g0;                  // global index of realisation b = 0
  const double* cre;            // host-coefficient mode: [nb][N][N] real parts
  const double* cim;
  BluArgs<R> blu;               // chirp-z family only
  const Gen64Entry* g64;        // float64 generator (MODE 2): its tables (fmc_gen64.h)
  unsigned long long* clk = nullptr;   // k_rows_wave: four words for the launch's clock stamps (fastmc_last_clock), or null
};

struct SubharmArgs {
  int enabled;                  // 0 off, 1 general 27-mode sum, 2 separable grids: 9 column-folded terms (dcol)
  const double* dcol;           // [nb][Np][9][2]  d_{p,i}(b, x) = sum_j c_{p,i,j} ex_{p,j}(x)   (enabled == 2)
  const double* coef;           // [nb][27][2] coloured, df-scaled coefficients c_m
  const double* mean;           // [nb][2]     sum_m c_m mu_m
  const double* ex;             // [27][Np][2] exp(i x fx_m) on the window columns
  const double* ey;             // [27][Np][2] exp(i y fy_m) on the window rows
};

template <class R>
struct ColArgs {
  int N, Np, lo, nb;
  int S = 0;                    // k_cols_pks with a run-time sub-row count (see RowArgs)
  const cpx<R>* V;              // [nb][Np][N]
  const cpx<R>* tw;
  const cpx<R>* om;
  int omS;
  const cpx<R>* cw;             // split columns (S > 1): [S][omS]
  int tw_global;                // direct family: twiddles read from global memory
  const double* W;              // [Np][Np]
  SubharmArgs sh;
  double* partial;              // [nb][Np][4]  (EPI 0)
  double* phs;                  // [2][nb][Np][Np] (EPI 1): Re screens then Im screens of this launch
  BluArgs<R> blu;               // chirp-z family only
};

// phi -> contribution of one window pixel to the four sums; sub-harmonics added first.
template <class R>
__device__ __forceinline__ void pixel_phase(const SubharmArgs& sh, int b, int Np, int yi, int xi, R& p1, R& p2) {
  if (sh.enabled == 2) {
    // mode (p, i, j) has fx = fx_{p,j}, fy = fy_{p,i} (funcs.py:233-240, fast.py:835-844: a 3 x 3 meshgrid
    // per level), so the x-dependent factors are folded per (realisation, column) by k_subharm_cols and a
    // pixel needs 9 complex multiply-adds instead of 27 x 2
    double sr = -sh.mean[b * 2 + 0], si = -sh.mean[b * 2 + 1];
    const double* d = sh.dcol + ((size_t)b * Np + xi) * 18;
#pragma unroll
    for (int l = 0; l < 9; ++l) {
      const int m = 9 * (l / 3) + 3 * (l % 3);          // any j: ey depends on (p, i) only
      const double eyr = sh.ey[(m * Np + yi) * 2], eyi = sh.ey[(m * Np + yi) * 2 + 1];
      sr += d[2 * l] * eyr - d[2 * l + 1] * eyi;
      si += d[2 * l] * eyi + d[2 * l + 1] * eyr;
    }
    p1 = (R)((double)p1 + sr);
    p2 = (R)((double)p2 + si);
  } else if (sh.enabled) {
    double sr = -sh.mean[b * 2 + 0], si = -sh.mean[b * 2 + 1];
    const double* cf = sh.coef + (size_t)b * 54;
#pragma unroll 3
    for (int m = 0; m < 27; ++m) {
      const double exr = sh.ex[(m * Np + xi) * 2], exi = sh.ex[(m * Np + xi) * 2 + 1];
      const double eyr = sh.ey[(m * Np + yi) * 2], eyi = sh.ey[(m * Np + yi) * 2 + 1];
      const double er = exr * eyr - exi * eyi, ei = exr * eyi + exi * eyr;
      sr += cf[2 * m] * er - cf[2 * m + 1] * ei;
      si += cf[2 * m] * ei + cf[2 * m + 1] * er;
    }
    p1 = (R)((double)p1 + sr);
    p2 = (R)((double)p2 + si);
  }
}

// sin and cos of a phase in float64: Cody-Waite reduction by pi/2 (three-term, FMA) and the
// fdlibm kernel polynomials on [-pi/4, pi/4] (< 1 ulp).  Phases are tens of radians (|phi| < 100
// even without AO); beyond 1e5 rad the library routine with its Payne-Hanek path takes over.
// ~35 VALU instructions instead of the ~155 of ocml's sincos: the column kernel calls it four
// times per wavefront.
__device__ __forceinline__ void sincos_r(double x, double& s, double& c) {
  if (!(fabs(x) < 1.0e5)) { sincos(x, &s, &c); return; }
  const double kd = rint(x * 0.63661977236758134308);
  double r = fma(-kd, 1.5707963267948965580e+00, x);
  r = fma(-kd, 6.1232339957367660359e-17, r);
  r = fma(-kd, -1.4973849048591698329e-33, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int n = (int)kd & 3;
  const double a = (n & 1) ? cr : sr;
  const double b = (n & 1) ? sr : cr;
  s = (n & 2) ? -a : a;
  c = ((n + 1) & 2) ? -b : b;
}
__device__ __forceinline__ void sincos_r(float x, double& s, double& c) {
  float fs, fc;
  sincosf(x, &fs, &fc);
  s = fs; c = fc;
}

// One DPP-permuted copy of a double (two 32-bit movs; all rows and banks enabled).
template <int CTRL>
__device__ __forceinline__ double dpp_copy(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the 64 lanes, the same value in every lane, in a fixed order (deterministic).  Within a
// 16-lane row by DPP (quad permutes, half-row mirror, row mirror: register-file crossbar, no LDS), then
// the four row sums by v_readlane -- instead of six ds_bpermute round trips per value.
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_copy<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_copy<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_copy<0x141>(v);    // row_half_mirror
  v += dpp_copy<0x140>(v);    // row_mirror
  const long long b = __double_as_longlong(v);
  const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  double r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int l2 = __builtin_amdgcn_readlane(lo, 16 * q), h2 = __builtin_amdgcn_readlane(hi, 16 * q);
    r[q] = __longlong_as_double(((long long)h2 << 32) | (unsigned int)l2);
  }
  return (r[0] + r[1]) + (r[2] + r[3]);
}

// Detector of one window column (fast/fast.py:647-668) shared by the column kernels of the wave, chirp-z and 50-lane families:
// slot s of lane l holds the phase pair (p1, p2) of window row yi = l + 64 s for the Re and Im screens of realisation b.
// EPI 0: W exp(i phi) summed over the column -> partial[b][xi][4] (deterministic wave reduction); EPI 1: the screens themselves.
template <class R, int NS, int EPI>
__device__ __forceinline__ void column_epilogue(const SubharmArgs& sh, const double* W, double* partial, double* phs, int nb, int Np, int b,
                                                int xi, int lane, const R (&p1s)[NS], const R (&p2s)[NS]) {
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int yi = lane + WAVE * s;
    if (yi < Np) {
      R p1 = p1s[s], p2 = p2s[s];
      pixel_phase<R>(sh, b, Np, yi, xi, p1, p2);
      if (EPI == 1) {
        const size_t plane = (size_t)Np * Np;
        phs[((size_t)b) * plane + (size_t)yi * Np + xi] = (double)p1;
        phs[((size_t)(nb + b)) * plane + (size_t)yi * Np + xi] = (double)p2;
      } else {
        const double wgt = W[(size_t)yi * Np + xi];
        double s1, c1, s2, c2;
        sincos_r(p1, s1, c1);
        sincos_r(p2, s2, c2);
        acc[0] += wgt * c1; acc[1] += wgt * s1; acc[2] += wgt * c2; acc[3] += wgt * s2;
      }
    }
  }
  if (EPI == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = wave_sum(acc[q]);
    if (lane == 0) {
      double* o = partial + ((size_t)b * Np + xi) * 4;
      o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
    }
  }
}

// ================================================================== wave family
template <class R, int P, int NS>
struct GpuExec {
  int lane;
  LaneRegs<R, P, NS>& r;
  template <class F> __device__ __forceinline__ void each(F f) { f(lane, r); }
  // 8-byte LDS accesses kept as single ds_read_b64 / ds_write_b64: hipcc otherwise pairs them into
  // ds_read2_b64, which runs at half the bytes per clock (MI355X_MICROARCH.md, LDS table).
  static __device__ __forceinline__ double ld(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  static __device__ __forceinline__ cpx<float> ld(const cpx<float>* p) {
    const double d = __hip_atomic_load(reinterpret_cast<const double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    cpx<float> o;
    __builtin_memcpy(&o, &d, 8);
    return o;
  }
  static __device__ __forceinline__ cpx<double> ld(const cpx<double>* p) { return *p; }     // 16 bytes: one ds_read_b128
  // Stores are left to hipcc (ds_write2_b64 pairs).  Tried in round 3 for the 16-byte elements of the one-pass exchange 2: a
  // 16-byte aligned ds_write_b128 -- conflict-free where the pair has 2-way bank conflicts -- ran the rows kernel 7 % SLOWER
  // (9.35 against 8.70 ms per 5000 realisations), single ds_write_b64s the same as the pairs: the store form sets the time.
  template <class E> static __device__ __forceinline__ void st(E* p, E v) { *p = v; }
  // compiler-only barrier for memory operations: loads after it are not hoisted above it
  static __device__ __forceinline__ void loadfence() { asm volatile("" ::: "memory"); }
  // "this value is needed HERE": loads feeding it are issued before this point and waited for once.
  static __device__ __forceinline__ void pin(double& x) { asm volatile("" : "+v"(x)); }
  static __device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
  static __device__ __forceinline__ void pin(cpx<float>& x) { asm volatile("" : "+v"(x.x), "+v"(x.y)); }
  // Exchange-image hand-off between the lanes of ONE wavefront.  DS instructions of a wave are
  // issued and executed by the LDS in program order, so a read issued after a write (or a write
  // after a read) of the same wave needs no s_waitcnt: wavefront-scope fences only stop the
  // compiler from moving or caching LDS accesses across this point.
  __device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
};

constexpr int ROWS_PER_WAVE = 8;   // rows kernel: rows per wave (4 / 16 / 32 / 64 measured -4 % ... +-1 %)
// Waves per workgroup: the twiddle tables are staged once per workgroup, so bigger groups leave
// more LDS for exchange buffers: 12 waves = 3 per SIMD at 132 VGPRs (f64, P = 16).
// P = 32 keeps 2 x 32 values per lane (>= 200 VGPRs) and 18 KiB of exchange buffer per wave: 6 waves
// (A/B at 2048^2 f64: 6 waves 141k it/s vs 4 waves 114k it/s; 8 do not fit the LDS).
// Windows of 129-256 pixels (NS = 4, P = 8, 16, 32) keep the NS = 2 configuration.
// The general-window instantiation (NS = P) carries a large `om` table: 4 waves.  P = 10, 12, 20, 24
// (3*2^k, 5*2^k): 8 waves (A/B at 640^2 / 768^2: +4 % / +1 % over 12, no spill at the 168-VGPR step);
// P = 18, 28 (radix-9 / radix-7 stage with many live temporaries): 4 waves, one per SIMD, no spill.
template <class R, int P, int NS> struct WaveCfg {
  static constexpr int WPB = (NS > 4 || (NS > 2 && NS == P)) ? 4 : (NS == 4 && P == 24) ? 6 :   // 256-pixel window tables: 6 waves fit the LDS
                              (P == 32 ? 6 : ((P > 16 && P / (P & -P) >= 7) ? 4 : (P > 24 ? 6 : ((P > 16 || (!is_pow2(P) && P > 8)) ? 8 : 12))));
  // (rows of the small grids, P <= 8: 8 / 10 / 16 waves per workgroup = 4 / 5 / 4 per SIMD measured within 1 % of 12 at
  // 128^2 ... 512^2: those rows are LDS-bound, not latency-bound)
  static constexpr int WPB_ROWS = WPB;
};

// Row / column variant D (what fastmc.hip:dispatch_wave picks by window):
//   0  the general row of fmc_wavefft.h (pruned_row_fft: P x 8 x 8, all eight planes): any P, any window, host coefficients, screens;
//   3  the same without the planes a centred window of up to 96 pixels never reads (centre_planes(P, 8, 0)): P = 18, 20, 24, 28;
//  P = 16 (1024, and 2048 / 4096 as sub-rows) runs the 16 x 4 lane factorisation (pruned_row_fft_d16r):
//   4  six of sixteen planes, dense images, sixteen waves per workgroup: centred windows of up to 96 pixels -- the BASELINE
//      workloads; also with host coefficients / screens at 1024^2;
//   8  eight planes, dense images: centred windows of 97-128 pixels (only four rows of the stage-2b table are staged, so the
//      tables still fit beside sixteen exchange buffers);
//   5  six planes in the twelve-wave kernels (split columns of 2048 / 4096, host coefficients / screens of the split rows);
//   6  eight planes, twelve waves (97-128 pixels on the split grids, or where the dense tables do not fit);
//   7  all sixteen planes, twelve waves: any other window (NS = 2, 4, 8) -- the 4-term sums alone pay for the larger butterfly.
//   9  (rows, MODE 1 only) the row of 4 in FOUR-wave workgroups (59 KB of LDS): the HBM-bound coefficient rows of the same-seed mode
//      as a light workgroup that fits a CU BESIDE three workgroups of the numpy-stream generator (fastmc_run_npstream on two streams)
template <class R, int P, int NS, int D> struct WCfg {
  static constexpr int OM_ROWS = (D >= 4) ? 4 : 8;     // stage-2b table rows in the LDS: the 16 x 4 row reads rows 1 ... 3
  static constexpr bool DENSE = (D == 4 || D == 8 || D == 9);
  static_assert(D == 0 || (D == 3 && NS == 2 && P > 16) || (P == 16 && NS == 2 && D >= 4 && D <= 9) || (D == 7 && P == 16),
                "pruned planes for NS = 2; the 16 x 4 row for P = 16");
  static_assert(WaveGeom<R, 16>::XELEMS >= D16_XELEMS, "the 16 x 4 row (D = 5) runs in the twelve-wave exchange buffer");
  static constexpr int WPB = D == 9 ? 4 : (DENSE ? 16 : WaveCfg<R, P, NS>::WPB_ROWS);
  // the column kernel: sixteen waves as the rows (A/B at 1024^2: two six-wave workgroups per CU -1 %, one of eight +18 %)
  static constexpr int WPB_COLS = DENSE ? 16 : WaveCfg<R, P, NS>::WPB;
  static constexpr int XELEMS = DENSE ? D16_XELEMS : WaveGeom<R, P>::XELEMS;
};
// the plane set of a P = 16 variant
template <int D> constexpr int d16r_mask() { return D == 7 ? 0xFFFF : (D == 6 || D == 8) ? D16R_WIDE_MASK : D16R_CENTRE_MASK; }

template <class R, int P, int OM_ROWS = 8>
__device__ __forceinline__ void load_tables(cpx<R>* s_tw, cpx<R>* s_om, const cpx<R>* tw, const cpx<R>* om, int omS) {
  for (int i = threadIdx.x; i < P * WAVE; i += blockDim.x) s_tw[i] = tw[i];
  for (int i = threadIdx.x; i < OM_ROWS * omS; i += blockDim.x) s_om[i] = om[i];    // the 16 x 4 row reads rows 1 ... 3 only
  __syncthreads();
}
// The wave family stages neither table's row 0 (w^0 = 1: no row reads it -- fmc_wavefft.h multiplies by tw1[a][.] for a >= 1 and
// by om[m][.] for m >= 1): 1 KB + 16 omS bytes (float64) that the float64 generator's tables need beside sixteen exchange
// buffers.  `s_tw` / `s_om` are the tables' VIRTUAL bases (row 0 would start there): WaveLds carves them.
template <class R, int P, int OM_ROWS = 8>
__device__ __forceinline__ void load_tables_skip0(cpx<R>* s_tw, cpx<R>* s_om, const cpx<R>* tw, const cpx<R>* om, int omS) {
  for (int i = WAVE + threadIdx.x; i < P * WAVE; i += blockDim.x) s_tw[i] = tw[i];
  for (int i = omS + threadIdx.x; i < OM_ROWS * omS; i += blockDim.x) s_om[i] = om[i];
  __syncthreads();
}
template <class R, int P>
__host__ __device__ constexpr size_t wave_table_bytes(int om_rows, int omS) {
  return (size_t)((P - 1) * WAVE + (om_rows - 1) * omS) * sizeof(cpx<R>);
}
// [generator tables (MODE 2)][tw1 rows 1 ... P-1][om rows 1 ... OM_ROWS-1][exchange buffers]
template <class R, int P, int OM_ROWS>
struct WaveLds {
  cpx<R>* s_tw;
  cpx<R>* s_om;
  typename Xch<R>::E* s_x;
  __device__ __forceinline__ WaveLds(unsigned char* smem, size_t gen_bytes, int omS) {
    cpx<R>* t = reinterpret_cast<cpx<R>*>(smem + gen_bytes);
    s_tw = t - WAVE;
    s_om = t + (P - 1) * WAVE - omS;
    s_x = reinterpret_cast<typename Xch<R>::E*>(t + (P - 1) * WAVE + (OM_ROWS - 1) * omS);
  }
};

// LDS carve (dynamic): [tw1 rows 1 ... P-1][om rows 1 ... 7][xbuf WPB*XELEMS 8-byte]
template <class R, int P, int NS>
__host__ __device__ constexpr size_t wave_lds_bytes(int omS) {
  return wave_table_bytes<R, P>(8, omS) + (size_t)WaveCfg<R, P, NS>::WPB * WaveGeom<R, P>::XELEMS * 8;
}
template <class R, int P, int NS, int D>
__host__ __device__ constexpr size_t wave_lds_bytes_cols(int omS) {
  return wave_table_bytes<R, P>(WCfg<R, P, NS, D>::OM_ROWS, omS) + (size_t)WCfg<R, P, NS, D>::WPB_COLS * WCfg<R, P, NS, D>::XELEMS * 8;
}
template <class R, int P, int NS, int D>
__host__ __device__ constexpr size_t wave_lds_bytes_d(int omS) {
  return wave_table_bytes<R, P>(WCfg<R, P, NS, D>::OM_ROWS, omS) + (size_t)WCfg<R, P, NS, D>::WPB * WCfg<R, P, NS, D>::XELEMS * 8;
}

// MODE 1 coefficients (host draws, numpy's stream drawn on the device): 16 bytes per element read once from a buffer of up to 1.7 GB --
// non-temporal loads (round 5: same-seed mode 96.6 -> 100.1 k it/s; profiles/r05_ab_generator_tables.txt section 14)
#define FMC_LDC(p) __builtin_nontemporal_load(p)
// S > 1: the row of NF = S * 64 P points is transformed as S interleaved sub-rows (kx = s mod S), each by
// the same P-per-lane pipeline, and the window outputs are combined, X[x] = sum_s w_NF^{s x} Y_s[x mod 64 P]
// (decimation in time, evaluated only for the window).  2048 = 2 x 1024 and 4096 = 4 x 1024 run the
// P = 16 pipeline at 3 waves per SIMD instead of a 32-values-per-lane pipeline at 2.
// rows of the small grids keep their stage-2b table values in registers over the rows of a wave (fmc_wavefft.h: OMC;
// A/B in round 3: rows -12 % at 128^2 and 256^2, -4.5 % at 512^2)
template <int P, int NS, int S, int D> constexpr bool row_omc() { return D == 0 && S == 1 && NS == 2 && P <= 8; }
template <class R, int P, int NS, int D, bool OMC = false, class Exec>
__device__ __forceinline__ void wave_row_fft(Exec& ex, typename Xch<R>::E* xbuf, const cpx<R>* s_tw, const cpx<R>* s_om, int omS, int lo, int Np) {
  if constexpr (D >= 4) pruned_row_fft_d16r<R, NS, d16r_mask<D>()>(ex, xbuf, s_tw, s_om, omS, lo, Np);
  else pruned_row_fft<R, P, NS, (D == 3 ? centre_planes(P, 8, 0) : 0xFF), OMC>(ex, xbuf, s_tw, s_om, omS, lo, Np);
}

template <class R, int P, int NS, int MODE, int S = 1, int D = 0>
__global__ __launch_bounds__((WCfg<R, P, NS, D>::WPB * 64)) void k_rows_wave(RowArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using G = WaveGeom<R, P>;
  using E = typename Xch<R>::E;
  // MODE 2 (float64 generator fused into the row): its 4 KB of tables (log, cos / sin) at the start of the LDS (a table offset IS the address)
  Gen64Entry* s_g64 = reinterpret_cast<Gen64Entry*>(smem);
  const WaveLds<R, P, WCfg<R, P, NS, D>::OM_ROWS> lds(smem, MODE == 2 ? GEN64_TABLE_BYTES : 0, A.omS);
  cpx<R>* s_tw = lds.s_tw;
  cpx<R>* s_om = lds.s_om;
  E* s_x = lds.s_x;
  if constexpr (MODE == 2) { gen64_lds0_check(s_g64); load_gen64_table(s_g64, A.g64); }
  load_tables_skip0<R, P, WCfg<R, P, NS, D>::OM_ROWS>(s_tw, s_om, A.tw, A.om, A.omS);

  // the wave index is wave-uniform: in an SGPR, so that row / realisation indices and the table base addresses
  // derived from it are scalar and the loads use the scalar-base + lane-offset form
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Effective shader clock of THIS launch (fastmc_last_clock): the first lane of the middle workgroup stamps the shader-clock
  // counter (s_memtime) and the constant-rate counter (s_memrealtime) before and after its rows -- about a hundred
  // microseconds in the middle of the launch, every CU busy with the same kernel: the clock the row time was paid in.
  const bool stamp = A.clk != nullptr && blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0;
  if (stamp) { A.clk[0] = (unsigned long long)clock64(); A.clk[1] = (unsigned long long)wall_clock64(); }
  E* xbuf = s_x + w * WCfg<R, P, NS, D>::XELEMS;
  const int N = S * G::N;           // full row length
  LaneRegs<R, P, NS> regs;
  GpuExec<R, P, NS> ex{lane, regs};
  // A workgroup owns LR consecutive rows (one 128-byte line of every V column) of RPW*WPB/LR
  // consecutive realisations, and its waves walk that tile row-fastest: the LR 16-byte pieces of a
  // line are stored by LR different waves within one or two iterations, so the line is complete
  // in L2 long before it is evicted.  (With one wave storing its own 8 rows over 8 iterations the
  // partially written lines in flight -- 256 CUs x 12 waves x Np lines -- equal the L2 capacity
  // and leave as partial writes: 2.2x write amplification, rows kernel 13.9 -> 12.6 ms per 5000 realisations.)
  constexpr int WPB = WCfg<R, P, NS, D>::WPB;
  constexpr int LR = 128 / (int)sizeof(cpx<R>);
  static_assert((ROWS_PER_WAVE * WPB) % LR == 0, "tile must hold whole lines");
  const int rpw = A.rpw ? A.rpw : ROWS_PER_WAVE;         // (a multiple of LR / gcd(WPB, LR): whole lines)
  const int BPG = rpw * WPB / LR;                        // realisations per workgroup
  const int nbb = (A.nb + BPG - 1) / BPG;
  constexpr bool OMC = row_omc<P, NS, S, D>();
  if constexpr (OMC) {
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
      for (int m = 1; m < 8; ++m) regs.omc[s2][m] = s_om[m * A.omS + min(lane + WAVE * s2, A.omS - 1)];
  }
  // the workgroups of a launch stay: each walks the launch's tiles from its own index in steps of the grid
  // (launch_rows_wave: as many workgroups as the device holds at once), so the tables are staged once per CU
  const int tiles = A.tiles ? A.tiles : (int)gridDim.x;
#pragma unroll 1
  for (int vb = blockIdx.x; vb < tiles; vb += gridDim.x) {
  const int b0 = (vb % nbb) * BPG;                       // realisation block fastest: neighbours share amp rows
  const int row0 = (vb / nbb) * LR;
#pragma unroll 1
  for (int rr = 0; rr < rpw; ++rr) {
    const int flat = rr * WPB + w;
    const int b = b0 + flat / LR;
    if (b >= A.nb) break;                                // wave-uniform
    const int ky = row0 + flat % LR;
    const uint64_t g = A.g0 + (uint64_t)b;
    const R* amp = A.amp + (size_t)ky * N;
    const float* ampf = A.ampf + (size_t)ky * N;
    R accr[NS], acci[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) { accr[s2] = (R)0; acci[s2] = (R)0; }
#pragma unroll 1
    for (int sp = 0; sp < S; ++sp) {
      // sub-row sp: kx = sp + S (lane + 64 j)
      if (MODE == 0) {
        xoshiro128p rs = row_stream(A.key, g, ky, sp + S * lane, WAVE * S);
#pragma unroll
        for (int j = 0; j < P; ++j) regs.v[j] = draw_coloured<R>(rs, ampf[sp + S * (lane + WAVE * j)]);
      } else if constexpr (MODE == 2) {
        // the generator at the reference's precision (fast/funcs.py:352-356, fast/fast.py:593-594): 53-bit normals, float64
        // colouring; four words per coefficient from one advance of the stream (g, ky, L)
        static_assert(sizeof(R) == 8, "the float64 generator feeds the float64 pipeline");
        xoshiro128p rs = row_stream(A.key, g, ky, sp + S * lane, WAVE * S);
        // one coefficient at a time, its colouring factor loaded one draw ahead: left alone the compiler issues the sixteen
        // float64 table loads (32 VGPRs) before the first draw and spills
        double an = (double)amp[sp + S * lane];
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const double a = an;
          if (j + 1 < P) an = (double)amp[sp + S * (lane + WAVE * (j + 1))];
          ex.loadfence();
          regs.v[j] = draw_coloured_f64(rs, a, Gen64Lds0{});
          // the draws one after the other: a finished coefficient and the stream state are pinned here, so that no
          // arithmetic of draw j + 1 starts (and holds registers) before draw j has retired its temporaries
          asm volatile("" : "+v"(regs.v[j].x), "+v"(regs.v[j].y), "+v"(rs.s0), "+v"(rs.s1), "+v"(rs.s2), "+v"(rs.s3));
        }
      } else {
        // coefficients from HBM: three float64 loads per element (real part, imaginary part, colouring factor).  All P at once are
        // 6 P VGPRs of loads in flight on top of the 4 P of the row (P = 16: 68 B of scratch per lane); in two halves the second
        // half's loads only start when the first has arrived.  So: chunks of four elements through two register buffers, chunk
        // k + 2 requested as soon as chunk k is consumed -- loads are in flight for the whole of the load phase, 48 VGPRs of them.
        const size_t base = ((size_t)b * N + ky) * N;
        constexpr int CH = (sizeof(R) == 8 && P > 8 && P % 4 == 0) ? 4 : P, NCH = P / CH;
        if constexpr (NCH == 1) {
#pragma unroll
          for (int j = 0; j < P; ++j) {
            const int kx = sp + S * (lane + WAVE * j);
            regs.v[j] = cscale(mk<R>((R)FMC_LDC(A.cre + base + kx), (R)FMC_LDC(A.cim + base + kx)), amp[kx]);
          }
        } else {
          R cr[2][CH], ci[2][CH], am[2][CH];
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
              const int kx = sp + S * (lane + WAVE * (k * CH + i));
              cr[k][i] = (R)FMC_LDC(A.cre + base + kx); ci[k][i] = (R)FMC_LDC(A.cim + base + kx); am[k][i] = amp[kx];
            }
#pragma unroll
          for (int k = 0; k < NCH; ++k) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
              regs.v[k * CH + i] = cscale(mk<R>(cr[k & 1][i], ci[k & 1][i]), am[k & 1][i]);
              asm volatile("" : "+v"(regs.v[k * CH + i].x), "+v"(regs.v[k * CH + i].y) : : "memory");      // (no later load moves above this)
            }
            if (k + 2 < NCH) {
#pragma unroll
              for (int i = 0; i < CH; ++i) {
                const int kx = sp + S * (lane + WAVE * ((k + 2) * CH + i));
                cr[k & 1][i] = (R)FMC_LDC(A.cre + base + kx); ci[k & 1][i] = (R)FMC_LDC(A.cim + base + kx); am[k & 1][i] = amp[kx];
              }
            }
          }
        }
      }
      wave_row_fft<R, P, NS, D, OMC>(ex, xbuf, s_tw, s_om, A.omS, A.lo, A.Np);
      if (S > 1) {
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
          const int oi = lane + WAVE * s2;
          if (oi < A.Np) {
            const cpx<R> c = A.cw[sp * A.omS + oi];
            accr[s2] += c.x * regs.xr[s2] - c.y * regs.xi[s2];
            acci[s2] += c.x * regs.xi[s2] + c.y * regs.xr[s2];
          }
        }
      }
    }
    if (S > 1) {
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) { regs.xr[s2] = accr[s2]; regs.xi[s2] = acci[s2]; }
    }
    // V is stored column-major per realisation, V[b][oi][ky], so that the column pass reads it
    // coalesced; the 8 consecutive rows of this wave complete one 128-byte line per window column.
    cpx<R>* out = A.V + (size_t)b * A.Np * N + ky;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int oi = lane + WAVE * s;
      if (oi < A.Np) out[(size_t)oi * N] = mk<R>(regs.xr[s], regs.xi[s]);
    }
  }
  if (A.tiles) __syncthreads();      // the waves of a workgroup stay within one tile of each other: the pieces of a V line leave together
  }
  if (stamp) { A.clk[2] = (unsigned long long)clock64(); A.clk[3] = (unsigned long long)wall_clock64(); }
}

// One element of a V column for the column pass of the one-pass wave kernels and the packed kernels.  V is written once by the row
// pass and read once here, a slab of 1-7 GB that no cache holds: the loads carry the non-temporal hint (round 5; 1024^2: columns
// 1.41 -> 1.26 ms per step, step +0.3 ... +1.1 % by box; 256^2 +2.5 %, 512^2 +1 %).  NOT where a line is read more than once (the split
// columns of 2048 / 4096 read every line in S passes: +9 % there), and never on the ROW pass's stores (-20 %: the L2 no longer
// combines the eight 16-byte pieces of a line).  profiles/r05_ab_generator_tables.txt section 13.
#ifndef FMC_V_NT_LOAD
#define FMC_V_NT_LOAD 1
#endif
template <class R>
__device__ __forceinline__ cpx<R> load_v(const cpx<R>* p) {
#if FMC_V_NT_LOAD
  typedef R vec2 __attribute__((ext_vector_type(2)));
  const vec2 t = __builtin_nontemporal_load(reinterpret_cast<const vec2*>(p));
  return mk<R>(t.x, t.y);
#else
  return *p;
#endif
}

// EPI 0: detector partial sums; EPI 1: write the cropped screens.
// Small grids (P <= 8): each wave is short-lived (one column, a few microseconds of mostly latency), so
// two workgroups per CU are worth a tighter register budget (6 waves per SIMD): column time -26 % at
// 128^2 / 256^2, -17 % at 512^2.  (The same hint on the rows kernel spills and gains nothing.)
// (Tried: the column's global loads issued before the table staging and its barrier: +20 % at 1024^2 -- the table copy queues
// behind 16 KB of column loads per wave.)
// the column kernels whose workgroups stay and walk the launch's columns (launch_cols_wave): the 1024-point pipeline, whole or two sub-rows
template <int P, int NS, int S> constexpr bool cols_walk() { return P == 16 && NS == 2 && S <= 2; }
template <class R, int P, int NS, int EPI, int S = 1, int D = 0>
__global__ __launch_bounds__((WCfg<R, P, NS, D>::WPB_COLS * 64), ((P <= 8 && P != 7 && NS == 2 && WaveCfg<R, P, NS>::WPB == 12) ? 6 : 1))
void k_cols_wave(ColArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using G = WaveGeom<R, P>;
  using E = typename Xch<R>::E;
  const WaveLds<R, P, WCfg<R, P, NS, D>::OM_ROWS> lds(smem, 0, A.omS);
  cpx<R>* s_tw = lds.s_tw;
  cpx<R>* s_om = lds.s_om;
  E* s_x = lds.s_x;

  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  E* xbuf = s_x + w * WCfg<R, P, NS, D>::XELEMS;
  // work item = (realisation b, window column xi), xi fastest: adjacent waves read adjacent columns
  // (launch_cols_wave: the workgroups of a large launch stay and each wave walks the items in steps of the grid -- the
  // tables are staged once per CU; the waves need nothing from each other after the staging barrier)
  const int N = S * G::N;
  LaneRegs<R, P, NS> regs;
  load_tables_skip0<R, P, WCfg<R, P, NS, D>::OM_ROWS>(s_tw, s_om, A.tw, A.om, A.omS);
  const int items = A.nb * A.Np;
  const int lane0 = lane;
  // (instantiations outside cols_walk: one item per wave, no loop -- the walk costs the tighter register budgets of the small
  // grids and the four-sub-row columns 100+ bytes of scratch)
  int item = blockIdx.x * WCfg<R, P, NS, D>::WPB_COLS + w;
  if (item >= items) return;   // whole wave exits; no block barrier follows
#pragma unroll 1
  do {
  // the lane index is made opaque per item: nothing lane-dependent is hoisted over the walk (hoisted table values and offsets
  // cost 240 bytes of scratch per lane otherwise)
  int lane = lane0;
  if constexpr (cols_walk<P, NS, S>()) asm volatile("" : "+v"(lane));
  GpuExec<R, P, NS> ex{lane, regs};
  const int b = item / A.Np;
  const int xi = item % A.Np;
  const cpx<R>* col = A.V + ((size_t)b * A.Np + xi) * N;
  if (S == 1) {
#pragma unroll
    for (int j = 0; j < P; ++j) regs.v[j] = load_v(col + lane + WAVE * j);
    wave_row_fft<R, P, NS, D>(ex, xbuf, s_tw, s_om, A.omS, A.lo, A.Np);
  } else {
    R accr[NS], acci[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) { accr[s2] = (R)0; acci[s2] = (R)0; }
#pragma unroll 1
    for (int sp = 0; sp < S; ++sp) {
#pragma unroll
      for (int j = 0; j < P; ++j) regs.v[j] = col[sp + S * (lane + WAVE * j)];      // (every line is read by S passes: no hint)
      wave_row_fft<R, P, NS, D>(ex, xbuf, s_tw, s_om, A.omS, A.lo, A.Np);
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        const int oi = lane + WAVE * s2;
        if (oi < A.Np) {
          const cpx<R> c = A.cw[sp * A.omS + oi];
          accr[s2] += c.x * regs.xr[s2] - c.y * regs.xi[s2];
          acci[s2] += c.x * regs.xi[s2] + c.y * regs.xr[s2];
        }
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) { regs.xr[s2] = accr[s2]; regs.xi[s2] = acci[s2]; }
  }
  column_epilogue<R, NS, EPI>(A.sh, A.W, A.partial, A.phs, A.nb, A.Np, b, xi, lane, regs.xr, regs.xi);
  item += (int)gridDim.x * WCfg<R, P, NS, D>::WPB_COLS;
  } while (cols_walk<P, NS, S>() && item < items);
}

// ================================================================== packed rows: N = 128 (L0 = 0), 256 (L0 = 1), 512 (L0 = 2)
// G = 8, 4, 2 rows (columns) per wavefront on the sixteen-values-per-lane pipeline of the 1024-point row (fmc_wavefft.h:
// packed_row_fft).  Generator: L = 16 L0 streams per row, stream q = kx mod L, sixteen advances each (fmc_core.h:
// stream_lanes) -- ONE Philox block per lane and G rows.  D 0: the six planes of a centred window of up to 96 pixels; D 1: all
// sixteen planes, any window of up to 256 pixels.  Sixteen waves per workgroup.
template <class R, int L0, int D> struct PkCfg {
  static constexpr int L = pk_lanes(L0), G = WAVE / L, N = 16 * L;
  static constexpr int B0M = D == 0 ? pk_centre_mask<L0>() : pk_all_mask<L0>();
  static constexpr int WMAX = D == 0 ? 96 : (N < 256 ? N : 256);   // widest window
  static constexpr int NSL = L0 == 2 ? WMAX / L : 1;       // output slots of L lanes (L0 = 2 only)
  // rows: as many waves per workgroup as run without spilling (A/B at 256^2: eight-wave workgroups at four per SIMD no better)
  static constexpr int WPB = (D == 1 && L0 <= 1) ? 8 : ((L0 <= 1 || D == 1) ? 12 : 16);
  // columns, D = 0: the rolled detector loop fits four waves per SIMD, and several small workgroups per CU overlap one
  // group's start-up (table copy, barrier, first loads) with the others' arithmetic: two of eight waves -9 % at 256^2, -8 %
  // at 512^2 against one of sixteen; four of four waves another -6 % at 128^2 / 256^2 (0 at 512^2; five of three and three
  // of five: worse).  Twiddles read from global memory instead of staged (no barrier): +5 ... +12 %
  // (profiles/r03_ab_packed_columns.txt)
  static constexpr int WPC = D == 0 ? (L0 == 2 ? 8 : 4) : WPB;
  static constexpr int CMINB = D == 0 ? 4 : 1;             // waves per SIMD the column kernel's register budget is cut for
  static constexpr int OM_ROWS = L0 == 2 ? 2 : 0;
};
// LDS carve (dynamic): [tw1 16 L cpx][om OM_ROWS omS cpx][xbuf wpb * D16_XELEMS 8-byte]
template <class R, int L0>
__host__ __device__ constexpr size_t pk_lds_bytes(int omS, int wpb) {
  return (size_t)(16 * pk_lanes(L0) + (L0 == 2 ? 2 : 0) * omS) * sizeof(cpx<R>) + (size_t)wpb * D16_XELEMS * 8;
}
template <class R, int L0>
__device__ __forceinline__ void pk_load_tables(cpx<R>* s_tw, cpx<R>* s_om, const cpx<R>* tw, const cpx<R>* om, int omS) {
  for (int i = threadIdx.x; i < 16 * pk_lanes(L0); i += blockDim.x) s_tw[i] = tw[i];
  if (L0 == 2)
    for (int i = threadIdx.x; i < 2 * omS; i += blockDim.x) s_om[i] = om[i];
  __syncthreads();
}

template <class R, int L0, int MODE, int D>
__global__ __launch_bounds__((PkCfg<R, L0, D>::WPB * 64)) void k_rows_pk(RowArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using C = PkCfg<R, L0, D>;
  using E = typename Xch<R>::E;
  constexpr int L = C::L, G = C::G, N = C::N, WPB = C::WPB;
  // MODE 2 (float64 generator fused into the rows, as in k_rows_wave): its 4 KB of tables at the start of the LDS
  Gen64Entry* s_g64 = reinterpret_cast<Gen64Entry*>(smem);
  cpx<R>* s_tw = reinterpret_cast<cpx<R>*>(smem + (MODE == 2 ? GEN64_TABLE_BYTES : 0));
  cpx<R>* s_om = s_tw + 16 * L;
  E* s_x = reinterpret_cast<E*>(s_om + C::OM_ROWS * A.omS);
  if constexpr (MODE == 2) { gen64_lds0_check(s_g64); load_gen64_table(s_g64, A.g64); }
  pk_load_tables<R, L0>(s_tw, s_om, A.tw, A.om, A.omS);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  E* xbuf = s_x + w * D16_XELEMS;
  LaneRegs<R, 16, C::NSL> regs;
  GpuExec<R, 16, C::NSL> ex{lane, regs};
  // tile walk as in k_rows_wave, in units of G rows: a workgroup owns the LR rows of one 128-byte line of every V column
  // (LR / G units) for ROWS_PER_WAVE * WPB * G / LR consecutive realisations
  constexpr int LR = 128 / (int)sizeof(cpx<R>), LU = LR / G;
  static_assert(LR % G == 0 && (ROWS_PER_WAVE * WPB) % LU == 0, "tile must hold whole lines");
  constexpr int BPG = ROWS_PER_WAVE * WPB / LU;
  const int nbb = (A.nb + BPG - 1) / BPG;
  const int q = lane & (L - 1), gl = lane / L;
  const int lane_in = gl * N + q;                        // this lane's first input of the G rows of a unit (rows are contiguous)
  // (round 6) the workgroups of a large launch stay and walk its tiles, as k_rows_wave's do: tables staged once per CU
  const int tiles = A.tiles ? A.tiles : (int)gridDim.x;
#pragma unroll 1
  for (int vb = blockIdx.x; vb < tiles; vb += gridDim.x) {
  const int b0 = (vb % nbb) * BPG;
  const int row0 = (vb / nbb) * LR;
#pragma unroll 1
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
    const int flat = rr * WPB + w;
    const int b = b0 + flat / LU;
    if (b >= A.nb) break;                                // wave-uniform
    const int ky0 = row0 + (flat % LU) * G;              // wave-uniform: scalar bases, lane offsets
    const uint64_t g = A.g0 + (uint64_t)b;
    if (MODE == 0) {
      const float* ampf = A.ampf + (size_t)ky0 * N;
      xoshiro128p rs = row_stream(A.key, g, ky0 + gl, q, L);
#pragma unroll
      for (int j = 0; j < 16; ++j) regs.v[j] = draw_coloured<R>(rs, ampf[lane_in + L * j]);
    } else if constexpr (MODE == 2) {
      static_assert(sizeof(R) == 8, "the float64 generator feeds the float64 pipeline");
      const R* amp = A.amp + (size_t)ky0 * N;
      xoshiro128p rs = row_stream(A.key, g, ky0 + gl, q, L);
      double an = (double)amp[lane_in];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double a = an;
        if (j + 1 < 16) an = (double)amp[lane_in + L * (j + 1)];
        ex.loadfence();
        regs.v[j] = draw_coloured_f64(rs, a, Gen64Lds0{});
        asm volatile("" : "+v"(regs.v[j].x), "+v"(regs.v[j].y), "+v"(rs.s0), "+v"(rs.s1), "+v"(rs.s2), "+v"(rs.s3));
      }
    } else {
      const R* amp = A.amp + (size_t)ky0 * N;
      const size_t base = ((size_t)b * N + ky0) * N;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        regs.v[j] = cscale(mk<R>((R)FMC_LDC(A.cre + base + lane_in + L * j), (R)FMC_LDC(A.cim + base + lane_in + L * j)), amp[lane_in + L * j]);
    }
    packed_row_fft<R, L0, C::NSL, C::B0M>(ex, xbuf, s_tw, s_om, A.omS, A.lo, A.Np);
    cpx<R>* out = A.V + (size_t)b * A.Np * N + ky0;      // V[b][oi][ky]
    packed_outputs<R, L0, C::NSL, C::B0M>(lane, regs, A.lo, A.Np,
                                          [&](int oi, R re, R im) { out[(uint32_t)(oi * N + gl)] = mk<R>(re, im); });   // scalar base + 32-bit lane offset
  }
  if (A.tiles) __syncthreads();      // the waves of a workgroup stay within one tile of each other
  }
}

// G window columns per wavefront; the detector sums of a column are reduced over the L lanes of its group (DPP inside the
// 16-lane rows, v_readlane across them: a fixed order).
template <class R, int L0, int EPI, int D>
__global__ __launch_bounds__((PkCfg<R, L0, D>::WPC * 64), (PkCfg<R, L0, D>::CMINB)) void k_cols_pk(ColArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using C = PkCfg<R, L0, D>;
  using E = typename Xch<R>::E;
  constexpr int L = C::L, G = C::G, N = C::N, WPC = C::WPC;
  cpx<R>* s_tw = reinterpret_cast<cpx<R>*>(smem);
  cpx<R>* s_om = s_tw + 16 * L;
  E* s_x = reinterpret_cast<E*>(s_om + C::OM_ROWS * A.omS);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  E* xbuf = s_x + w * D16_XELEMS;
  const int q = lane & (L - 1), gl = lane / L;
  LaneRegs<R, 16, C::NSL> regs;
  GpuExec<R, 16, C::NSL> ex{lane, regs};
  pk_load_tables<R, L0>(s_tw, s_om, A.tw, A.om, A.omS);
  // work item = (realisation b, group of G window columns), group fastest; the waves of a workgroup take neighbouring items
  // (one item per wave: a loop over several items makes the compiler hoist the per-plane invariants out of it and spill)
  const int ngrp = (A.Np + G - 1) / G;
  {
    const int item = blockIdx.x * WPC + w;
    if (item >= A.nb * ngrp) return;                       // wave-uniform; no block barrier follows
    const int b = item / ngrp;
    const int xi = (item % ngrp) * G + gl;
    const bool live = xi < A.Np;                           // the last group of a realisation may be short
    const cpx<R>* col = A.V + ((size_t)b * A.Np + (item % ngrp) * G) * N;
    const uint32_t lane_in = (live ? gl : 0) * N + q;
#pragma unroll
    for (int j = 0; j < 16; ++j) regs.v[j] = load_v(col + lane_in + L * j);
    packed_row_fft<R, L0, C::NSL, C::B0M>(ex, xbuf, s_tw, s_om, A.omS, A.lo, A.Np);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    auto pixel = [&](int yi, R p1, R p2) {
      pixel_phase<R>(A.sh, b, A.Np, yi, xi, p1, p2);
      if (EPI == 1) {
        const size_t plane = (size_t)A.Np * A.Np;
        A.phs[((size_t)b) * plane + (size_t)yi * A.Np + xi] = (double)p1;
        A.phs[((size_t)(A.nb + b)) * plane + (size_t)yi * A.Np + xi] = (double)p2;
      } else {
        const double wgt = A.W[(size_t)yi * A.Np + xi];
        double s1, c1, s2, c2;
        sincos_r(p1, s1, c1);
        sincos_r(p2, s2, c2);
        acc[0] += wgt * c1; acc[1] += wgt * s1; acc[2] += wgt * c2; acc[3] += wgt * s2;
      }
    };
    if constexpr (D == 0) {
      // the (at most six / three) outputs of a lane go through its own slots of the exchange buffer and the detector is a
      // rolled loop over them: a sixth of the code of the unrolled form (two inlined sincos instead of twelve) and its
      // registers -- the column kernel of the small grids is bound by the start-up of its short-lived waves, not by arithmetic
      // (N = 128: the planes of a = i, then those of a = i + 8)
      constexpr int NOUT = L0 == 2 ? C::NSL : popcount16(C::B0M), FIRST = L0 == 0 ? 1 : (L0 == 1 ? 5 : 0);
      constexpr int NM = L0 == 0 ? 2 : 1, STEP = L0 == 2 ? 32 : 16;
      static_assert(L0 == 2 || C::B0M == (((1 << NOUT) - 1) << FIRST), "contiguous planes");
      cpx<R>* ob = reinterpret_cast<cpx<R>*>(xbuf) + lane;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
#pragma unroll
        for (int p = 0; p < NOUT; ++p)
          ob[WAVE * p] = L0 == 2 ? mk<R>(regs.xr[p % C::NSL], regs.xi[p % C::NSL]) : regs.v[(8 * m + FIRST + p) & 15];
        ex.sync();
        const int y0 = L0 == 2 ? q : q + 8 * m + 16 * FIRST - A.lo;
#pragma unroll 1
        for (int p = 0; p < NOUT; ++p) {
          const int yi = y0 + STEP * p;
          if (live && yi >= 0 && yi < A.Np) {
            const cpx<R> v = ob[WAVE * p];
            pixel(yi, v.x, v.y);
          }
        }
        ex.sync();
      }
    } else {
      packed_outputs<R, L0, C::NSL, C::B0M>(lane, regs, A.lo, A.Np, [&](int yi, R p1, R p2) { if (live) pixel(yi, p1, p2); });
    }
    if (EPI == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double v = acc[k];
        v += dpp_copy<0xB1>(v);     // quad_perm [1,0,3,2]
        v += dpp_copy<0x4E>(v);     // quad_perm [2,3,0,1]
        v += dpp_copy<0x141>(v);    // row_half_mirror: every lane holds the sum of its 8 lanes
        if (L0 >= 1) v += dpp_copy<0x140>(v);    // row_mirror: ... of its 16-lane row
        if (L0 == 2) {
          const long long bits = __double_as_longlong(v);
          const int lo32 = (int)(bits & 0xffffffffll), hi32 = (int)(bits >> 32);
          double r[4];
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2) {
            const int l2 = __builtin_amdgcn_readlane(lo32, 16 * k2), h2 = __builtin_amdgcn_readlane(hi32, 16 * k2);
            r[k2] = __longlong_as_double(((long long)h2 << 32) | (unsigned int)l2);
          }
          v = gl ? r[2] + r[3] : r[0] + r[1];
        }
        acc[k] = v;
      }
      if (q == 0 && live) {
        double* o = A.partial + ((size_t)b * A.Np + xi) * 4;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
      }
    }
  }
}

// ================================================================== packed SUB-ROWS: N = S x 256 (768, 1280, 1536, 1792), S x 128 (640, 896, 1152), S x 64 (192, 320, 448, 576)
// The row pass of the grids whose one-row-per-wave kernel holds 10 ... 28 values per lane (fmc_wavefft.h: pks_accumulate has the
// arithmetic): a wavefront owns G = 4 / 8 consecutive rows and transforms them one sub-row index s at a time on the packed
// pipeline of the 256 / 128-point grid -- sixteen draws per lane and pass from ONE generator stream (t = s + S q of SL = S L =
// N / 16: fmc_core.h stream_lanes), six planes accumulated in registers, V written once per row.  Centred windows of up to 96
// pixels, device generator (MODE 0 float32 draw, MODE 2 float64 generator); every other window / mode stays with the one-row-per-wave
// kernels.  The pair k_rows_pks / k_cols_pks keeps V permuted along ky (sub-row major).  Tile walk as k_rows_wave.
#ifndef FMC_PKS_WPB
#define FMC_PKS_WPB 12
#endif
#ifndef FMC_PKS_WPB0
#define FMC_PKS_WPB0 8
#endif
template <class R, int L0_, int S, int NPL = 6> struct PksCfg {
  // L0 = 1 / 0: sub-rows of 256 / 128 points on the packed pipeline (sixteen values per lane, L = 16 / 8 lanes per sub-row);
  // L0 = -1: sub-rows of SIXTY-FOUR points (192, 320, 448, 576): eight values per lane, eight lanes per sub-row (fmc_wavefft.h: pks64_pass)
  // S > 0: the sub-row count at compile time (192 ... 1792); S = 0 / -2: an odd / even count at RUN TIME (RowArgs::S: the grids of
  // fmc_core.h wave_rt_split up to 3840 -- 2304 = 9 x 256 ... 3840 = 15 x 256, 1920 ... 3456 = 15, 21, 27 x 128, 1344 / 1728 = 21 / 27 x 64)
  static constexpr int L0 = L0_, L = L0 < 0 ? 8 : pk_lanes(L0), VPL = L0 < 0 ? 8 : 16, G = WAVE / L, M = VPL * L, NM = pks_nm<L0, NPL>();
  static constexpr int SP = S > 0 ? S : (S == 0 ? 3 : 2);        // a count of the same parity: the plane set depends on it only
  static constexpr int B0M = L0 < 0 ? 0xFF : pks_plane_mask(L0 < 0 ? 0 : L0, SP, NPL), FIRST = pks_first_plane(L0 < 0 ? 0 : L0, SP, NPL);
  static constexpr int SPAN = pks_span(NPL);             // table entries per sub-row (NPL = 8: centred windows of up to 128 pixels)
  static constexpr int TWN = VPL * L;                    // entries of the sub-transform's twiddle table
  // 256-point sub-rows: 155 registers with the float64 generator, twelve waves; 128-point sub-rows carry twelve accumulators (48
  // more registers for float64): eight waves; 64-point sub-rows: twelve accumulators but eight values: twelve waves
  // (eight planes: two / four accumulators more per lane -- 163 registers with 256-point sub-rows: still twelve waves; 209 / 170 with
  // 128 / 64-point ones: eight)
  // (all sixteen planes, L0 = 1 only: sixteen accumulators -- eight waves)
  static constexpr int WPB = NPL > 8 ? 8 : (NPL > 6 ? (L0 == 1 ? 12 : 8) : (L0 == 0 ? (sizeof(R) == 8 ? FMC_PKS_WPB0 : 12) : FMC_PKS_WPB));
};
// LDS carve (dynamic): [generator tables (MODE 2)][tw1 16 L cpx][pcw S x 96 cpx][xbuf WPB * D16_XELEMS 8-byte]
// A run-time count (S <= 0) keeps only the CURRENT pass's 96 entries of pcw, one copy per wave (pks_slice): the table of S = 63 sub-rows
// would be 94 KB; the slice of pass s is loaded while the pass's draws run.
template <class R, int L0, int S, int NPL = 6>
__host__ __device__ constexpr size_t pks_lds_bytes(int Sr) {
  using C = PksCfg<R, L0, S, NPL>;
  return (size_t)(C::TWN + (S > 0 ? Sr : C::WPB) * C::SPAN) * sizeof(cpx<R>) + (size_t)C::WPB * D16_XELEMS * 8;
}
// The 96 entries of pass sp into the wave's slice by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane to base + 16 lane, no
// register in between -- the float64 rows have none to spare across their draws): one instruction of the whole wave and one of its
// lower half.  The reader calls lds_dma_wait() first.  float64 pipeline only (16-byte entries).
// An LDS-DMA is a pending LDS write on the VECTOR-MEMORY counter: nothing orders it against the wave's DS reads, and hipcc does not
// wait for it on its own -- neither at a wavefront-scope fence (ex.sync() is a compiler barrier only) nor at an LDS read that aliases its
// destination (seen in the ISA of k_rows_pbz: no s_waitcnt between the DMA and the read; the draws in between hid it in every test
// until a read followed the DMA directly).  Every reader of DMA'd data calls this first.
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int SPAN>
__device__ __forceinline__ void pks_slice_load(const cpx<double>* cw, cpx<double>* slice, int sp, int lane) {
  const cpx<double>* src = cw + sp * SPAN + lane;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)slice, 16, 0, 0);
  if (SPAN >= 128 || lane < 32)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64), (__attribute__((address_space(3))) void*)(slice + 64), 16, 0, 0);
  if constexpr (SPAN == 256) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 128), (__attribute__((address_space(3))) void*)(slice + 128), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 192), (__attribute__((address_space(3))) void*)(slice + 192), 16, 0, 0);
  }
}
template <int SPAN>
__device__ __forceinline__ void pks_slice_load(const cpx<float>*, cpx<float>*, int, int) {}      // (never instantiated for a launch: fastmc.hip pks_variant)
template <class R, int L0, int S, int MODE, int NPL = 6>
__global__ __launch_bounds__((PksCfg<R, L0, S, NPL>::WPB * 64)) void k_rows_pks(RowArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using C = PksCfg<R, L0, S, NPL>;
  constexpr int PKS_SPAN = C::SPAN;                       // (shadows the six-plane constant)
  using E = typename Xch<R>::E;
  constexpr int L = C::L, G = C::G, WPB = C::WPB, VPL = C::VPL;
  const int Sr = S > 0 ? S : A.S, N = Sr * C::M;          // (S > 0: constants, folded; else the launch's)
  Gen64Entry* s_g64 = reinterpret_cast<Gen64Entry*>(smem);
  cpx<R>* s_tw = reinterpret_cast<cpx<R>*>(smem + (MODE == 2 ? GEN64_TABLE_BYTES : 0));
  cpx<R>* s_cw = s_tw + C::TWN;
  constexpr bool SLICE = S <= 0;                          // pcw by the pass, one copy per wave
  E* s_x = reinterpret_cast<E*>(s_cw + (SLICE ? WPB : Sr) * PKS_SPAN);
  if constexpr (MODE == 2) { gen64_lds0_check(s_g64); load_gen64_table(s_g64, A.g64); }
  for (int i = threadIdx.x; i < C::TWN; i += blockDim.x) s_tw[i] = A.tw[i];
  if constexpr (!SLICE)
    for (int i = threadIdx.x; i < Sr * PKS_SPAN; i += blockDim.x) s_cw[i] = A.cw[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  E* xbuf = s_x + w * D16_XELEMS;
  LaneRegs<R, 16, C::NM> regs;
  GpuExec<R, 16, C::NM> ex{lane, regs};
  constexpr int LR = 128 / (int)sizeof(cpx<R>), LU = LR / G;
  static_assert(LR % G == 0 && (ROWS_PER_WAVE * WPB) % LU == 0, "tile must hold whole lines");
  constexpr int BPG = ROWS_PER_WAVE * WPB / LU;
  const int nbb = (A.nb + BPG - 1) / BPG;
  const int q = lane & (L - 1), gl = lane / L;
  // V of these grids is stored PERMUTED along ky so that the column pass (k_cols_pks) reads its sub-rows contiguously: row
  // ky = s + S m lives at position s M + m of its window column.  A unit's G rows are therefore S apart (ky = kyb + S g: G
  // consecutive POSITIONS, a half / whole 128-byte line), a tile the LR positions of one line of one row class s.
  constexpr int M = C::M, TPS = M / LR;
  static_assert(M % LR == 0, "whole lines per row class");
  const int lane_in = gl * Sr * N + q;                   // this lane's first input of a sub-row (colouring tables SUB-ROW MAJOR: k_make_amp amp_p)
  const int tiles = A.tiles ? A.tiles : (int)gridDim.x;
#pragma unroll 1
  for (int vb = blockIdx.x; vb < tiles; vb += gridDim.x) {
  const int b0 = (vb % nbb) * BPG;
  const int tr = vb / nbb;
  const int s_r = tr / TPS, m0 = (tr % TPS) * LR;
#pragma unroll 1
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
    const int flat = rr * WPB + w;
    const int b = b0 + flat / LU;
    if (b >= A.nb) break;                                // wave-uniform
    const int mu = m0 + (flat % LU) * G;                 // position of the unit's first row within its class
    const int ky0 = s_r + Sr * mu;                       // ... and that row
    const uint64_t g = A.g0 + (uint64_t)b;
    pks_clear<R, L0, NPL>(ex);
#pragma unroll 1
    for (int sp = 0; sp < Sr; ++sp) {
      // sub-row sp of the G rows: kx = sp + S (q + L j), stream t = sp + S q of SL = S L
      xoshiro128p rs = row_stream(A.key, g, ky0 + Sr * gl, sp + Sr * q, Sr * L);
      // (the wave's slice: its reads of the pass before were issued, and waited for, before this -- DS operations of a wave run in order)
      if constexpr (SLICE) pks_slice_load<PKS_SPAN>(A.cw, s_cw + w * PKS_SPAN, sp, lane);
      if (MODE == 0) {
        const float* ampf = A.ampf + (size_t)ky0 * N + sp * C::M + lane_in;
#pragma unroll
        for (int j = 0; j < VPL; ++j) regs.v[j] = draw_coloured<R>(rs, ampf[L * j]);
      } else if constexpr (MODE == 2) {
        static_assert(sizeof(R) == 8, "the float64 generator feeds the float64 pipeline");
        const R* amp = A.amp + (size_t)ky0 * N + sp * C::M + lane_in;
        double an = (double)amp[0];
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
          const double a = an;
          if (j + 1 < VPL) an = (double)amp[L * (j + 1)];
          ex.loadfence();
          regs.v[j] = draw_coloured_f64(rs, a, Gen64Lds0{});
          asm volatile("" : "+v"(regs.v[j].x), "+v"(regs.v[j].y), "+v"(rs.s0), "+v"(rs.s1), "+v"(rs.s2), "+v"(rs.s3));
        }
      }
      const cpx<R>* cw_s = SLICE ? s_cw + w * PKS_SPAN : s_cw + sp * PKS_SPAN;
      if constexpr (SLICE) { lds_dma_wait(); ex.sync(); }      // (the draws' own loads have been consumed: nothing else is outstanding)
      if constexpr (L0 < 0) pks64_pass<R, NPL>(ex, xbuf, s_tw, cw_s);
      else {
        packed_row_fft<R, L0, C::NM, C::B0M>(ex, xbuf, s_tw, (const cpx<R>*)nullptr, 0, 0, A.Np);
        pks_accumulate<R, L0, C::FIRST, NPL>(ex, cw_s);
      }
    }
    cpx<R>* out = A.V + (size_t)b * A.Np * N + s_r * M + mu;      // V[b][oi][position of ky]
    pks_outputs<R, L0, NPL>(lane, regs, N, A.lo, A.Np, [&](int oi, R re, R im) { out[(uint32_t)(oi * N + gl)] = mk<R>(re, im); });
  }
  if (A.tiles) __syncthreads();      // (as k_rows_wave: the waves of a workgroup stay within one tile of each other)
  }
}

// The column pass of the same grids: G window columns per wavefront, S passes over a column's sub-rows (contiguous in the permuted V),
// the detector as k_cols_pk's rolled loop over the lane's accumulators, the sums reduced over the L lanes of a column by DPP.
template <class R, int L0, int S, int NPL = 6> struct PksColCfg {
  // (a run-time sub-row count: twelve waves -- sixteen exchange buffers + sixteen table slices would be 163 KB)
  // 122 registers (M = 256) / 152 (M = 128: twelve accumulators) with float64: four / three waves per SIMD; the exchange buffers and
  // the tables of sixteen / twelve waves fit the LDS (150 KB / 117 KB at most)
  static constexpr int WPC = NPL > 8 ? 8 : (NPL > 6 ? 12 : (S <= 0 ? 12 : ((L0 != 0 || sizeof(R) == 4) ? 16 : 12)));      // (64-point sub-rows: 126 registers; eight planes: 128-162; sixteen: eight waves)
};
template <class R, int L0, int S, int NPL = 6>
__host__ __device__ constexpr size_t pks_cols_lds_bytes(int Sr) {
  return (size_t)(PksCfg<R, L0, S, NPL>::TWN + (S > 0 ? Sr : PksColCfg<R, L0, S, NPL>::WPC) * pks_span(NPL)) * sizeof(cpx<R>) + (size_t)PksColCfg<R, L0, S, NPL>::WPC * D16_XELEMS * 8;
}
template <class R, int L0, int S, int EPI, int NPL = 6>
__global__ __launch_bounds__((PksColCfg<R, L0, S, NPL>::WPC * 64)) void k_cols_pks(ColArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using C = PksCfg<R, L0, S, NPL>;
  using E = typename Xch<R>::E;
  constexpr int PKS_SPAN = C::SPAN;
  constexpr int L = C::L, G = C::G, M = C::M, NM = C::NM, VPL = C::VPL, WPC = PksColCfg<R, L0, S, NPL>::WPC;
  const int Sr = S > 0 ? S : A.S, N = Sr * M;
  cpx<R>* s_tw = reinterpret_cast<cpx<R>*>(smem);
  cpx<R>* s_cw = s_tw + C::TWN;
  constexpr bool SLICE = S <= 0;                          // pcw by the pass, one copy per wave (as k_rows_pks)
  E* s_x = reinterpret_cast<E*>(s_cw + (SLICE ? WPC : Sr) * PKS_SPAN);
  for (int i = threadIdx.x; i < C::TWN; i += blockDim.x) s_tw[i] = A.tw[i];
  if constexpr (!SLICE)
    for (int i = threadIdx.x; i < Sr * PKS_SPAN; i += blockDim.x) s_cw[i] = A.cw[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  E* xbuf = s_x + w * D16_XELEMS;
  const int q = lane & (L - 1), gl = lane / L;
  LaneRegs<R, 16, NM> regs;
  GpuExec<R, 16, NM> ex{lane, regs};
  const int ngrp = (A.Np + G - 1) / G;
  const int item = blockIdx.x * WPC + w;
  if (item >= A.nb * ngrp) return;                         // wave-uniform; no block barrier follows
  const int b = item / ngrp;
  const int xi = (item % ngrp) * G + gl;
  const bool live = xi < A.Np;                             // the last group of a realisation may be short
  const cpx<R>* col = A.V + ((size_t)b * A.Np + (item % ngrp) * G) * N;
  const uint32_t lane_in = (live ? gl : 0) * N + q;
  pks_clear<R, L0, NPL>(ex);
#pragma unroll 1
  for (int sp = 0; sp < Sr; ++sp) {
#pragma unroll
    for (int j = 0; j < VPL; ++j) regs.v[j] = load_v(col + sp * M + lane_in + L * j);
    const cpx<R>* cw_s = SLICE ? s_cw + w * PKS_SPAN : s_cw + sp * PKS_SPAN;
    if constexpr (SLICE) { pks_slice_load<PKS_SPAN>(A.cw, s_cw + w * PKS_SPAN, sp, lane); lds_dma_wait(); ex.sync(); }      // (with the pass's loads of V, needed at once anyway)
    if constexpr (L0 < 0) pks64_pass<R, NPL>(ex, xbuf, s_tw, cw_s);
    else {
      packed_row_fft<R, L0, NM, C::B0M>(ex, xbuf, s_tw, (const cpx<R>*)nullptr, 0, 0, A.Np);
      pks_accumulate<R, L0, C::FIRST, NPL>(ex, cw_s);
    }
  }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  auto pixel = [&](int yi, R p1, R p2) {
    pixel_phase<R>(A.sh, b, A.Np, yi, xi, p1, p2);
    if (EPI == 1) {
      const size_t plane = (size_t)A.Np * A.Np;
      A.phs[((size_t)b) * plane + (size_t)yi * A.Np + xi] = (double)p1;
      A.phs[((size_t)(A.nb + b)) * plane + (size_t)yi * A.Np + xi] = (double)p2;
    } else {
      const double wgt = A.W[(size_t)yi * A.Np + xi];
      double s1, c1, s2, c2;
      sincos_r(p1, s1, c1);
      sincos_r(p2, s2, c2);
      acc[0] += wgt * c1; acc[1] += wgt * s1; acc[2] += wgt * c2; acc[3] += wgt * s2;
    }
  };
  cpx<R>* ob = reinterpret_cast<cpx<R>*>(xbuf) + lane;
  constexpr int PH = NPL > 8 ? 8 : NPL;                   // planes per pass (sixteen planes, L0 = 1: two passes of eight, plane 8 m + p)
#pragma unroll
  for (int m = 0; m < NM; ++m) {
#pragma unroll
    for (int p = 0; p < PH; ++p) ob[WAVE * p] = regs.omc[m][p];
    ex.sync();
    const int y0 = N / 2 - 8 * NPL + q + (L0 == 1 ? 128 : 8) * m - A.lo;     // (L0 = 1: q = a, e = a + 16 (8 m + p); L0 = 0, -1: e = q + 8 m + 16 p)
#pragma unroll 1
    for (int p = 0; p < PH; ++p) {
      const int yi = y0 + 16 * p;
      if (live && yi >= 0 && yi < A.Np) {
        const cpx<R> v = ob[WAVE * p];
        pixel(yi, v.x, v.y);
      }
    }
    ex.sync();
  }
  if (EPI == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double v = acc[k];
      v += dpp_copy<0xB1>(v);     // quad_perm [1,0,3,2]
      v += dpp_copy<0x4E>(v);     // quad_perm [2,3,0,1]
      v += dpp_copy<0x141>(v);    // row_half_mirror: every lane holds the sum of its 8 lanes
      if (L0 >= 1) v += dpp_copy<0x140>(v);    // row_mirror: ... of its 16-lane row
      acc[k] = v;
    }
    if (q == 0 && live) {
      double* o = A.partial + ((size_t)b * A.Np + xi) * 4;
      o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
    }
  }
}

// ================================================================== chirp-z family (any N with 64 P >= N + Np - 1)
// The row / column passes of the wave family for grid sizes that are not 64 P: every 1-D transform is a chirp-z
// (Bluestein) transform on the same pipeline (fmc_bluestein.h), window outputs only.  Same generator streams as the
// direct family (64 streams per row, stream L = kx mod 64), so the two agree to rounding.
// Waves per workgroup.  The chirp-z row keeps P complex values, the chirp tables' loads and the generator live at once:
// at P = 16 (f64) twelve waves (168-VGPR cap) spill 270 B per lane and run 17.6 ms per 5000 realisations at N = 500,
// eight waves (256-VGPR cap, no spill) 12.8 ms; at P = 24 eight waves (164 B of spill) beat four (no spill, one wave per
// SIMD) 48.9 to 70.3 ms at N = 1000; at P = 32 four waves beat six (690 B of spill) 140 to 207 ms at N = 1500.
template <class R, int P, int NS> struct BluCfg {
  static constexpr int W0 = WaveCfg<R, P, NS>::WPB;
  static constexpr int W1 = (NS == 4 && W0 > 8) ? 8 : W0;     // 256-pixel window tables: 8 waves fit the LDS
  static constexpr int CAP = sizeof(R) == 8 ? (P >= 28 ? 4 : (P >= 16 ? 8 : 12)) : 12;
  static constexpr int WPB = W1 > CAP ? CAP : W1;
  // the column kernel has no generator and fits the 168-VGPR step at P = 16: twelve waves (500^2: 2.59 against 3.08 ms)
  static constexpr int WPB_COLS = (P == 16 && NS == 2) ? W1 : WPB;
};
template <class R, int P, int NS>
__host__ __device__ constexpr size_t blu_lds_bytes(int omS, int wpb) {
  return (size_t)(P * WAVE + 8 * omS + 64) * sizeof(cpx<R>) + (size_t)wpb * BluGeom<R, P>::XELEMS * 8;
}

// BLK: rows longer than the largest M, cut into A.blu.SB input blocks of A.blu.B (fmc_bluestein.h header); the generator
// stream of a lane runs on across the blocks (B is a multiple of 64).
template <class R, int P, int NS, int MODE, bool BLK = false>
__global__ __launch_bounds__((BluCfg<R, P, NS>::WPB * 64)) void k_rows_blu(RowArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using BG = BluGeom<R, P>;
  using E = typename Xch<R>::E;
  // MODE 2 (float64 generator fused into the rows, as in k_rows_wave): its 4 KB of tables at the start of the LDS
  Gen64Entry* s_g64 = reinterpret_cast<Gen64Entry*>(smem);
  cpx<R>* s_tw = reinterpret_cast<cpx<R>*>(smem + (MODE == 2 ? GEN64_TABLE_BYTES : 0));
  cpx<R>* s_om = s_tw + P * WAVE;
  cpx<R>* s_twf = s_om + 8 * A.omS;
  E* s_x = reinterpret_cast<E*>(s_twf + 64);
  for (int i = threadIdx.x; i < 64; i += blockDim.x) s_twf[i] = A.blu.twf[i];
  if constexpr (MODE == 2) { gen64_lds0_check(s_g64); load_gen64_table(s_g64, A.g64); }
  load_tables<R, P>(s_tw, s_om, A.tw, A.om, A.omS);

  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  E* xbuf = s_x + w * BG::XELEMS;
  const int N = A.N;
  LaneRegs<R, P, NS> regs;
  GpuExec<R, P, NS> ex{lane, regs};
  constexpr int WPB = BluCfg<R, P, NS>::WPB;
  constexpr int LR = 128 / (int)sizeof(cpx<R>);
  static_assert((ROWS_PER_WAVE * WPB) % LR == 0, "tile must hold whole lines");
  const int rpw = A.rpw ? A.rpw : ROWS_PER_WAVE;         // (launch: pick_rpw)
  const int BPG = rpw * WPB / LR;
  const int nbb = (A.nb + BPG - 1) / BPG;
  const int b0 = (blockIdx.x % nbb) * BPG;
  const int row0 = (blockIdx.x / nbb) * LR;
  for (int rr = 0; rr < rpw; ++rr) {
    const int flat = rr * WPB + w;
    const int b = b0 + flat / LR;
    if (b >= A.nb) break;                                // wave-uniform
    const int ky = row0 + flat % LR;
    if (ky >= N) continue;                               // wave-uniform (N need not be a multiple of LR)
    const uint64_t g = A.g0 + (uint64_t)b;
    // The chirp tables do not depend on the row: left alone, the compiler hoists their 2 P + NS 16-byte loads out of
    // the row loop and then spills them (270 B per lane at P = 16).  An offset it cannot see through keeps the loads
    // inside the iteration.
    int zoff = 0;
    asm volatile("" : "+s"(zoff));
    const cpx<R>* pre = A.blu.pre + zoff;
    const cpx<R>* vhat = A.blu.vhat + zoff;
    const cpx<R>* post = A.blu.post + zoff;
    if constexpr (BLK) {
      R accr[NS], acci[NS];
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) { accr[s2] = (R)0; acci[s2] = (R)0; }
      const float* ampf = A.ampf + (size_t)ky * N;
      const R* amp = A.amp + (size_t)ky * N;
      const size_t base = ((size_t)b * N + ky) * N;
      xoshiro128p rs = row_stream(A.key, g, ky, lane, WAVE);
      const int nj = A.blu.B / WAVE;                     // values per lane and block
#pragma unroll 1
      for (int jb = 0; jb < A.blu.SB; ++jb) {
        const int k0 = jb * A.blu.B;
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const int kx = k0 + lane + WAVE * j;
          const bool in = j < nj && kx < N;              // draws only for coefficients of the row, in stream order
          if (MODE == 0) regs.v[j] = in ? cmul(draw_coloured<R>(rs, ampf[kx]), pre[kx]) : mk<R>((R)0, (R)0);
          else if constexpr (MODE == 2) {
            if constexpr (sizeof(R) == 8) {
              regs.v[j] = in ? cmul(draw_coloured_f64(rs, (double)amp[kx], Gen64Lds0{}), pre[kx]) : mk<R>((R)0, (R)0);
              asm volatile("" : "+v"(regs.v[j].x), "+v"(regs.v[j].y), "+v"(rs.s0), "+v"(rs.s1), "+v"(rs.s2), "+v"(rs.s3));
            }
          }
          else regs.v[j] = in ? cmul(cscale(mk<R>((R)FMC_LDC(A.cre + base + kx), (R)FMC_LDC(A.cim + base + kx)), amp[kx]), pre[kx]) : mk<R>((R)0, (R)0);
        }
        bluestein_row<R, P, NS>(ex, xbuf, s_tw, s_om, A.omS, s_twf, vhat + (size_t)jb * (WAVE * P), A.Np);
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) { accr[s2] += regs.xr[s2]; acci[s2] += regs.xi[s2]; }
      }
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) { regs.xr[s2] = accr[s2]; regs.xi[s2] = acci[s2]; }
    } else {
    if (MODE == 0) {
      const float* ampf = A.ampf + (size_t)ky * N;
      xoshiro128p rs = row_stream(A.key, g, ky, lane, WAVE);
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int kx = lane + WAVE * j;
        regs.v[j] = kx < N ? cmul(draw_coloured<R>(rs, ampf[kx]), pre[kx]) : mk<R>((R)0, (R)0);
      }
    } else if constexpr (MODE == 2) {
      // the generator at the reference's precision, one coefficient at a time (see k_rows_wave)
      static_assert(sizeof(R) == 8, "the float64 generator feeds the float64 pipeline");
      const R* amp = A.amp + (size_t)ky * N;
      xoshiro128p rs = row_stream(A.key, g, ky, lane, WAVE);
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int kx = lane + WAVE * j;
        regs.v[j] = kx < N ? cmul(draw_coloured_f64(rs, (double)amp[kx], Gen64Lds0{}), pre[kx]) : mk<R>((R)0, (R)0);
        asm volatile("" : "+v"(regs.v[j].x), "+v"(regs.v[j].y), "+v"(rs.s0), "+v"(rs.s1), "+v"(rs.s2), "+v"(rs.s3));
      }
    } else {
      const size_t base = ((size_t)b * N + ky) * N;
      const R* amp = A.amp + (size_t)ky * N;
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int kx = lane + WAVE * j;
        regs.v[j] = kx < N ? cmul(cscale(mk<R>((R)FMC_LDC(A.cre + base + kx), (R)FMC_LDC(A.cim + base + kx)), amp[kx]), pre[kx]) : mk<R>((R)0, (R)0);
      }
    }
    bluestein_row<R, P, NS>(ex, xbuf, s_tw, s_om, A.omS, s_twf, vhat, A.Np);
    }
    cpx<R>* out = A.V + (size_t)b * A.Np * N + ky;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int oi = lane + WAVE * s;
      if (oi < A.Np) {
        const cpx<R> q = post[oi];
        out[(size_t)oi * N] = mk<R>(q.x * regs.xr[s] + q.y * regs.xi[s], q.y * regs.xr[s] - q.x * regs.xi[s]);   // post * conj(Y)
      }
    }
  }
}

// (the blocked column carries the window accumulators: eight waves per workgroup as the rows, else 132 B of scratch)
template <class R, int P, int NS, int EPI, bool BLK = false>
__global__ __launch_bounds__(((BLK ? BluCfg<R, P, NS>::WPB : BluCfg<R, P, NS>::WPB_COLS) * 64)) void k_cols_blu(ColArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using BG = BluGeom<R, P>;
  using E = typename Xch<R>::E;
  cpx<R>* s_tw = reinterpret_cast<cpx<R>*>(smem);
  cpx<R>* s_om = s_tw + P * WAVE;
  cpx<R>* s_twf = s_om + 8 * A.omS;
  E* s_x = reinterpret_cast<E*>(s_twf + 64);
  for (int i = threadIdx.x; i < 64; i += blockDim.x) s_twf[i] = A.blu.twf[i];
  load_tables<R, P>(s_tw, s_om, A.tw, A.om, A.omS);

  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  E* xbuf = s_x + w * BG::XELEMS;
  const int item = blockIdx.x * (BLK ? BluCfg<R, P, NS>::WPB : BluCfg<R, P, NS>::WPB_COLS) + w;
  if (item >= A.nb * A.Np) return;   // whole wave exits; no block barrier follows
  const int b = item / A.Np;
  const int xi = item % A.Np;
  const int N = A.N;
  LaneRegs<R, P, NS> regs;
  GpuExec<R, P, NS> ex{lane, regs};
  const cpx<R>* col = A.V + ((size_t)b * A.Np + xi) * N;
  if constexpr (BLK) {
    R accr[NS], acci[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) { accr[s2] = (R)0; acci[s2] = (R)0; }
    const int nj = A.blu.B / WAVE;
#pragma unroll 1
    for (int jb = 0; jb < A.blu.SB; ++jb) {
      const int k0 = jb * A.blu.B;
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int ky = k0 + lane + WAVE * j;
        regs.v[j] = (j < nj && ky < N) ? cmul(col[ky], A.blu.pre[ky]) : mk<R>((R)0, (R)0);
      }
      bluestein_row<R, P, NS>(ex, xbuf, s_tw, s_om, A.omS, s_twf, A.blu.vhat + (size_t)jb * (WAVE * P), A.Np);
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) { accr[s2] += regs.xr[s2]; acci[s2] += regs.xi[s2]; }
    }
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) { regs.xr[s2] = accr[s2]; regs.xi[s2] = acci[s2]; }
  } else {
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int ky = lane + WAVE * j;
    regs.v[j] = ky < N ? cmul(col[ky], A.blu.pre[ky]) : mk<R>((R)0, (R)0);
  }
  bluestein_row<R, P, NS>(ex, xbuf, s_tw, s_om, A.omS, s_twf, A.blu.vhat, A.Np);
  }
  R p1s[NS], p2s[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int yi = lane + WAVE * s;
    const cpx<R> q = A.blu.post[yi < A.Np ? yi : 0];
    p1s[s] = q.x * regs.xr[s] + q.y * regs.xi[s];          // post * conj(Y)
    p2s[s] = q.y * regs.xr[s] - q.x * regs.xi[s];
  }
  column_epilogue<R, NS, EPI>(A.sh, A.W, A.partial, A.phs, A.nb, A.Np, b, xi, lane, p1s, p2s);
}

// ================================================================== chirp-z rows on the packed 256-point pipeline (round 6)
// The row pass of the chirp-z grids for windows of up to 128 pixels (fmc_bluestein.h: pbz_block / pbz_finish): a wavefront owns FOUR
// consecutive rows and walks them in blocks of 128 inputs -- eight draws per lane and block, lane q of a row reads the generator
// streams q, q + 16, q + 32, q + 48 of the 64 its row has (kx = 128 jb + q + 16 j belongs to stream kx mod 64, draw kx / 64: two draws
// per stream and block, in order), so the draws are those of k_rows_blu and of the direct family.  The column pass: k_cols_pbz below
// (standard V).  NPL: planes of the inverse transform kept (6: windows of up to 96 pixels, 8: up to 128).  float64 pipeline.
// Tile walk as k_rows_wave; a tile = the LR rows of one 128-byte line of V x BPG realisations.
constexpr int PBZ_WPB = 8;      // sixteen accumulators + sixteen values + four generator states per lane: two waves per SIMD
// LDS: [generator tables (MODE 2)][tw 256][V^ and pre-chirp of the block in hand, one copy per wave: 256 + 128 entries][exchange buffers]
template <class R> __host__ __device__ constexpr size_t pbz_lds_bytes(int waves) { return (size_t)(PBZ_M + waves * (PBZ_M + 2 * PBZ_B)) * sizeof(cpx<R>) + (size_t)waves * D16_XELEMS * 8; }
// V^_j of the block in hand into the wave's LDS copy by LDS-DMA (four instructions of 64 lanes x 16 bytes, no register in between):
// issued before the block's draws, so the product does not wait for sixteen loads from L2 per lane (fmc_kernels.h: pks_slice_load)
template <int CHUNKS>      // 64 entries each
__device__ __forceinline__ void pbz_lds_load(const cpx<double>* src, cpx<double>* dst, int lane) {
#pragma unroll
  for (int i = 0; i < CHUNKS; ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * i + lane),
                                     (__attribute__((address_space(3))) void*)(dst + 64 * i), 16, 0, 0);
}
// per wave: V^ of the block (256 entries) and two buffers of 128 pre-chirp factors (the block in hand and the next one)
constexpr int PBZ_SLICE = PBZ_M + 2 * PBZ_B;
template <class R, int NPL, int MODE>
__global__ __launch_bounds__(PBZ_WPB * 64) void k_rows_pbz(RowArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using E = typename Xch<R>::E;
  static_assert(sizeof(R) == 8, "chirp-z grids run the float64 pipeline (fastmc_create)");
  Gen64Entry* s_g64 = reinterpret_cast<Gen64Entry*>(smem);
  cpx<R>* s_tw = reinterpret_cast<cpx<R>*>(smem + (MODE == 2 ? GEN64_TABLE_BYTES : 0));
  cpx<R>* s_vh = s_tw + PBZ_M;
  E* s_x = reinterpret_cast<E*>(s_vh + PBZ_WPB * PBZ_SLICE);
  if constexpr (MODE == 2) { gen64_lds0_check(s_g64); load_gen64_table(s_g64, A.g64); }
  for (int i = threadIdx.x; i < PBZ_M; i += blockDim.x) s_tw[i] = A.tw[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  E* xbuf = s_x + w * D16_XELEMS;
  cpx<R>* vh = s_vh + w * PBZ_SLICE;
  LaneRegs<R, 16, 1> regs;
  GpuExec<R, 16, 1> ex{lane, regs};
  constexpr int WPB = PBZ_WPB, G = 4, LR = 128 / (int)sizeof(cpx<R>), LU = LR / G, BPG = ROWS_PER_WAVE * WPB / LU;
  const int N = A.N, SB = A.blu.SB;
  const int nbb = (A.nb + BPG - 1) / BPG;
  const int q = lane & 15, gl = lane >> 4;
  const int tiles = A.tiles ? A.tiles : (int)gridDim.x;
#pragma unroll 1
  for (int vb = blockIdx.x; vb < tiles; vb += gridDim.x) {
  const int b0 = (vb % nbb) * BPG;
  const int row0 = (vb / nbb) * LR;
#pragma unroll 1
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
    const int flat = rr * WPB + w;
    const int b = b0 + flat / LU;
    if (b >= A.nb) break;                                // wave-uniform
    const int ky0 = row0 + (flat % LU) * G;
    if (ky0 >= N) continue;                              // wave-uniform (N need not be a multiple of LR)
    const int ky = ky0 + gl;
    const bool live = ky < N;                            // the last unit of a grid may be short
    const int kyc = live ? ky : N - 1;
    const uint64_t g = A.g0 + (uint64_t)b;
    cpx<R> acc[16];
#pragma unroll
    for (int bb = 0; bb < 16; ++bb) acc[bb] = mk<R>((R)0, (R)0);
    auto acc_of = [&](int) { return acc; };
    xoshiro128p rs[4];
    if (MODE != 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rs[i] = row_stream(A.key, g, kyc, q + 16 * i, WAVE);
    }
    const float* ampf = A.ampf + (size_t)kyc * N;
    const R* amp = A.amp + (size_t)kyc * N;
    const size_t base = ((size_t)b * N + kyc) * N;
    // The wave's LDS copies are filled by LDS-DMA a step ahead of their use and retired by lds_dma_wait() where the draws' own loads
    // have been consumed: V^ of block jb at the start of block jb for the product at its end; the pre-chirp factors of block jb + 1 at
    // the start of block jb, into the other of two buffers.  A copy is never overwritten under a read of the step before: those reads
    // have returned (their values were used) before the DMA is issued.
    pbz_lds_load<2>(A.blu.pre, vh + PBZ_M, lane);        // (pre: zero beyond N, (SB + 1) * 128 entries)
    lds_dma_wait();
    ex.sync();
#pragma unroll 1
    for (int jb = 0; jb < SB; ++jb) {
      const int k0 = jb * PBZ_B + q;
      pbz_lds_load<4>(A.blu.vhat + (size_t)jb * PBZ_M, vh, lane);
      pbz_lds_load<2>(A.blu.pre + (jb + 1) * PBZ_B, vh + PBZ_M + ((jb + 1) & 1) * PBZ_B, lane);
      const cpx<R>* pre = vh + PBZ_M + (jb & 1) * PBZ_B + q;
      if constexpr (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kx = k0 + 16 * j;
          regs.v[j] = kx < N ? cmul(cscale(mk<R>((R)FMC_LDC(A.cre + base + kx), (R)FMC_LDC(A.cim + base + kx)), amp[kx]), pre[16 * j]) : mk<R>((R)0, (R)0);
        }
      } else {
        // The block's eight colouring factors in ONE batch of loads (index clamped to the row), then eight draws in straight-line
        // code: beyond the end of the row the draw still runs -- the streams belong to this row only, nothing reads them again -- and
        // the pre-chirp factor there is zero.  (With a branch per coefficient every draw waited for its own load from L2.)
        double a8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kxc = min(k0 + 16 * j, N - 1);
          a8[j] = MODE == 0 ? (double)ampf[kxc] : (double)amp[kxc];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (MODE == 0) regs.v[j] = cmul(draw_coloured<R>(rs[j & 3], (float)a8[j]), pre[16 * j]);
          else if constexpr (MODE == 2) {
            regs.v[j] = cmul(draw_coloured_f64(rs[j & 3], a8[j], Gen64Lds0{}), pre[16 * j]);
            asm volatile("" : "+v"(regs.v[j].x), "+v"(regs.v[j].y), "+v"(rs[j & 3].s0), "+v"(rs[j & 3].s1), "+v"(rs[j & 3].s2), "+v"(rs[j & 3].s3));
          }
        }
      }
#pragma unroll
      for (int j = 8; j < 16; ++j) regs.v[j] = mk<R>((R)0, (R)0);
      lds_dma_wait();
      ex.sync();
      pbz_block<R>(ex, xbuf, s_tw, vh, acc_of);
    }
    pbz_finish<R, NPL>(ex, xbuf, s_tw, acc_of);
    if (live) {
      cpx<R>* out = A.V + (size_t)b * A.Np * N + ky;
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        const int t = q + 16 * p;
        if (t < A.Np) {
          const cpx<R> pq = A.blu.post[t];
          out[(size_t)t * N] = mk<R>(pq.x * regs.v[p].x + pq.y * regs.v[p].y, pq.y * regs.v[p].x - pq.x * regs.v[p].y);   // post * conj(Y)
        }
      }
    }
  }
  if (A.tiles) __syncthreads();      // (as k_rows_wave: the waves of a workgroup stay within one tile of each other)
  }
}

// The column pass of the same grids on the same pipeline: FOUR window columns per wavefront (sixteen lanes each) in blocks of 128
// rows of V (standard layout), one pruned inverse per column, then the detector as k_cols_pks has it: the window pixels handed through
// the lane's own slots of the exchange buffer to a rolled loop, the sums reduced over the sixteen lanes of a column by DPP.
constexpr int PBZ_WPC = 8;
template <class R, int NPL, int EPI>
__global__ __launch_bounds__(PBZ_WPC * 64) void k_cols_pbz(ColArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using E = typename Xch<R>::E;
  static_assert(sizeof(R) == 8, "chirp-z grids run the float64 pipeline (fastmc_create)");
  cpx<R>* s_tw = reinterpret_cast<cpx<R>*>(smem);
  cpx<R>* s_vh = s_tw + PBZ_M;
  E* s_x = reinterpret_cast<E*>(s_vh + PBZ_WPC * PBZ_SLICE);
  for (int i = threadIdx.x; i < PBZ_M; i += blockDim.x) s_tw[i] = A.tw[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  E* xbuf = s_x + w * D16_XELEMS;
  cpx<R>* vh = s_vh + w * PBZ_SLICE;
  constexpr int G = 4;
  const int q = lane & 15, gl = lane >> 4;
  LaneRegs<R, 16, 1> regs;
  GpuExec<R, 16, 1> ex{lane, regs};
  const int N = A.N, SB = A.blu.SB;
  const int ngrp = (A.Np + G - 1) / G;
  const int item = blockIdx.x * PBZ_WPC + w;
  if (item >= A.nb * ngrp) return;                         // wave-uniform; no block barrier follows
  const int b = item / ngrp;
  const int xi = (item % ngrp) * G + gl;
  const bool live = xi < A.Np;                             // the last group of a realisation may be short
  const cpx<R>* col = A.V + ((size_t)b * A.Np + (live ? xi : (item % ngrp) * G)) * N;
  cpx<R> acc[16];
#pragma unroll
  for (int bb = 0; bb < 16; ++bb) acc[bb] = mk<R>((R)0, (R)0);
  auto acc_of = [&](int) { return acc; };
  // the loads of block jb + 1 are issued before block jb is transformed (one wavefront walks its columns block by block: without
  // this every block waits for its own loads -- the one-column-per-wave kernel has all of a column's loads in flight at once)
  cpx<R> nxt[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) nxt[j] = (q + 16 * j < N) ? load_v(col + q + 16 * j) : mk<R>((R)0, (R)0);
#pragma unroll 1
  for (int jb = 0; jb < SB; ++jb) {
    const int k0 = jb * PBZ_B + q;
    pbz_lds_load<4>(A.blu.vhat + (size_t)jb * PBZ_M, vh, lane);      // (as k_rows_pbz)
    pbz_lds_load<2>(A.blu.pre + jb * PBZ_B, vh + PBZ_M, lane);
    lds_dma_wait();                                                  // (with the block's loads of V, needed at once anyway)
    ex.sync();
    const cpx<R>* pre = vh + PBZ_M + q;
#pragma unroll
    for (int j = 0; j < 8; ++j) regs.v[j] = cmul(nxt[j], pre[16 * j]);       // (pre is zero beyond N)
#pragma unroll
    for (int j = 8; j < 16; ++j) regs.v[j] = mk<R>((R)0, (R)0);
    if (jb + 1 < SB) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ky = k0 + PBZ_B + 16 * j;
        nxt[j] = ky < N ? load_v(col + ky) : mk<R>((R)0, (R)0);
      }
    }
    pbz_block<R>(ex, xbuf, s_tw, vh, acc_of);
  }
  pbz_finish<R, NPL>(ex, xbuf, s_tw, acc_of);
  double sums[4] = {0.0, 0.0, 0.0, 0.0};
  auto pixel = [&](int yi, R p1, R p2) {
    pixel_phase<R>(A.sh, b, A.Np, yi, xi, p1, p2);
    if (EPI == 1) {
      const size_t plane = (size_t)A.Np * A.Np;
      A.phs[((size_t)b) * plane + (size_t)yi * A.Np + xi] = (double)p1;
      A.phs[((size_t)(A.nb + b)) * plane + (size_t)yi * A.Np + xi] = (double)p2;
    } else {
      const double wgt = A.W[(size_t)yi * A.Np + xi];
      double s1, c1, s2, c2;
      sincos_r(p1, s1, c1);
      sincos_r(p2, s2, c2);
      sums[0] += wgt * c1; sums[1] += wgt * s1; sums[2] += wgt * c2; sums[3] += wgt * s2;
    }
  };
  cpx<R>* ob = reinterpret_cast<cpx<R>*>(xbuf) + lane;
#pragma unroll
  for (int p = 0; p < NPL; ++p) {
    const int t = q + 16 * p;
    const cpx<R> pq = A.blu.post[t];                       // (128 entries, zero beyond the window)
    ob[WAVE * p] = mk<R>(pq.x * regs.v[p].x + pq.y * regs.v[p].y, pq.y * regs.v[p].x - pq.x * regs.v[p].y);      // post * conj(Y)
  }
  ex.sync();
#pragma unroll 1
  for (int p = 0; p < NPL; ++p) {
    const int yi = q + 16 * p;
    if (live && yi < A.Np) {
      const cpx<R> v = ob[WAVE * p];
      pixel(yi, v.x, v.y);
    }
  }
  if (EPI == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double v = sums[k];
      v += dpp_copy<0xB1>(v);     // quad_perm [1,0,3,2]
      v += dpp_copy<0x4E>(v);     // quad_perm [2,3,0,1]
      v += dpp_copy<0x141>(v);    // row_half_mirror: every lane holds the sum of its 8 lanes
      v += dpp_copy<0x140>(v);    // row_mirror: ... of its 16-lane row
      sums[k] = v;
    }
    if (q == 0 && live) {
      double* o = A.partial + ((size_t)b * A.Np + xi) * 4;
      o[0] = sums[0]; o[1] = sums[1]; o[2] = sums[2]; o[3] = sums[3];
    }
  }
}

// ================================================================== 50-lane family (N = 50 P: 100, 200, 250, 500, 1000, ...)
// Round decimal grids (fast/conf.py NPXLS 1000 etc.) on the mixed-radix row of fmc_mrfft.h: the kernels of the wave family
// with k = lane + 50 j inputs on lanes 0-49 and 50 generator streams per row (stream L = kx mod 50, fmc_core.h
// stream_lanes).  Waves per workgroup: the exchange buffer is 69 P elements (P = 20: 11 KB) and the radix-P stage holds 2 P
// values per lane, as in the wave family.
// (A/B at 800^2 f64, P = 16: 8 waves per workgroup 10.8 ms per 5000 realisations, 12 waves under the 168-VGPR cap 14.2 ms)
// The same two kernels also serve the 64-lane pipeline (LN = 64, SPLIT only): wave-family grids N = 64 P S whose sub-row
// count is not one of the compiled ones (2304 = 2 x 1152, 2560 = 2 x 1280, 3072 = 2 x 1536, ...; fmc_core.h: wave_rt_split).
template <class R, int P, int LN> struct LaneFam;
template <class R, int P> struct LaneFam<R, P, MR_LN> {
  using G = MrGeom<R, P>;
  static constexpr int L0 = G::L0, XELEMS = G::XELEMS, N = G::N;
  static __device__ __forceinline__ int osign(int n_full) { return mr_osign(n_full); }
  static constexpr int L1 = G::L1;
  template <int NS, int PR, class Exec>
  static __device__ __forceinline__ void fft(Exec& ex, typename Xch<R>::E* xbuf, const cpx<R>* tw, const cpx<R>* om, int omS, int lo,
                                             int Np, int os) {
    pruned_row_fft_mr<R, P, NS, (PR ? centre_planes(P, 10, PR == 1 ? 5 : 0) : 0x3FF)>(ex, xbuf, tw, om, omS, lo, Np, os);
  }
};
template <class R, int P> struct LaneFam<R, P, WAVE> {
  using G = WaveGeom<R, P>;
  static constexpr int L0 = 8, XELEMS = G::XELEMS, N = G::N;
  static __device__ __forceinline__ int osign(int) { return 0; }      // 64 P S is a multiple of 4
  static constexpr int L1 = 8;
  template <int NS, int PR, class Exec>
  static __device__ __forceinline__ void fft(Exec& ex, typename Xch<R>::E* xbuf, const cpx<R>* tw, const cpx<R>* om, int omS, int lo,
                                             int Np, int os) {
    pruned_row_fft<R, P, NS, (PR ? centre_planes(P, 8, 0) : 0xFF)>(ex, xbuf, tw, om, omS, lo, Np, os);
  }
};
template <class R, int P, int NS, int LN = MR_LN> struct MrCfg {
  static constexpr int W0 = WaveCfg<R, P, NS>::WPB;
  static constexpr int WPB = (LN == MR_LN && sizeof(R) == 8 && P >= 16 && W0 > 8) ? 8 : W0;
};
template <class R, int P, int NS, int LN = MR_LN>
__host__ __device__ constexpr size_t mr_lds_bytes(int omS) {
  return (size_t)(P * WAVE + LaneFam<R, P, LN>::L0 * omS) * sizeof(cpx<R>) + (size_t)MrCfg<R, P, NS, LN>::WPB * LaneFam<R, P, LN>::XELEMS * 8;
}
template <class R, int P, int LN>
__device__ __forceinline__ void load_tables_mr(cpx<R>* s_tw, cpx<R>* s_om, const cpx<R>* tw, const cpx<R>* om, int omS) {
  for (int i = threadIdx.x; i < P * WAVE; i += blockDim.x) s_tw[i] = tw[i];
  for (int i = threadIdx.x; i < LaneFam<R, P, LN>::L0 * omS; i += blockDim.x) s_om[i] = om[i];
  __syncthreads();
}

// SPLIT: the row of N = S * 50 P points as S interleaved sub-rows (kx = s mod S, S = N / 50 P at run time, <= 5), window
// outputs combined by decimation in time, X[x] = sum_s w_N^{s x} Y_s[x mod 50 P] (cw), as the wave family does for 2048 / 4096.
// PR: planes of the exchange-2 image the kernel keeps: 0 all; 1 / 2 those of a centred window of up to 96 pixels whose centre
// block has residue 5 / 0 (50 P S with S odd / S even, and every 64 P S grid): the others are neither stored nor computed.
template <class R, int P, int NS, int MODE, bool SPLIT = false, int LN = MR_LN, int PR = 0>
__global__ __launch_bounds__((MrCfg<R, P, NS, LN>::WPB * 64)) void k_rows_mr(RowArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using G = LaneFam<R, P, LN>;
  using E = typename Xch<R>::E;
  // MODE 2 (float64 generator fused into the rows, as in k_rows_wave): its 4 KB of tables at the start of the LDS
  Gen64Entry* s_g64 = reinterpret_cast<Gen64Entry*>(smem);
  cpx<R>* s_tw = reinterpret_cast<cpx<R>*>(smem + (MODE == 2 ? GEN64_TABLE_BYTES : 0));
  cpx<R>* s_om = s_tw + P * WAVE;
  E* s_x = reinterpret_cast<E*>(s_om + G::L0 * A.omS);
  if constexpr (MODE == 2) { gen64_lds0_check(s_g64); load_gen64_table(s_g64, A.g64); }
  load_tables_mr<R, P, LN>(s_tw, s_om, A.tw, A.om, A.omS);

  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane < LN ? lane : LN - 1;            // 50 lanes: idle lanes repeat lane 49 (in-bounds loads; nothing reads their column)
  E* xbuf = s_x + w * G::XELEMS;
  const int N = SPLIT ? A.N : G::N;
  const int S = SPLIT ? A.N / G::N : 1;
  LaneRegs<R, P, NS> regs;
  GpuExec<R, P, NS> ex{lane, regs};
  constexpr int WPB = MrCfg<R, P, NS, LN>::WPB;
  constexpr int LR = 128 / (int)sizeof(cpx<R>);
  static_assert((ROWS_PER_WAVE * WPB) % LR == 0, "tile must hold whole lines");
  const int rpw = A.rpw ? A.rpw : ROWS_PER_WAVE;         // (launch: pick_rpw)
  const int BPG = rpw * WPB / LR;
  const int nbb = (A.nb + BPG - 1) / BPG;
  const int b0 = (blockIdx.x % nbb) * BPG;
  const int row0 = (blockIdx.x / nbb) * LR;
  for (int rr = 0; rr < rpw; ++rr) {
    const int flat = rr * WPB + w;
    const int b = b0 + flat / LR;
    if (b >= A.nb) break;                                // wave-uniform
    const int ky = row0 + flat % LR;
    if (ky >= N) continue;                               // wave-uniform (N need not be a multiple of LR)
    const uint64_t g = A.g0 + (uint64_t)b;
    R accr[NS], acci[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) { accr[s2] = (R)0; acci[s2] = (R)0; }
#pragma unroll 1
    for (int sp = 0; sp < S; ++sp) {
      if (MODE == 0) {
        const float* ampf = A.ampf + (size_t)ky * N + sp;
        xoshiro128p rs = row_stream(A.key, g, ky, sp + S * li, LN * S);
#pragma unroll
        for (int j = 0; j < P; ++j) regs.v[j] = draw_coloured<R>(rs, ampf[S * (li + LN * j)]);
      } else if constexpr (MODE == 2) {
        // the generator at the reference's precision, one coefficient at a time (see k_rows_wave)
        static_assert(sizeof(R) == 8, "the float64 generator feeds the float64 pipeline");
        const R* amp = A.amp + (size_t)ky * N + sp;
        xoshiro128p rs = row_stream(A.key, g, ky, sp + S * li, LN * S);
        double an = (double)amp[S * li];
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const double a = an;
          if (j + 1 < P) an = (double)amp[S * (li + LN * (j + 1))];
          ex.loadfence();
          regs.v[j] = draw_coloured_f64(rs, a, Gen64Lds0{});
          asm volatile("" : "+v"(regs.v[j].x), "+v"(regs.v[j].y), "+v"(rs.s0), "+v"(rs.s1), "+v"(rs.s2), "+v"(rs.s3));
        }
      } else {
        const size_t base = ((size_t)b * N + ky) * N + sp;
        const R* amp = A.amp + (size_t)ky * N + sp;
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const int kx = S * (li + LN * j);
          regs.v[j] = cscale(mk<R>((R)FMC_LDC(A.cre + base + kx), (R)FMC_LDC(A.cim + base + kx)), amp[kx]);
        }
      }
      G::template fft<NS, PR>(ex, xbuf, s_tw, s_om, A.omS, A.lo, A.Np, G::osign(N));
      if (SPLIT) {
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
          const int oi = lane + WAVE * s2;
          if (oi < A.Np) {
            const cpx<R> c = A.cw[sp * A.omS + oi];
            accr[s2] += c.x * regs.xr[s2] - c.y * regs.xi[s2];
            acci[s2] += c.x * regs.xi[s2] + c.y * regs.xr[s2];
          }
        }
      }
    }
    if (SPLIT) {
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) { regs.xr[s2] = accr[s2]; regs.xi[s2] = acci[s2]; }
    }
    cpx<R>* out = A.V + (size_t)b * A.Np * N + ky;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int oi = lane + WAVE * s;
      if (oi < A.Np) out[(size_t)oi * N] = mk<R>(regs.xr[s], regs.xi[s]);
    }
  }
}

template <class R, int P, int NS, int EPI, bool SPLIT = false, int LN = MR_LN, int PR = 0>
__global__ __launch_bounds__((MrCfg<R, P, NS, LN>::WPB * 64)) void k_cols_mr(ColArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using G = LaneFam<R, P, LN>;
  using E = typename Xch<R>::E;
  cpx<R>* s_tw = reinterpret_cast<cpx<R>*>(smem);
  cpx<R>* s_om = s_tw + P * WAVE;
  E* s_x = reinterpret_cast<E*>(s_om + G::L0 * A.omS);
  load_tables_mr<R, P, LN>(s_tw, s_om, A.tw, A.om, A.omS);

  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane < LN ? lane : LN - 1;
  E* xbuf = s_x + w * G::XELEMS;
  const int item = blockIdx.x * MrCfg<R, P, NS, LN>::WPB + w;
  if (item >= A.nb * A.Np) return;   // whole wave exits; no block barrier follows
  const int b = item / A.Np;
  const int xi = item % A.Np;
  const int N = SPLIT ? A.N : G::N;
  const int S = SPLIT ? A.N / G::N : 1;
  LaneRegs<R, P, NS> regs;
  GpuExec<R, P, NS> ex{lane, regs};
  const cpx<R>* col = A.V + ((size_t)b * A.Np + xi) * N;
  if (!SPLIT) {
#pragma unroll
    for (int j = 0; j < P; ++j) regs.v[j] = col[li + LN * j];
    G::template fft<NS, PR>(ex, xbuf, s_tw, s_om, A.omS, A.lo, A.Np, G::osign(N));
  } else {
    R accr[NS], acci[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) { accr[s2] = (R)0; acci[s2] = (R)0; }
#pragma unroll 1
    for (int sp = 0; sp < S; ++sp) {
#pragma unroll
      for (int j = 0; j < P; ++j) regs.v[j] = col[sp + S * (li + LN * j)];
      G::template fft<NS, PR>(ex, xbuf, s_tw, s_om, A.omS, A.lo, A.Np, G::osign(N));
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        const int oi = lane + WAVE * s2;
        if (oi < A.Np) {
          const cpx<R> c = A.cw[sp * A.omS + oi];
          accr[s2] += c.x * regs.xr[s2] - c.y * regs.xi[s2];
          acci[s2] += c.x * regs.xi[s2] + c.y * regs.xr[s2];
        }
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) { regs.xr[s2] = accr[s2]; regs.xi[s2] = acci[s2]; }
  }
  column_epilogue<R, NS, EPI>(A.sh, A.W, A.partial, A.phs, A.nb, A.Np, b, xi, lane, regs.xr, regs.xi);
}

// ================================================================== direct family (any N <= 4096)
constexpr int DIRECT_THREADS = 256;
constexpr int DIRECT_RESYNC = 16;   // terms between exact twiddle re-reads in the direct kernels

template <class R, int MODE>
__global__ __launch_bounds__(DIRECT_THREADS) void k_rows_direct(RowArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = A.N;
  // [twiddles N][row N][partials]; with tw_global (grids whose 3 N complex do not fit the LDS, e.g. a
  // full-window transform at 4096) the table stays in global memory / L2
  cpx<R>* s_lds = reinterpret_cast<cpx<R>*>(smem);
  const cpx<R>* s_tw = A.tw_global ? A.tw : s_lds;
  cpx<R>* s_row = s_lds + (A.tw_global ? 0 : N);
  const int b = blockIdx.x % A.nb;
  const int ky = blockIdx.x / A.nb;
  const uint64_t g = A.g0 + (uint64_t)b;
  const R* amp = A.amp + (size_t)ky * N;
  if (!A.tw_global)
    for (int i = threadIdx.x; i < N; i += blockDim.x) s_lds[i] = A.tw[i];
  if (MODE == 0) {
    const int SL = stream_lanes(N);
    for (int L = threadIdx.x; L < SL && L < N; L += blockDim.x) {   // one sequential stream per stream index (SL > blockDim beyond 4096)
      xoshiro128p rs = row_stream(A.key, g, ky, L, SL);
      for (int kx = L; kx < N; kx += SL) s_row[kx] = draw_coloured<R>(rs, A.ampf[(size_t)ky * N + kx]);
    }
  } else {
    const size_t base = ((size_t)b * N + ky) * N;
    for (int kx = threadIdx.x; kx < N; kx += blockDim.x)
      s_row[kx] = cscale(mk<R>((R)FMC_LDC(A.cre + base + kx), (R)FMC_LDC(A.cim + base + kx)), amp[kx]);
  }
  __syncthreads();
  // Every window output is a length-N sum; with Np < blockDim the sum is cut into S segments so
  // that all threads work, and the S partials are added in a fixed order (deterministic).
  const int h = N / 2;
  const int S = (A.Np < (int)blockDim.x) ? (int)blockDim.x / A.Np : 1;
  const int K = (N + S - 1) / S;
  cpx<R>* s_part = s_row + N;                       // [S][Np]
  for (int item = threadIdx.x; item < S * A.Np; item += blockDim.x) {
    const int oi = item % A.Np, seg = item / A.Np;
    const int k0 = seg * K, k1 = min(N, k0 + K);
    const int q = shifted_exponent_step(A.lo + oi, N);
    int e = (int)(((long long)q * (h + k0)) % N);
    cpx<R> acc = mk<R>((R)0, (R)0);
    // running twiddle w_N^{e_k} by recurrence (no bank-conflicting table look-up per term), re-read
    // exactly from the table every DIRECT_RESYNC terms so that rounding does not accumulate
    const int e_step = (int)(((long long)q * DIRECT_RESYNC) % N);
    const cpx<R> wq = s_tw[q];
    for (int kb = k0; kb < k1; kb += DIRECT_RESYNC) {
      cpx<R> wk = s_tw[e];
      const int kend = min(k1, kb + DIRECT_RESYNC);
      for (int k = kb; k < kend; ++k) {
        acc = cfma(s_row[k], wk, acc);
        wk = cmul(wk, wq);
      }
      e += e_step;
      if (e >= N) e -= N;
    }
    s_part[seg * A.Np + oi] = acc;
  }
  __syncthreads();
  for (int oi = threadIdx.x; oi < A.Np; oi += blockDim.x) {
    cpx<R> acc = s_part[oi];
    for (int seg = 1; seg < S; ++seg) acc = acc + s_part[seg * A.Np + oi];
    A.V[((size_t)b * A.Np + oi) * N + ky] = acc;
  }
}

template <class R, int EPI>
__global__ __launch_bounds__(DIRECT_THREADS) void k_cols_direct(ColArgs<R> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double s_red[4][DIRECT_THREADS / 64];
  const int N = A.N;
  cpx<R>* s_lds = reinterpret_cast<cpx<R>*>(smem);
  const cpx<R>* s_tw = A.tw_global ? A.tw : s_lds;
  cpx<R>* s_col = s_lds + (A.tw_global ? 0 : N);
  const int b = blockIdx.x / A.Np;
  const int xi = blockIdx.x % A.Np;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    if (!A.tw_global) s_lds[i] = A.tw[i];
    s_col[i] = A.V[((size_t)b * A.Np + xi) * N + i];
  }
  __syncthreads();
  const int h = N / 2;
  const int S = (A.Np < (int)blockDim.x) ? (int)blockDim.x / A.Np : 1;
  const int K = (N + S - 1) / S;
  cpx<R>* s_part = s_col + N;                       // [S][Np]
  for (int item = threadIdx.x; item < S * A.Np; item += blockDim.x) {
    const int yi = item % A.Np, seg = item / A.Np;
    const int k0 = seg * K, k1 = min(N, k0 + K);
    const int q = shifted_exponent_step(A.lo + yi, N);
    int e = (int)(((long long)q * (h + k0)) % N);
    cpx<R> part = mk<R>((R)0, (R)0);
    const int e_step = (int)(((long long)q * DIRECT_RESYNC) % N);
    const cpx<R> wq = s_tw[q];
    for (int kb = k0; kb < k1; kb += DIRECT_RESYNC) {
      cpx<R> wk = s_tw[e];
      const int kend = min(k1, kb + DIRECT_RESYNC);
      for (int k = kb; k < kend; ++k) {
        part = cfma(s_col[k], wk, part);
        wk = cmul(wk, wq);
      }
      e += e_step;
      if (e >= N) e -= N;
    }
    s_part[seg * A.Np + yi] = part;
  }
  __syncthreads();
  double acc4[4] = {0.0, 0.0, 0.0, 0.0};
  for (int yi = threadIdx.x; yi < A.Np; yi += blockDim.x) {
    cpx<R> acc = s_part[yi];
    for (int seg = 1; seg < S; ++seg) acc = acc + s_part[seg * A.Np + yi];
    R p1 = acc.x, p2 = acc.y;
    pixel_phase<R>(A.sh, b, A.Np, yi, xi, p1, p2);
    if (EPI == 1) {
      const size_t plane = (size_t)A.Np * A.Np;
      A.phs[((size_t)b) * plane + (size_t)yi * A.Np + xi] = (double)p1;
      A.phs[((size_t)(A.nb + b)) * plane + (size_t)yi * A.Np + xi] = (double)p2;
    } else {
      const double wgt = A.W[(size_t)yi * A.Np + xi];
      double s1, c1, s2, c2;
      sincos_r(p1, s1, c1);
      sincos_r(p2, s2, c2);
      acc4[0] += wgt * c1; acc4[1] += wgt * s1; acc4[2] += wgt * c2; acc4[3] += wgt * s2;
    }
  }
  if (EPI == 0) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double v = wave_sum(acc4[q]);
      if (lane == 0) s_red[q][w] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      double v = 0.0;
      for (int i = 0; i < DIRECT_THREADS / 64; ++i) v += s_red[threadIdx.x][i];
      A.partial[((size_t)b * A.Np + xi) * 4 + threadIdx.x] = v;
    }
  }
}

// ================================================================== spectrum -> colouring amplitudes
// amp = sqrt(powerspec) * df (fast/fast.py:594 and the `rand * df` of funcs.py:213); amp_s carries the
// input-side fftshift sign (-1)^(ky+kx).  bad[0] counts entries that are negative, NaN or infinite.
// ampf / ampf_s: the same two tables times sqrt(2 ln 2), rounded to float32: colouring of the device generator's draws.
// amp_p / ampf_p (grids of the packed sub-rows, else null): amp_s / ampf_s with every row stored SUB-ROW MAJOR -- entry kx = s + S m at
// s M + m -- so that pass s of k_rows_pks reads M consecutive values of each of its rows.  (Read in place, a pass touches every
// 128-byte line of the row for an S-th of its bytes: S N^2 x 8 bytes of L1 fills per realisation, 35-42 TB/s at 1792 / 3840 -- the
// 64 bytes per clock and CU the vector L1 can fill, and the bound of those rows until round 6.)
template <class R>
__global__ void k_make_amp(const double* ps, double df, int N, R* amp, R* amp_s, float* ampf, float* ampf_s, unsigned int* bad,
                           R* amp_p, float* ampf_p, int S) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * N) return;
  const double p = ps[i];
  if (!(p >= 0.0) || isinf(p)) { atomicAdd(bad, 1u); return; }
  const double v = sqrt(p) * df;
  const int ky = (int)(i / N), kx = (int)(i % N);
  amp[i] = (R)v;
  amp_s[i] = (R)(((ky + kx) & 1) ? -v : v);
  const float vk = (float)(v * 1.1774100225154746910);      // * sqrt(2 ln 2): see box_muller_scaled
  ampf[i] = vk;
  ampf_s[i] = ((ky + kx) & 1) ? -vk : vk;
  if (amp_p) {
    const size_t j = (size_t)ky * N + (size_t)(kx % S) * (N / S) + kx / S;
    amp_p[j] = (R)(((ky + kx) & 1) ? -v : v);
    ampf_p[j] = ((ky + kx) & 1) ? -vk : vk;
  }
}

#if FMC_TU == 0   // non-template kernels: one definition in the library
// ================================================================== sub-harmonic coefficients
// coef[b][m] = rand_lo[b][m] * sqrt(ps_lo[m]) * df_lo[level(m)]  (fast/fast.py:600-601, funcs.py:243)
// mean[b]    = sum_m coef[b][m] * mu[m],  mu[m] = grid mean of mode m (funcs.py:253)
struct ShCoefArgs {
  int nb;
  RngKey key;
  uint64_t g0;
  const double* sh_re;   // host mode: [nb][27] or NULL -> device RNG
  const double* sh_im;
  const double* scale;   // [27] sqrt(ps_lo) * df_lo
  const double* mu;      // [27][2]
  double* coef;          // [nb][27][2]
  double* mean;          // [nb][2]
  int rng_f64;           // device draws at float64 precision (box_muller_f64; low bits from blocks of STREAM_SUBHARM_LO)
};

__global__ void k_subharm_coeffs(ShCoefArgs A) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= A.nb) return;
  const uint64_t g = A.g0 + (uint64_t)b;
  double mr = 0.0, mi = 0.0;
  double cr[27], ci[27];
  if (A.sh_re) {
    for (int m = 0; m < 27; ++m) { cr[m] = A.sh_re[b * 27 + m]; ci[m] = A.sh_im[b * 27 + m]; }
  } else {
    for (int mp = 0; mp < 14; ++mp) {   // pairs (mp, mp + 14), H = ceil(27/2)
      const u32x4 x = philox4x32_10((uint32_t)mp, STREAM_SUBHARM, (uint32_t)g, (uint32_t)(g >> 32), A.key.k0, A.key.k1);
      if (A.rng_f64) {
        const u32x4 y = philox4x32_10((uint32_t)mp, STREAM_SUBHARM_LO, (uint32_t)g, (uint32_t)(g >> 32), A.key.k0, A.key.k1);
        double a, bb, c, d;
        box_muller_f64(x.a, x.b, y.a, y.b, a, bb);
        box_muller_f64(x.c, x.d, y.c, y.d, c, d);
        cr[mp] = a; ci[mp] = bb;
        if (mp + 14 < 27) { cr[mp + 14] = c; ci[mp + 14] = d; }
        continue;
      }
      float a, bb, c, d;
      box_muller(x.a, x.b, a, bb);
      box_muller(x.c, x.d, c, d);
      cr[mp] = a; ci[mp] = bb;
      if (mp + 14 < 27) { cr[mp + 14] = c; ci[mp + 14] = d; }
    }
  }
  for (int m = 0; m < 27; ++m) {
    const double r = cr[m] * A.scale[m], i = ci[m] * A.scale[m];
    A.coef[(b * 27 + m) * 2] = r;
    A.coef[(b * 27 + m) * 2 + 1] = i;
    mr += r * A.mu[2 * m] - i * A.mu[2 * m + 1];
    mi += r * A.mu[2 * m + 1] + i * A.mu[2 * m];
  }
  A.mean[b * 2] = mr;
  A.mean[b * 2 + 1] = mi;
}

// d_{p,i}(b, x) = sum_j c_{p,i,j}(b) exp(i x fx_{p,j}) for every window column x: one thread per (b, x).
__global__ void k_subharm_cols(const double* coef, const double* ex, int nb, int Np, double* dcol) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb * Np) return;
  const int b = t / Np, xi = t % Np;
  const double* cf = coef + (size_t)b * 54;
  for (int l = 0; l < 9; ++l) {
    double dr = 0.0, di = 0.0;
    for (int j = 0; j < 3; ++j) {
      const int m = 9 * (l / 3) + 3 * (l % 3) + j;
      const double exr = ex[(m * Np + xi) * 2], exi = ex[(m * Np + xi) * 2 + 1];
      dr += cf[2 * m] * exr - cf[2 * m + 1] * exi;
      di += cf[2 * m] * exi + cf[2 * m + 1] * exr;
    }
    dcol[(size_t)t * 18 + 2 * l] = dr;
    dcol[(size_t)t * 18 + 2 * l + 1] = di;
  }
}

// ================================================================== finalize (fast/fast.py:647-668)
struct FinArgs {
  int nb, Np, coherent;
  int64_t n_real, j0;         // total realisations of the run, index of this launch's first one
  const double* partial;      // [nb][Np][4]
  const double* logamp;       // [2*n_real] in output order, or NULL -> device draw
  double logamp_sigma;        // sqrt(logamp_var)
  int rng_f64;                // device draw at float64 precision
  RngKey key;
  uint64_t g0;                // global realisation index of j0
  double dx2, norm;           // dx^2, sum(W) * dx^2
  double* out;                // [2*n_real] or [2*n_real][2]
};

// One wavefront per realisation: lanes stride over the Np column partials (32 contiguous bytes
// each), fixed-shape shuffle tree -> deterministic.
__global__ __launch_bounds__(256) void k_finalize(FinArgs A) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= A.nb) return;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  const double* p = A.partial + (size_t)b * A.Np * 4;
  for (int xi = lane; xi < A.Np; xi += 64)
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] += p[xi * 4 + q];
#pragma unroll
  for (int q = 0; q < 4; ++q) s[q] = wave_sum(s[q]);
  if (lane >= 2) return;
  const int part = lane;
  const int64_t j = A.j0 + b;
  const int64_t o = part * A.n_real + j;
  double chi;
  if (A.logamp) chi = A.logamp[o];
  else if (A.rng_f64) chi = draw_logamp_normal_f64(A.key, 2 * (A.g0 + (uint64_t)b) + part) * A.logamp_sigma;
  else chi = (double)draw_logamp_normal(A.key, 2 * (A.g0 + (uint64_t)b) + part) * A.logamp_sigma;
  const double e = exp(chi);
  const double ar = (e * (s[2 * part] * A.dx2)) / A.norm;
  const double ai = (e * (s[2 * part + 1] * A.dx2)) / A.norm;
  if (A.coherent) { A.out[2 * o] = ar; A.out[2 * o + 1] = ai; }
  else A.out[o] = ar * ar + ai * ai;
}

// ================================================================== frozen-flow time series
// Fast.compute_phs_temporal (fast/fast.py:619-633) + compute_detector (647-668) for one chunk of M
// time steps: every layer's N x N screen is sampled bilinearly (RectBivariateSpline kx=ky=1: FITPACK
// clamps arguments to the knot range [0, N-1]) on the wind-shifted, wrapped and sorted pupil grid,
// rolled back (numpy.roll by -shift), summed over layers, then W exp(i phi) is reduced.
struct TemporalArgs {
  int N, Np, L, M, coherent;
  const double* screens;   // [L][N][N]
  const double* xs;        // [L][M][Np] sorted row coordinates
  const double* ys;        // [L][M][Np] sorted column coordinates
  const int* roll;         // [L][2][M]
  const double* W;         // [Np][Np]
  const double* logamp;    // [M]
  double dx2, norm;
  double* out;             // [M] or [M][2]; NULL: phases only
  double* phs;             // [M][Np][Np] the summed, shifted phase of every time step (Fast.phs, fast/fast.py:633), or NULL
};

__device__ __forceinline__ void bilinear_cell(double x, int N, int& i, double& t) {
  if (x < 0.0) x = 0.0;
  if (x > (double)(N - 1)) x = (double)(N - 1);
  i = (int)floor(x);
  if (i > N - 2) i = N - 2;
  t = x - (double)i;
}

__global__ __launch_bounds__(256) void k_temporal_detect(TemporalArgs A) {
  __shared__ double s_red[2][4];
  const int j = blockIdx.x;
  const int Np = A.Np, N = A.N;
  double sr = 0.0, si = 0.0;
  for (int pix = threadIdx.x; pix < Np * Np; pix += blockDim.x) {
    const int a = pix / Np, b = pix % Np;
    double phi = 0.0;
    for (int l = 0; l < A.L; ++l) {
      const int ra = (a + A.roll[(l * 2 + 0) * A.M + j]) % Np;
      const int rb = (b + A.roll[(l * 2 + 1) * A.M + j]) % Np;
      int i0, j0;
      double t, u;
      bilinear_cell(A.xs[((size_t)l * A.M + j) * Np + ra], N, i0, t);
      bilinear_cell(A.ys[((size_t)l * A.M + j) * Np + rb], N, j0, u);
      const double* z = A.screens + (size_t)l * N * N + (size_t)i0 * N + j0;
      phi += (1 - t) * ((1 - u) * z[0] + u * z[1]) + t * ((1 - u) * z[N] + u * z[N + 1]);
    }
    if (A.phs) A.phs[(size_t)j * Np * Np + pix] = phi;
    double s, c;
    sincos_r(phi, s, c);
    const double w = A.W[pix];
    sr += w * c;
    si += w * s;
  }
  if (!A.out) return;          // phases only (uniform over the block)
  sr = wave_sum(sr);
  si = wave_sum(si);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { s_red[0][wv] = sr; s_red[1][wv] = si; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tr = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    const double ti = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    const double e = exp(A.logamp[j]);
    const double ar = (e * (tr * A.dx2)) / A.norm, ai = (e * (ti * A.dx2)) / A.norm;
    if (A.coherent) { A.out[2 * j] = ar; A.out[2 * j + 1] = ai; }
    else A.out[j] = ar * ar + ai * ai;
  }
}

// ================================================================== histogram of dB_rel
__global__ void k_histogram(const double* out, int64_t n, int coherent, double lo, double hi, int nbins,
                            unsigned long long* bins) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double pw = coherent ? out[2 * i] * out[2 * i] + out[2 * i + 1] * out[2 * i + 1] : out[i];
  const double db = 10.0 * log10(pw);
  int k;
  if (!(db >= lo)) k = nbins;            // underflow (and NaN)
  else if (db >= hi) k = nbins + 1;      // overflow
  else { k = (int)((db - lo) / (hi - lo) * nbins); if (k >= nbins) k = nbins - 1; }
  atomicAdd(&bins[k], 1ULL);
}

// ================================================================== result statistics on the device
// FastResult.avg_power / scintillation_index (fast/fast.py:965-983) and the fade probability of
// comms.fade_prob (fast/comms.py:171-177) without moving the per-iteration vector to the host.
// Deterministic two-stage reduction: block partials, then one thread per quantity.
constexpr int STATS_MAX_THR = 16;
constexpr int STATS_NQ = 5;   // sum r, sum r^2, sum 10log10 r, min r, max r  (+ counts below thresholds)

__global__ __launch_bounds__(256) void k_stats_partial(const double* out, int64_t n, int coherent, const double* thr,
                                                       int n_thr, double* partial) {
  __shared__ double s_q[4][STATS_NQ + STATS_MAX_THR];
  double q[STATS_NQ + STATS_MAX_THR];
  for (int i = 0; i < STATS_NQ + STATS_MAX_THR; ++i) q[i] = 0.0;
  q[3] = INFINITY; q[4] = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double r = coherent ? out[2 * i] * out[2 * i] + out[2 * i + 1] * out[2 * i + 1] : out[i];
    q[0] += r; q[1] += r * r; q[2] += 10.0 * log10(r);
    q[3] = fmin(q[3], r); q[4] = fmax(q[4], r);
    for (int t = 0; t < n_thr; ++t) q[STATS_NQ + t] += (r < thr[t]) ? 1.0 : 0.0;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = 0; i < STATS_NQ + n_thr; ++i) {
    double v = q[i];
    for (int o = 32; o >= 1; o >>= 1) {
      const double u = __shfl_xor(v, o, 64);
      v = (i == 3) ? fmin(v, u) : (i == 4) ? fmax(v, u) : v + u;
    }
    if (lane == 0) s_q[wv][i] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < STATS_NQ + n_thr; i += blockDim.x) {
    double v = s_q[0][i];
    for (int w2 = 1; w2 < 4; ++w2) v = (i == 3) ? fmin(v, s_q[w2][i]) : (i == 4) ? fmax(v, s_q[w2][i]) : v + s_q[w2][i];
    partial[(size_t)blockIdx.x * (STATS_NQ + STATS_MAX_THR) + i] = v;
  }
}

__global__ void k_stats_final(const double* partial, int nblocks, int nq, double* result) {
  const int i = threadIdx.x;
  if (i >= nq) return;
  double v = partial[i];
  for (int b = 1; b < nblocks; ++b) {
    const double u = partial[(size_t)b * (STATS_NQ + STATS_MAX_THR) + i];
    v = (i == 3) ? fmin(v, u) : (i == 4) ? fmax(v, u) : v + u;
  }
  result[i] = v;
}

// ================================================================== link metrics (fast/comms.py:171-262)
// One launch per query; every block leaves 4 partial values combined by k_link_final with the
// operators (+, +, min, max).  `mean` is read from the statistics pass (sum r / n).
constexpr int LM_FADE = 0, LM_BER_OOK = 1, LM_SEP_QAM = 2;

__device__ __forceinline__ double link_sample(const double* x, int64_t i, int coherent) {
  return coherent ? x[2 * i] * x[2 * i] + x[2 * i + 1] * x[2 * i + 1] : x[i];
}

__device__ __forceinline__ double q_function(double v) { return 0.5 * erfc(v * 0.70710678118654752440); }

__global__ __launch_bounds__(256) void k_link_query(const double* x, int64_t n, int coherent, const double* sum_r, int kind,
                                                    double p0, double p1, double* partial) {
  __shared__ double s_q[4][4];
  double q0 = 0.0, q1 = 0.0, q2 = (double)n, q3 = -1.0;
  const double inv_mean = (double)n / sum_r[0];
  double a = 0.0, c = 0.0;
  if (kind == LM_BER_OOK) a = sqrt(pow(10.0, p0 / 10.0));
  if (kind == LM_SEP_QAM) {
    a = 3.0 / (p0 - 1.0) * pow(10.0, p1 / 10.0);
    c = (sqrt(p0) - 1.0) / sqrt(p0);
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double r = link_sample(x, i, coherent);
    if (kind == LM_FADE) {
      const bool below = r < p0;
      if (below) {
        q0 += 1.0;
        if (i > 0 && !(link_sample(x, i - 1, coherent) < p0)) q1 += 1.0;
      } else {
        q2 = fmin(q2, (double)i);
        q3 = fmax(q3, (double)i);
      }
    } else if (kind == LM_BER_OOK) {
      q0 += q_function(r * inv_mean * a);
    } else {
      const double s = r * inv_mean;
      const double q = q_function(sqrt(a * s * s));
      q0 += 4.0 * (c * q - c * c * q * q);
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int o = 32; o >= 1; o >>= 1) {
    q0 += __shfl_xor(q0, o, 64);
    q1 += __shfl_xor(q1, o, 64);
    q2 = fmin(q2, __shfl_xor(q2, o, 64));
    q3 = fmax(q3, __shfl_xor(q3, o, 64));
  }
  if (lane == 0) { s_q[wv][0] = q0; s_q[wv][1] = q1; s_q[wv][2] = q2; s_q[wv][3] = q3; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w2 = 1; w2 < 4; ++w2) {
      q0 += s_q[w2][0]; q1 += s_q[w2][1]; q2 = fmin(q2, s_q[w2][2]); q3 = fmax(q3, s_q[w2][3]);
    }
    double* o = partial + 4 * (size_t)blockIdx.x;
    o[0] = q0; o[1] = q1; o[2] = q2; o[3] = q3;
  }
}

__global__ void k_link_final(const double* partial, int nblocks, int64_t n, const double* sum_r, int kind, double* result) {
  if (threadIdx.x != 0) return;
  double q0 = 0.0, q1 = 0.0, q2 = (double)n, q3 = -1.0;
  for (int b = 0; b < nblocks; ++b) {       // fixed order: the sums do not depend on scheduling
    q0 += partial[4 * b]; q1 += partial[4 * b + 1];
    q2 = fmin(q2, partial[4 * b + 2]); q3 = fmax(q3, partial[4 * b + 3]);
  }
  if (kind == LM_FADE) { result[0] = q0; result[1] = q1; result[2] = q2; result[3] = q3; }
  else { result[0] = q0; result[1] = sum_r[0] / (double)n; result[2] = (double)n; result[3] = 0.0; }
}

// ================================================================== generator read-back (parity tests)
__global__ void k_rng_coeffs(RngKey key, uint64_t g, int N, int rng_f64, const Gen64Entry* g64, double* out) {
  const int SL = stream_lanes(N);
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * SL) return;
  const int ky = idx / SL, l = idx % SL;
  if (l >= N) return;
  xoshiro128p rs = row_stream(key, g, ky, l, SL);
  for (int kx = l; kx < N; kx += SL) {
    cpx<double> c;
    if (rng_f64) {
      c = draw_coloured_f64(rs, 1.0, g64);     // the arithmetic of the fused rows (MODE 2), bit for bit
    } else {
      c = draw_coeff<double>(rs);
    }
    out[2 * ((size_t)ky * N + kx)] = c.x;
    out[2 * ((size_t)ky * N + kx) + 1] = c.y;
  }
}

// GPU_RNG_PRECISION 'f64': the coefficients of nb realisations at float64 precision, written where host-coefficient mode
// keeps its uploads ([nb][N][N] real parts, imaginary parts); the MODE 1 kernels then colour them in float64 as the
// reference does (fast/fast.py:594).  One thread per stream, neighbouring threads neighbouring columns.  Only the kernel
// families without a fused form (MODE 2 of the row kernels) take this detour through HBM.
__global__ __launch_bounds__(256) void k_gen_coeffs_f64(RngKey key, uint64_t g0, int nb, int N, const Gen64Entry* g64, double* cre, double* cim) {
  __shared__ Gen64Entry s_g64[GEN64_LOG_ENTRIES + GEN64_TRIG_ENTRIES];
  load_gen64_table(s_g64, g64);
  __syncthreads();
  const int SL = stream_lanes(N);
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)nb * N * SL) return;
  const int l = (int)(idx % SL), ky = (int)((idx / SL) % N), b = (int)(idx / ((int64_t)SL * N));
  if (l >= N) return;
  const uint64_t g = g0 + (uint64_t)b;
  xoshiro128p rs = row_stream(key, g, ky, l, SL);
  const size_t base = ((size_t)b * N + ky) * N;
  for (int kx = l; kx < N; kx += SL) {
    const cpx<double> c = draw_coloured_f64(rs, 1.0, s_g64);
    cre[base + kx] = c.x;
    cim[base + kx] = c.y;
  }
}

__global__ void k_rng_logamp(RngKey key, uint64_t it0, int64_t n, int rng_f64, double* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = rng_f64 ? draw_logamp_normal_f64(key, it0 + (uint64_t)i) : (double)draw_logamp_normal(key, it0 + (uint64_t)i);
}

#endif   // FMC_TU == 0

}  // namespace fmc
