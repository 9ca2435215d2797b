// fmc_wavefft.h -- one wavefront = one N-point row: output-pruned forward DFT, N = 64*P (and, further down, the packed
// rows of the small grids: 8 / 4 / 2 rows of 128 / 256 / 512 points per wavefront).
//
// A lane holds P inputs in registers (k = lane + 64*j), P = 2^k times 1, 3, 5, 7 or 9, <= 32.  The transform is factored
//     N = P x 8 x 8 :  in-register radix-P  ->  LDS exchange  ->  in-register radix-8
//                      ->  LDS exchange  ->  8-term sums for the WANTED outputs only
// because FAST keeps only the Np x Np pupil window of every N x N screen
// (fast/fast.py:390,596): outputs lo .. lo+Np-1.  Output slot s of lane l is window
// index oi = l + 64*s.
//
//   X[x] = sum_k c[k] w_N^{kx},  k = l + 64 j,  x = a + P b  (a < P, b < 64)
//        = sum_l w_N^{l a} w_64^{l b} Z_l[a],          Z_l[a] = sum_j c[l+64j] w_P^{ja}   (stage 1)
//   l = l0 + 8 l1, b = b0 + 8 b1:
//        = sum_l0 w_64^{l0 b} U[a][l0][b0],   U[a][l0][b0] = sum_l1 w_8^{l1 b0} T_{l0+8 l1}[a] (stage 2a)
//   with T_l[a] = w_N^{l a} Z_l[a].  The last sum (stage 2b) is evaluated only for window x.
//
// The per-lane phases are written once, against an executor: on the GPU `each` runs the
// body for this lane and `sync` is a wave-level LDS fence; in tests/emu `each` loops over
// 64 lanes on the host, so the index arithmetic below is unit-tested without a GPU.
#pragma once
#include "fmc_core.h"


namespace fmc {

// LDS exchange element: 8 bytes in both precisions (ds_write_b64 / ds_read_b64).
//   float : one complex64 per element, one pass;
//   double: one component per element, two passes (re, then im) through the same buffer.
template <class R> struct Xch;
template <> struct Xch<float> {
  using E = cpx<float>;
  static constexpr int NC = 1;
  static FMC_HD E pack(cpx<float> v, int) { return v; }
  static FMC_HD void unpack(cpx<float>& v, E e, int) { v = e; }
  static FMC_HD void acc(float& xr, float& xi, cpx<float> om, E f, int) {
    xr += om.x * f.x - om.y * f.y;
    xi += om.x * f.y + om.y * f.x;
  }
  static FMC_HD void first(float& xr, float& xi, E f, int) { xr += f.x; xi += f.y; }     // the m = 0 term: w^0 = 1
};
template <> struct Xch<double> {
  using E = double;
  static constexpr int NC = 2;
  static FMC_HD E pack(cpx<double> v, int c) { return c ? v.y : v.x; }
  static FMC_HD void unpack(cpx<double>& v, E e, int c) { if (c) v.y = e; else v.x = e; }
  static FMC_HD void acc(double& xr, double& xi, cpx<double> om, E f, int c) {
    if (c == 0) { xr += om.x * f; xi += om.y * f; }
    else        { xr -= om.y * f; xi += om.x * f; }
  }
  static FMC_HD void first(double& xr, double& xi, E f, int c) { if (c == 0) xr += f; else xi += f; }
};

// v or -v by a sign-bit xor (one integer instruction per value on the GPU).
FMC_HD float flip_sign(float v, bool neg) {
  uint32_t b;
  __builtin_memcpy(&b, &v, 4);
  b ^= (uint32_t)neg << 31;
  __builtin_memcpy(&v, &b, 4);
  return v;
}
FMC_HD double flip_sign(double v, bool neg) {
  uint64_t b;
  __builtin_memcpy(&b, &v, 8);
  b ^= (uint64_t)neg << 63;
  __builtin_memcpy(&v, &b, 8);
  return v;
}

template <class R, int P>
struct WaveGeom {
  static constexpr int N = WAVE * P;
  static constexpr int NB = (P + 7) / 8;           // radix-8 butterflies per lane in stage 2a (slots with a >= P idle)
  static constexpr int VN = NB * 8;                // register values per lane (stage 2a works on groups of 8)
  static constexpr int SE = 72;                    // row stride of exchange-1 image  E[a][l]: a*72 + l
  // exchange-2 image F[a][b0][l0] at  a + FL*l0 + FB*b0 : conflict-free for the 16-lane write groups
  // ((a mod 8) + 18 l0 covers 16 banks) and for the 32-lane read groups (a + 16*(b0 parity) covers 32)
  static constexpr int FL = P + 2;
  static constexpr int FB = 8 * FL;
  static constexpr int XELEMS = (P * SE > 8 * FB) ? P * SE : 8 * FB;   // 8-byte elements per wave
  static_assert(P >= 2 && P <= 32 && (P / (P & -P) == 1 || P / (P & -P) == 3 || P / (P & -P) == 5 || P / (P & -P) == 7 ||
                                      P / (P & -P) == 9),
                "wave FFT supports N = 64 P with P = 2^k times 1, 3, 5, 7 or 9, 2 <= P <= 32");
};

// Twiddles per prefetch chunk in stage 1 (0 = let the compiler place the loads): bounded by the
// registers left beside the 2P values of the row (f64: 4 VGPRs per twiddle).
template <class R, int P, int NS = 2>
constexpr int tw_chunk() {
  // four output slots leave no registers for the prefetch: 2048^2 / Np = 152 runs 116 / 138 / 153 k it/s with chunks of 4 / 2 / 0
  if (NS > 2 && sizeof(R) == 8) return 0;
  // measured at 1024^2 (P = 16): rows -1.6 % (f64), -5 % (f32); P <= 8 loses 2-6 % (more registers,
  // fewer waves per SIMD) and P > 24 has no registers to spare
  return (P < 12) ? 0 : (sizeof(R) == 8) ? (P <= 16 ? 4 : 0) : (P <= 16 ? P - 1 : (P <= 24 ? 8 : 0));
}

// Per-lane registers of the pipeline.  Wide enough for the 50-lane factorisation of fmc_mrfft.h too (P L0 / 64 radix-10
// butterflies per lane); elements a kernel never touches cost nothing (the array is scalarised).
constexpr int lane_regs_vn(int P) {
  const int w = (P + 7) / 8 * 8, m = (P * 5 + 63) / 64 * 10;
  return w > m ? w : m;
}
template <class R, int P, int NS>
struct LaneRegs {
  cpx<R> v[lane_regs_vn(P)];      // inputs (first P) -> stage values
  R xr[NS];      // outputs, real part       (slot s  <->  window index lane + 64 s)
  R xi[NS];      // outputs, imaginary part
  cpx<R> omc[NS][8];   // stage-2b table values of this lane's outputs kept in registers (OMC kernels only; untouched otherwise)
};

// ROW-0 INVARIANT: no row function of this file reads tw1[0 * 64 + .] or om[0 * omS + .] (w^0 = 1: the first term of every product /
// sum is taken as it is).  The GPU kernels rely on it -- they do not stage row 0 and address the tables through bases one row
// BELOW the LDS block (fmc_kernels.h: WaveLds) -- and emu_wavefft.cpp checks it on the host by poisoning row 0 with NaN.
// Tables (precomputed on the host in float64, stored as R):
//   tw1[a*64 + l]  = w_N^{l a}                                   (P*64 complex)
//   om[m*omS + oi] = w_64^{m * b(oi)},  m < 8                    (8*omS complex), b(oi) = (lo+oi) / P; row m = 0 (= 1) is
//                    never read: the first term of every sum is taken as it is
// The output-side fftshift sign of fmc_core.h (even N), (-1)^(lo+oi), is applied to the finished sums by a sign-bit xor
// (`osign` 0; 1: the opposite sign; -1: none -- the second transform of the chirp-z row).
// Which residues b0 = b mod 8 of the output blocks b = x / P the window [lo, lo + Np) touches: stage 2b reads only
// those planes of the exchange-2 image, so the others need not be stored (a window of 82 at P = 16 touches 6 of 8).
// General form: residues modulo L1 (8: wave pipeline; 10: 50-lane pipeline) of the output blocks b = x / P, x in [lo, lo + Np).
FMC_HD int window_planes(int lo, int Np, int P, int L1) {
  const int b_lo = lo / P, b_hi = (lo + Np - 1) / P;
  if (b_hi - b_lo >= L1 - 1) return (1 << L1) - 1;
  int m = 0;
  for (int b = b_lo; b <= b_hi; ++b) m |= 1 << (b % L1);
  return m;
}
// The planes a window of up to W pixels touches when it is centred on the start of a block of residue c (compile-time
// plane sets of the pruned kernels: centred pupil windows sit at N / 2, residue 0 for 64 P S and 50 P S with S even, 5 for
// 50 P S with S odd).
FMC_HD constexpr int centre_planes(int P, int L1, int c, int W = 96) {
  int m = 0;
  const int period = P * L1;
  for (int t = -W / 2; t < W - W / 2; ++t) {
    const int x = ((c * P + t) % period + period) % period;
    m |= 1 << (x / P);
  }
  return m;
}
// B0M: planes of the exchange-2 image kept (compile time: the others are neither stored nor computed).  A run-time mask was
// tried and bought nothing: no dead-code elimination, and the conditional stores lose their ds_write2 pairing.
// OMC: the stage-2b table values om[m][oi] of this lane's outputs come from r.omc (loaded once by the caller and kept over
// all the rows of a wave) instead of the LDS: on the small grids (P <= 8) those 28 ds_read_b128 per row were a third of the
// LDS time of a row that is LDS-bound (256^2: 91 LDS instructions per 256-point row against 83 per 1024-point row).
template <class R, int P, int NS, int B0M = 0xFF, bool OMC = false, class Exec>
FMC_HD void pruned_row_fft(Exec& ex, typename Xch<R>::E* xbuf, const cpx<R>* tw1, const cpx<R>* om,
                           int omS, int lo, int Np, int osign = 0) {
  using G = WaveGeom<R, P>;
  using X = Xch<R>;
  using E = typename X::E;
  constexpr int NC = X::NC;
  const int nslots = (Np + WAVE - 1) / WAVE;

  // ---- stage 1: radix-P in registers, twiddle, to natural order
  ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
    cpx<R> z[P];
    // Twiddle loads in chunks of CH, double-buffered: chunk 0 is in flight under the in-register
    // DFT, chunk k+1 under the multiplies of chunk k, one wait (ex.pin) per chunk.  Left to itself
    // the compiler keeps two loads in flight and exposes ~P/2 LDS round trips per row.
    constexpr int CH = tw_chunk<R, P, NS>();
    constexpr int NCH = CH ? (P - 1 + CH - 1) / CH : 0;
    cpx<R> t[2][CH ? CH : 1];
    if (CH) {
#pragma unroll
      for (int q = 0; q < CH; ++q)
        if (1 + q < P) t[0][q] = tw1[(1 + q) * WAVE + lane];
    }
#pragma unroll
    for (int j = 0; j < P; ++j) z[j] = r.v[j];
    dft_reg<P, R>(z);
    r.v[0] = z[0];
    if (CH) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (k + 1 < NCH) {
#pragma unroll
          for (int q = 0; q < CH; ++q)
            if (1 + (k + 1) * CH + q < P) t[(k + 1) & 1][q] = tw1[(1 + (k + 1) * CH + q) * WAVE + lane];
        }
#pragma unroll
        for (int q = 0; q < CH; ++q)
          if (1 + k * CH + q < P) { ex.pin(t[k & 1][q].x); ex.pin(t[k & 1][q].y); }
#pragma unroll
        for (int q = 0; q < CH; ++q)
          if (1 + k * CH + q < P) r.v[1 + k * CH + q] = cmul(z[1 + k * CH + q], t[k & 1][q]);
      }
    } else {
#pragma unroll
      for (int a = 1; a < P; ++a) r.v[a] = cmul(z[a], tw1[a * WAVE + lane]);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) { r.xr[s] = (R)0; r.xi[s] = (R)0; }
  });

  // ---- exchange 1 (+ stage 2a after the last component)
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
#pragma unroll
      for (int a = 0; a < P; ++a) ex.st(xbuf + a * G::SE + lane, X::pack(r.v[a], c));
    });
    ex.sync();
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
      const int l0 = lane & 7, i = lane >> 3;
#pragma unroll
      for (int jj = 0; jj < G::NB; ++jj)
        if ((P % 8 == 0) || i + 8 * jj < P) {
#pragma unroll
          for (int m = 0; m < 8; ++m)
            X::unpack(r.v[jj * 8 + m], ex.ld(xbuf + (i + 8 * jj) * G::SE + l0 + 8 * m), c);
        }
    });
    ex.sync();
  }
  ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
#pragma unroll
    for (int jj = 0; jj < G::NB; ++jj) {
      cpx<R> t[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) t[m] = r.v[jj * 8 + m];
      fft_dif<8, R>(t);
#pragma unroll
      for (int b0 = 0; b0 < 8; ++b0) r.v[jj * 8 + b0] = t[brev(b0, 3)];   // natural b0 order
    }
  });

  // ---- exchange 2 + stage 2b (pruned): 8-term sums for the window outputs
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
      const int l0 = lane & 7, i = lane >> 3;
#pragma unroll
      for (int jj = 0; jj < G::NB; ++jj)
        if ((P % 8 == 0) || i + 8 * jj < P) {
#pragma unroll
          for (int b0 = 0; b0 < 8; ++b0)
            if ((B0M >> b0) & 1)   // planes no window output reads are not stored
              ex.st(xbuf + (i + 8 * jj) + G::FL * l0 + G::FB * b0, X::pack(r.v[jj * 8 + b0], c));
        }
    });
    ex.sync();
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s < nslots) {
          const int oi = lane + WAVE * s;
          if (oi < Np) {
            const int x = lo + oi;
            const int a = x % P;            // P is a compile-time constant: mask / shift for powers of two
            const int b0 = (x / P) & 7;
            const E* f = xbuf + a + G::FB * b0;
            if (P >= 12 && NS == 2) {      // the eight loads of a sum issued together, one wait
            cpx<R> w[8];
            E fv[8];
            fv[0] = ex.ld(f);
#pragma unroll
            for (int m = 1; m < 8; ++m) { w[m] = om[m * omS + oi]; fv[m] = ex.ld(f + G::FL * m); }
            ex.pin(fv[0]);
#pragma unroll
            for (int m = 1; m < 8; ++m) { ex.pin(w[m].x); ex.pin(w[m].y); ex.pin(fv[m]); }
            X::first(r.xr[s], r.xi[s], fv[0], c);
#pragma unroll
            for (int m = 1; m < 8; ++m) X::acc(r.xr[s], r.xi[s], w[m], fv[m], c);
            } else {
            X::first(r.xr[s], r.xi[s], ex.ld(f), c);
#pragma unroll
            for (int m = 1; m < 8; ++m) X::acc(r.xr[s], r.xi[s], OMC ? r.omc[s][m] : om[m * omS + oi], ex.ld(f + G::FL * m), c);
            }
            if (c == NC - 1 && osign >= 0) {
              const bool neg = ((x ^ osign) & 1) != 0;
              r.xr[s] = flip_sign(r.xr[s], neg);
              r.xi[s] = flip_sign(r.xi[s], neg);
            }
          }
        }
      }
    });
    ex.sync();
  }
}

// ---------------------------------------------------------------- P = 16: dense LDS images
// Exchange buffer of the P = 16 rows below: E[a][l] at 66 a + l (16 x 66 eight-byte elements = 8448 B per wave), small enough
// for SIXTEEN waves per workgroup beside the tables (16 x 8448 B + 28 KB = the 160 KB of a CU): four waves per SIMD.
constexpr int D16_SE = 66;
constexpr int D16_XELEMS = 16 * D16_SE;
// P of the wave pipeline whose centred 96-pixel window leaves planes unread (P <= 14: all eight are touched)
FMC_HD constexpr bool prune_pays(int P, int L1, int c) { return centre_planes(P, L1, c) != (1 << L1) - 1; }

// ---------------------------------------------------------------- P = 16, dense images, 64 = 16 x 4 on the lanes
// The same row with the lane dimension factored 64 = L1 x L0 = 16 x 4 instead of 8 x 8:
//     in-register radix-16  ->  exchange 1  ->  in-register radix-16 (ONE butterfly per lane: lane (a = lane & 15, l0 = lane >> 4)
//     owns T_{l0 + 4 l1}[a], l1 < 16)  ->  exchange 2  ->  4-term sums for the window outputs
//   X[x] = sum_{l0 < 4} w_64^{l0 b} U[a][l0][b0],   U[a][l0][b0] = sum_{l1 < 16} w_16^{l1 b0} T_{l0 + 4 l1}[a],   b0 = b mod 16.
// The pruned stage has 4 terms instead of 8, so it reads half the exchange-2 elements and 3 instead of 7 table values per
// output, and the exchange-2 image has SIXTEEN planes of which a centred window of up to 96 pixels touches six
// ({13, 14, 15, 0, 1, 2}): ten are neither stored nor computed (the radix-16 network loses the operations that only fed
// them).  Per row and lane: the same float64 operation count as the 8 x 8 form with six of eight planes (stage 2a ~136
// instead of 104, stage 2b 28 instead of 60), but 73 KB instead of 103 KB through the LDS -- and the LDS share of the issue
// time goes with the bytes.  Images: exchange 1 as before (66 a + l; the owners read 66 a + l0 + 4 l1: 2 a + l0 distinct in
// a half-wave); exchange 2 F[a][b0][l0] at a + 16 b0 + 256 l0 (writes: 16 consecutive a; reads: x mod 256 consecutive).
// Uses rows m = 1, 2, 3 of the `om` table (w_64^{m b}).
FMC_HD constexpr int popcount16(int m) { int n = 0; for (int i = 0; i < 16; ++i) n += (m >> i) & 1; return n; }
// NP planes around b0 = 0: {16 - NP / 2, ..., 15, 0, ..., NP - NP / 2 - 1}
FMC_HD constexpr int centre_run_mask(int NP) {
  int m = 0;
  for (int p = 0; p < NP; ++p) m |= 1 << ((p - NP / 2 + 16) & 15);
  return m;
}
constexpr int D16R_CENTRE_MASK = centre_planes(16, 16, 0);
static_assert(D16R_CENTRE_MASK == 0xE007, "planes {13, 14, 15, 0, 1, 2}");
constexpr int D16R_WIDE_MASK = centre_planes(16, 16, 0, 128);      // centred windows of up to 128 pixels
static_assert(D16R_WIDE_MASK == 0xF00F, "planes {12, ..., 15, 0, ..., 3}");
static_assert(centre_run_mask(6) == D16R_CENTRE_MASK && centre_run_mask(8) == D16R_WIDE_MASK, "contiguous plane sets");
template <class R, int NS, int B0M = 0xFFFF, class Exec>
FMC_HD void pruned_row_fft_d16r(Exec& ex, typename Xch<R>::E* xbuf, const cpx<R>* tw1, const cpx<R>* om,
                                int omS, int lo, int Np) {
  constexpr int P = 16;
  using X = Xch<R>;
  using E = typename X::E;
  constexpr int NC = X::NC;
  const int nslots = (Np + WAVE - 1) / WAVE;
  ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
    cpx<R> z[P];
#pragma unroll
    for (int j = 0; j < P; ++j) z[j] = r.v[j];
    dft_reg<P, R>(z);
    r.v[0] = z[0];
#pragma unroll
    for (int a = 1; a < P; ++a) r.v[a] = cmul(z[a], tw1[a * WAVE + lane]);
#pragma unroll
    for (int s = 0; s < NS; ++s) { r.xr[s] = (R)0; r.xi[s] = (R)0; }
  });
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
#pragma unroll
      for (int a = 0; a < P; ++a) ex.st(xbuf + a * D16_SE + lane, X::pack(r.v[a], c));
    });
    ex.sync();
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
      const int a = lane & 15, l0 = lane >> 4;
#pragma unroll
      for (int l1 = 0; l1 < 16; ++l1) X::unpack(r.v[l1], ex.ld(xbuf + a * D16_SE + l0 + 4 * l1), c);
    });
    ex.sync();
  }
  ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
    cpx<R> t[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) t[m] = r.v[m];
    fft_dif<16, R>(t);
#pragma unroll
    for (int b0 = 0; b0 < 16; ++b0) r.v[b0] = t[brev(b0, 4)];
  });
  // (Round 3 tried the two conflict-free forms of this exchange: aligned ds_write_b128 stores, rows +7 %; real and imaginary
  // parts in two 8-byte images 512-byte multiples apart, rows +4.3 %, columns -1.5 %.  The ds_write2_b64 pairs with their
  // 2-way store conflicts are the fastest form: profiles/r03_ab_fma_vs_aligned_stores.txt, r03_ab_split_exchange2_images.txt.)
  // float64 with a centred plane set (<= 8 planes, contiguous around b0 = 0): exchange 2 in ONE pass with 16-byte elements --
  // plane b0 at position p = (b0 + NP / 2) & 15 < NP, element (a, p, l0) at a + 16 p + 16 NP l0 (<= 512 elements of 16 bytes):
  // the table values are read once instead of once per component, two hand-offs instead of four.
  constexpr int NP = popcount16(B0M);
  if constexpr (sizeof(R) == 8 && NP <= 8 && B0M == centre_run_mask(NP)) {
    cpx<R>* cbuf = reinterpret_cast<cpx<R>*>(xbuf);
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
      const int a = lane & 15, l0 = lane >> 4;
#pragma unroll
      for (int b0 = 0; b0 < 16; ++b0)
        if ((B0M >> b0) & 1) ex.st(cbuf + a + 16 * ((b0 + NP / 2) & 15) + 16 * NP * l0, r.v[b0]);
    });
    ex.sync();
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s < nslots) {
          const int oi = lane + WAVE * s;
          if (oi < Np) {
            const int x = lo + oi;
            const cpx<R>* f = cbuf + ((x + 8 * NP) & 255);        // a + 16 p: consecutive outputs, consecutive elements
            cpx<R> acc = ex.ld(f);
#pragma unroll
            for (int m = 1; m < 4; ++m) acc = cfma(om[m * omS + oi], ex.ld(f + 16 * NP * m), acc);
            const bool neg = (x & 1) != 0;
            r.xr[s] = flip_sign(acc.x, neg);
            r.xi[s] = flip_sign(acc.y, neg);
          }
        }
      }
    });
    ex.sync();
    return;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
      const int a = lane & 15, l0 = lane >> 4;
#pragma unroll
      for (int b0 = 0; b0 < 16; ++b0)
        if ((B0M >> b0) & 1) ex.st(xbuf + a + 16 * b0 + 256 * l0, X::pack(r.v[b0], c));
    });
    ex.sync();
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s < nslots) {
          const int oi = lane + WAVE * s;
          if (oi < Np) {
            const int x = lo + oi;
            const E* f = xbuf + (x & 255);
            X::first(r.xr[s], r.xi[s], ex.ld(f), c);
#pragma unroll
            for (int m = 1; m < 4; ++m) X::acc(r.xr[s], r.xi[s], om[m * omS + oi], ex.ld(f + 256 * m), c);
            if (c == NC - 1) {
              const bool neg = (x & 1) != 0;
              r.xr[s] = flip_sign(r.xr[s], neg);
              r.xi[s] = flip_sign(r.xi[s], neg);
            }
          }
        }
      }
    });
    ex.sync();
  }
}

// ---------------------------------------------------------------- N = 128, 256, 512: 8 / 4 / 2 rows per wavefront (packed rows)
// The 16 x 16 x L0 factorisation above with L0 = 1 (N = 256) or 2 (N = 512) leaves lanes idle if one wave transforms one
// row, and the P x 8 x 8 row at P = 4 / 8 is LDS-bound (a 256-point row costs 56 % of a 1024-point row).  Here a wavefront
// transforms G = 4 / L0 rows AT ONCE: lane = g L + q (row g of the wave's G, q < L = 16 L0) holds c[q + L j], j < 16, of
// its row, and the pipeline is the one of pruned_row_fft_d16r inside every group of L lanes:
//   Z_q[a] = sum_j c[q + L j] w_16^{ja},  T_q[a] = w_N^{qa} Z_q[a]               (stage 1, radix 16 in registers)
//   exchange 1 inside the group: lane (a = q & 15, l0 = q >> 4) collects T_{l0 + L0 l1}[a], l1 < 16
//   U[a][l0][b0] = sum_l1 w_16^{l1 b0} T_{l0 + L0 l1}[a]                         (stage 2, ONE radix-16 butterfly per lane)
//   X[a + 16 b] = sum_{l0 < L0} w_L^{l0 b} U[a][l0][b mod 16]                     (L0 = 1: nothing left to do)
// L0 = 1: the lane's outputs are its planes, x = a + 16 b0, left in r.v[b0] (natural order, output-side fftshift sign
// applied); L0 = 2: exchange 2 and two-term sums into output slots, oi = q + 32 s (r.xr / r.xi).  Exchange-1 image as in
// the 1024-point row (SE a + lane) with SE = 65 for L0 = 1 (66 would put rows g and g + 2 on the same banks under the
// lane-group rules tools/lds_bank_check.py models) and 66 for L0 = 2: conflict-free writes and reads.
// B0M: planes b0 kept.  A centred window of up to 96 pixels touches six: {5, ..., 10} at N = 256 (centre block 8), {13, 14,
// 15, 0, 1, 2} at N = 512 (centre block 16 = 0 mod 16).
// N = 128 (L0 = 0 below, L = 8 lanes per row, EIGHT rows per wavefront): the lane dimension holds 8 = 16 / 2 points, so lane
// (g, i < 8) owns TWO radix-8 butterflies of stage 2, a = i and a = i + 8: X[a + 16 b] = sum_{q < 8} w_8^{q b} T_q[a], b < 8,
// left in r.v[8 m + b] (a = i + 8 m).  B0M is then a mask over the eight b; a centred window touches {1, ..., 6}.
template <int L0> constexpr int PK_SE = (L0 == 2) ? 66 : 65;
constexpr int pk_lanes(int L0) { return L0 == 0 ? 8 : 16 * L0; }
template <int L0> constexpr int pk_centre_mask() { return L0 == 0 ? centre_planes(16, 8, 4) : centre_planes(16, 16, (8 * L0) & 15); }
template <int L0> constexpr int pk_all_mask() { return L0 == 0 ? 0xFF : 0xFFFF; }
static_assert(pk_centre_mask<1>() == 0x07E0 && pk_centre_mask<2>() == D16R_CENTRE_MASK && pk_centre_mask<0>() == 0x7E,
              "planes {5 ... 10} / {13 ... 2} / {1 ... 6}");
template <class R, int L0, int NSL, int B0M = 0xFFFF, class Exec>
FMC_HD void packed_row_fft(Exec& ex, typename Xch<R>::E* xbuf, const cpx<R>* tw1, const cpx<R>* om, int omS, int lo, int Np) {
  static_assert(L0 == 0 || L0 == 1 || L0 == 2, "N = 128, 256 or 512");
  constexpr int P = 16, L = pk_lanes(L0), SE = PK_SE<L0>;
  using X = Xch<R>;
  using E = typename X::E;
  constexpr int NC = X::NC;
  ex.each([&](int lane, LaneRegs<R, P, NSL>& r) {
    cpx<R> z[P];
#pragma unroll
    for (int j = 0; j < P; ++j) z[j] = r.v[j];
    dft_reg<P, R>(z);
    r.v[0] = z[0];
#pragma unroll
    for (int a = 1; a < P; ++a) r.v[a] = cmul(z[a], tw1[a * L + (lane & (L - 1))]);      // the G groups read the same entries
  });
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ex.each([&](int lane, LaneRegs<R, P, NSL>& r) {
#pragma unroll
      for (int a = 0; a < P; ++a) ex.st(xbuf + a * SE + lane, X::pack(r.v[a], c));
    });
    ex.sync();
    ex.each([&](int lane, LaneRegs<R, P, NSL>& r) {
      const int q = lane & (L - 1);
      if constexpr (L0 == 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const E* e = xbuf + (q + 8 * m) * SE + (lane - q);
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) X::unpack(r.v[8 * m + qq], ex.ld(e + qq), c);
        }
      } else {
        const int a = q & 15, l0 = q >> 4;
        const E* e = xbuf + a * SE + (lane - q) + l0;
#pragma unroll
        for (int l1 = 0; l1 < 16; ++l1) X::unpack(r.v[l1], ex.ld(e + L0 * l1), c);
      }
    });
    ex.sync();
  }
  ex.each([&](int lane, LaneRegs<R, P, NSL>& r) {
    const bool neg = (lane & 1) != 0;            // L0 < 2: x = a + 16 b has the parity of the lane
    if constexpr (L0 == 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        cpx<R> t[8];
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) t[qq] = r.v[8 * m + qq];
        fft_dif<8, R>(t);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          r.v[8 * m + b] = t[brev(b, 3)];
          if ((B0M >> b) & 1) { r.v[8 * m + b].x = flip_sign(r.v[8 * m + b].x, neg); r.v[8 * m + b].y = flip_sign(r.v[8 * m + b].y, neg); }
        }
      }
    } else {
      cpx<R> t[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) t[m] = r.v[m];
      fft_dif<16, R>(t);
#pragma unroll
      for (int b0 = 0; b0 < 16; ++b0) {
        r.v[b0] = t[brev(b0, 4)];
        if (L0 == 1 && ((B0M >> b0) & 1)) { r.v[b0].x = flip_sign(r.v[b0].x, neg); r.v[b0].y = flip_sign(r.v[b0].y, neg); }
      }
    }
  });
  if constexpr (L0 == 2) {
    constexpr int NP = popcount16(B0M);
    if constexpr (sizeof(R) == 8 && NP <= 8 && B0M == centre_run_mask(NP)) {
      // one pass with 16-byte elements: plane b0 at position p = (b0 + NP / 2) & 15 < NP, element (g, a, p, l0) at
      // 32 NP g + a + 16 p + 16 NP l0
      cpx<R>* cbuf = reinterpret_cast<cpx<R>*>(xbuf);
      ex.each([&](int lane, LaneRegs<R, P, NSL>& r) {
        const int q = lane & (L - 1), a = q & 15, l0 = q >> 4, g = lane >> 5;
#pragma unroll
        for (int b0 = 0; b0 < 16; ++b0)
          if ((B0M >> b0) & 1) ex.st(cbuf + 32 * NP * g + a + 16 * ((b0 + NP / 2) & 15) + 16 * NP * l0, r.v[b0]);
      });
      ex.sync();
      ex.each([&](int lane, LaneRegs<R, P, NSL>& r) {
        const int q = lane & (L - 1), g = lane >> 5;
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
          const int oi = q + L * s;
          if (oi < Np) {
            const int x = lo + oi;
            const cpx<R>* f = cbuf + 32 * NP * g + ((x + 8 * NP) & 255);
            const cpx<R> acc = cfma(om[omS + oi], ex.ld(f + 16 * NP), ex.ld(f));
            const bool neg = (x & 1) != 0;
            r.xr[s] = flip_sign(acc.x, neg);
            r.xi[s] = flip_sign(acc.y, neg);
          }
        }
      });
      ex.sync();
    } else {
      ex.each([&](int, LaneRegs<R, P, NSL>& r) {
#pragma unroll
        for (int s = 0; s < NSL; ++s) { r.xr[s] = (R)0; r.xi[s] = (R)0; }
      });
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        ex.each([&](int lane, LaneRegs<R, P, NSL>& r) {
          const int q = lane & (L - 1), a = q & 15, l0 = q >> 4, g = lane >> 5;
#pragma unroll
          for (int b0 = 0; b0 < 16; ++b0)
            if ((B0M >> b0) & 1) ex.st(xbuf + 512 * g + a + 16 * b0 + 256 * l0, X::pack(r.v[b0], c));
        });
        ex.sync();
        ex.each([&](int lane, LaneRegs<R, P, NSL>& r) {
          const int q = lane & (L - 1), g = lane >> 5;
#pragma unroll
          for (int s = 0; s < NSL; ++s) {
            const int oi = q + L * s;
            if (oi < Np) {
              const int x = lo + oi;
              const E* f = xbuf + 512 * g + (x & 255);
              X::first(r.xr[s], r.xi[s], ex.ld(f), c);
              X::acc(r.xr[s], r.xi[s], om[omS + oi], ex.ld(f + 256), c);
              if (c == NC - 1) {
                const bool neg = (x & 1) != 0;
                r.xr[s] = flip_sign(r.xr[s], neg);
                r.xi[s] = flip_sign(r.xi[s], neg);
              }
            }
          }
        });
        ex.sync();
      }
    }
  }
}
// f(oi, re, im) for every window output this lane holds after packed_row_fft (oi ascending).
template <class R, int L0, int NSL, int B0M, class F>
FMC_HD void packed_outputs(int lane, const LaneRegs<R, 16, NSL>& r, int lo, int Np, F f) {
  if constexpr (L0 == 0) {
    const int i = lane & 7;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int m = 0; m < 2; ++m)
        if ((B0M >> b) & 1) {
          const int oi = i + 8 * m + 16 * b - lo;
          if (oi >= 0 && oi < Np) f(oi, r.v[8 * m + b].x, r.v[8 * m + b].y);
        }
  } else if constexpr (L0 == 1) {
    const int a = lane & 15;
#pragma unroll
    for (int b0 = 0; b0 < 16; ++b0)
      if ((B0M >> b0) & 1) {
        const int oi = a + 16 * b0 - lo;
        if (oi >= 0 && oi < Np) f(oi, r.v[b0].x, r.v[b0].y);
      }
  } else {
    const int q = lane & 31;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int oi = q + 32 * s;
      if (oi < Np) f(oi, r.xr[s], r.xi[s]);
    }
  }
}
// ---------------------------------------------------------------- N = S x 256 / S x 128: packed SUB-ROWS (round 6)
// A row of N = S M points (M = 16 L: the packed 256-point pipeline, L0 = 1, or the 128-point one, L0 = 0) as S interleaved
// sub-rows c_s[m] = c[s + S m]:
//     X[x] = sum_{s < S} w_N^{s x} Y_s[x mod M],      Y_s = DFT_M(c_s)            (decimation in time)
// evaluated for the window only.  A wavefront transforms the G = 4 / 8 rows of a unit one sub-row index s at a time with
// packed_row_fft -- sixteen values per lane whatever N, where the one-row-per-wave form holds N / 64 = 10 ... 28.  The window's
// centre N / 2 falls on M / 2 of every sub-transform when S is odd and on 0 when S is even, so six planes of the
// sub-transform ARE the outputs of a centred window of up to 96 pixels -- x = N / 2 - 48 + e, e < 96, whatever S:
//   L0 = 1: lane (g, a) holds e = a + 16 p in plane b0 = (FIRST + p) mod 16, FIRST = 5 (S odd) or 13 (S even);
//   L0 = 0: lane (g, i) holds e = i + 8 m + 16 p, m < 2, in r.v[8 m + FIRST + p], FIRST = 1 (S odd)
// and the sum over s runs in the lane's own registers: no exchange between sub-rows, no second image.  The accumulators live
// in r.omc[m][p] (the one-row kernels' register table is unused here).  Twiddles: pcw[s * 96 + e] = w_N^{s x} -- 96 entries per
// sub-row that depend on N only (build_pcw); the G groups of a wavefront read the same sixteen.
constexpr int PKS_SPAN = 96;                                         // outputs of a sub-transform the six planes hold
// NPL = 8 (late round 6): EIGHT planes -- x = N / 2 - 64 + e, e < 128: centred windows of 97 ... 128 pixels, which fell back to staged
// draws on one-row-per-wave rows at a third to a quarter of the rate.  One plane more on either side: FIRST - 1, eight (sixteen)
// accumulators per lane, 128 table entries per sub-row.  Run-time sub-row counts only (any S).
constexpr int pks_span(int NPL) { return 16 * NPL; }
// a values per lane (L0 = -1: the 64-point form, same accumulator layout as L0 = 0); NPL = 16 (ALL planes of a 256-point sub-transform:
// centred windows of up to 256 pixels, L0 = 1 only) keeps its sixteen accumulators in both halves of r.omc: plane p in omc[p >> 3][p & 7]
template <int L0, int NPL = 6> constexpr int pks_nm() { return (L0 <= 0 || NPL > 8) ? 2 : 1; }
constexpr int pks_first_plane(int L0, int S, int NPL = 6) { return (L0 == 0 ? ((S & 1) ? 1 : 5) : ((S & 1) ? 5 : 13)) - (NPL - 6) / 2; }
constexpr int pks_plane_mask(int L0, int S, int NPL = 6) {
  int m = 0;
  for (int p = 0; p < NPL; ++p) m |= 1 << ((pks_first_plane(L0, S, NPL) + p) & (L0 == 0 ? 7 : 15));
  return m;
}
static_assert(pks_plane_mask(1, 3) == pk_centre_mask<1>() && pks_plane_mask(0, 5) == pk_centre_mask<0>() && pks_plane_mask(1, 6) == D16R_CENTRE_MASK,
              "the centred planes of the 256 / 128-point grids, and the planes around 0");
template <class R, int L0, int NPL = 6, class Exec>
FMC_HD void pks_clear(Exec& ex) {
  static_assert(NPL <= 8 || L0 == 1, "all sixteen planes: 256-point sub-rows only");
  ex.each([&](int, LaneRegs<R, 16, pks_nm<L0, NPL>()>& r) {
#pragma unroll
    for (int m = 0; m < pks_nm<L0, NPL>(); ++m)
#pragma unroll
      for (int p = 0; p < (NPL > 8 ? 8 : NPL); ++p) r.omc[m][p] = mk<R>((R)0, (R)0);
  });
}
// after packed_row_fft of sub-row s (its planes in r.v, output-side sign applied): acc[m][p] += w_N^{s x} Y_s[x mod M]
template <class R, int L0, int FIRST, int NPL = 6, class Exec>
FMC_HD void pks_accumulate(Exec& ex, const cpx<R>* pcw_s) {
  ex.each([&](int lane, LaneRegs<R, 16, pks_nm<L0, NPL>()>& r) {
    if constexpr (L0 == 1) {
      const cpx<R>* w = pcw_s + (lane & 15);
#pragma unroll
      for (int p = 0; p < NPL; ++p) r.omc[p >> 3][p & 7] = cfma(ex.ld(w + 16 * p), r.v[(FIRST + p) & 15], r.omc[p >> 3][p & 7]);
    } else {
      const cpx<R>* w = pcw_s + (lane & 7);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < NPL; ++p) r.omc[m][p] = cfma(ex.ld(w + 8 * m + 16 * p), r.v[8 * m + ((FIRST + p) & 7)], r.omc[m][p]);
    }
  });
}
// f(oi, re, im) for every window output this lane holds after the last pks_accumulate; N the full row length
template <class R, int L0, int NPL = 6, class F>
FMC_HD void pks_outputs(int lane, const LaneRegs<R, 16, pks_nm<L0, NPL>()>& r, int N, int lo, int Np, F f) {
  const int a = lane & (L0 <= 0 ? 7 : 15);
  if constexpr (L0 == 1) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      const int oi = N / 2 - 8 * NPL + a + 16 * p - lo;
      if (oi >= 0 && oi < Np) f(oi, r.omc[p >> 3][p & 7].x, r.omc[p >> 3][p & 7].y);
    }
  } else {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int oi = N / 2 - 8 * NPL + a + 8 * m + 16 * p - lo;
        if (oi >= 0 && oi < Np) f(oi, r.omc[m][p].x, r.omc[m][p].y);
      }
  }
}
template <class R, class CosSin>
inline void build_pcw(cpx<R>* pcw, int N, int S, CosSin cs, int NPL = 6) {
  const int span = pks_span(NPL);
  for (int sp = 0; sp < S; ++sp)
    for (int e = 0; e < span; ++e) {
      const long long x = N / 2 - span / 2 + e;
      double c, sn;
      cs((double)((sp * x) % N) / N, &c, &sn);
      pcw[sp * span + e] = mk<R>((R)c, (R)(-sn));
    }
}

// ---------------------------------------------------------------- N = S x 64 (192, 320, 448, 576): packed sub-rows of SIXTY-FOUR points
// The sub-transform is shorter than the window: every one of its 64 outputs is needed, and those whose residue falls twice into the 96
// window positions x = N / 2 - 48 + e (S odd: N / 2 - 48 = 48 mod 64, so residues 48 ... 63 and 0 ... 15) are needed for BOTH, with
// different twiddles.  A wavefront transforms EIGHT rows at a time, EIGHT lanes per sub-row and eight values per lane (64 = 8 x 8):
// lane (g, q) holds c_s[q + 8 j], j < 8,
//   Z_q[a] = sum_j c_s[q + 8 j] w_8^{ja},   T_q[a] = w_64^{qa} Z_q[a]              (radix 8 in registers, twiddle)
//   exchange inside the eight lanes: lane a collects T_{q'}[a], q' < 8
//   Y_s[a + 8 b] = sum_{q'} w_8^{q' b} T_{q'}[a], b < 8                            (ONE radix-8 butterfly per lane: all 64 outputs)
//   acc[e] += pcw[s][e] Y_s[a + 8 b(p')],  e = a + 8 p', p' < 12,  b(p') = (p' + 6) mod 8
// -- the SAME 96-entry table as the longer sub-rows, and the accumulator layout of the 128-point form (lane i < 8 holds e = i + 8 m +
// 16 p in r.omc[m][p], m < 2, p < 6: p' = m + 2 p), so pks_clear / pks_outputs and the column kernel's epilogue serve L0 = -1 as they
// serve L0 = 0.  Twelve accumulators and eight values per lane: three to four waves per SIMD where a sixteen-values form with four
// lanes per sub-row (24 accumulators, 255 registers) ran two (profiles/r06_ab_packed_subrows.txt section 6).  Exchange image E[a][lane]
// at 65 a + lane.  The generator draws N / 8 streams of EIGHT advances on these grids (fmc_core.h: stream_lanes).
constexpr int PKS64_SE = 65;
// one sub-row pass: r.v[j] = c_s[q + 8 j], j < 8 (input-side sign folded in) -> the accumulators; tw64[a * 8 + q] = w_64^{q a}
template <class R, int NPL = 6, class Exec>
FMC_HD void pks64_pass(Exec& ex, typename Xch<R>::E* xbuf, const cpx<R>* tw64, const cpx<R>* pcw_s) {
  using X = Xch<R>;
  using E = typename X::E;
  constexpr int NC = X::NC;
  ex.each([&](int lane, LaneRegs<R, 16, 2>& r) {
    cpx<R> z[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = r.v[j];
    fft_dif<8, R>(z);
    r.v[0] = z[0];
#pragma unroll
    for (int a = 1; a < 8; ++a) r.v[a] = cmul(z[brev(a, 3)], tw64[a * 8 + (lane & 7)]);
  });
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ex.each([&](int lane, LaneRegs<R, 16, 2>& r) {
#pragma unroll
      for (int a = 0; a < 8; ++a) ex.st(xbuf + a * PKS64_SE + lane, X::pack(r.v[a], c));
    });
    ex.sync();
    ex.each([&](int lane, LaneRegs<R, 16, 2>& r) {
      const int q = lane & 7;
      const E* e = xbuf + q * PKS64_SE + (lane - q);
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) X::unpack(r.v[qq], ex.ld(e + qq), c);
    });
    ex.sync();
  }
  ex.each([&](int lane, LaneRegs<R, 16, 2>& r) {
    const bool neg = (lane & 1) != 0;            // x = a + 8 b (+ 64 k) has the parity of a: the output-side fftshift sign
    const cpx<R>* w = pcw_s + (lane & 7);
    cpx<R> t[8];
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) t[qq] = r.v[qq];
    fft_dif<8, R>(t);
    cpx<R> y[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) { y[b] = t[brev(b, 3)]; y[b].x = flip_sign(y[b].x, neg); y[b].y = flip_sign(y[b].y, neg); }
#pragma unroll
    // (e = a + 8 p', x = N / 2 - 8 NPL + e: N / 2 = 32 mod 64 for odd S, so b(p') = (p' + 12 - NPL) mod 8 -- 6 for six planes, 4 for eight)
    for (int pp = 0; pp < 2 * NPL; ++pp) r.omc[pp & 1][pp >> 1] = cfma(ex.ld(w + 8 * pp), y[(pp + 12 - NPL) & 7], r.omc[pp & 1][pp >> 1]);
  });
}
template <class R, class CosSin>
inline void build_tw64(cpx<R>* tw, CosSin cs) {
  for (int a = 0; a < 8; ++a)
    for (int q = 0; q < 8; ++q) {
      double c, sn;
      cs((double)((q * a) % 64) / 64.0, &c, &sn);
      tw[a * 8 + q] = mk<R>((R)c, (R)(-sn));
    }
}

// tw1[a * L + q] = w_N^{q a}, N = 16 L (16 L entries);  om[1 * omS + oi] = w_L^{b}, b = ((lo + oi) / 16) mod L  (L0 = 2
// only; row 0 is never read)
template <class R, class CosSin>
inline void build_tw1_pk(cpx<R>* tw1, int L, CosSin cs) {
  const int N = 16 * L;
  for (int a = 0; a < 16; ++a)
    for (int l = 0; l < L; ++l) {
      double c, s;
      cs((double)((l * a) % N) / N, &c, &s);
      tw1[a * L + l] = mk<R>((R)c, (R)(-s));
    }
}
template <class R, class CosSin>
inline void build_om_pk(cpx<R>* om, int omS, int L, int lo, int Np, CosSin cs) {
  for (int m = 0; m < 2; ++m)
    for (int oi = 0; oi < omS; ++oi) {
      if (oi >= Np) { om[m * omS + oi] = mk<R>((R)0, (R)0); continue; }
      const int b = ((lo + oi) / 16) % L;
      double c, s;
      cs((double)((m * b) % L) / L, &c, &s);
      om[m * omS + oi] = mk<R>((R)c, (R)(-s));
    }
}

// Host-side construction of the two tables (float64 trigonometry by the caller-supplied functor
// `cs(turns, &c, &s)` = cos/sin(2*pi*turns)).
template <class R, class CosSin>
inline void build_tw1(cpx<R>* tw1, int P, CosSin cs) {
  const int N = WAVE * P;
  for (int a = 0; a < P; ++a)
    for (int l = 0; l < WAVE; ++l) {
      double c, s;
      cs((double)((l * a) % N) / N, &c, &s);
      tw1[a * WAVE + l] = mk<R>((R)c, (R)(-s));
    }
}
template <class R, class CosSin>
inline void build_om(cpx<R>* om, int omS, int P, int lo, int Np, CosSin cs) {
  for (int m = 0; m < 8; ++m)
    for (int oi = 0; oi < omS; ++oi) {
      if (oi >= Np) { om[m * omS + oi] = mk<R>((R)0, (R)0); continue; }
      const int x = lo + oi;
      const int b = (x / P) & 63;
      double c, s;
      cs((double)((m * b) % 64) / 64.0, &c, &s);
      om[m * omS + oi] = mk<R>((R)c, (R)(-s));
    }
}

}  // namespace fmc
