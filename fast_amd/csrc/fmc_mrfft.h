// fmc_mrfft.h -- one wavefront = one N-point row for decimal grid sizes: N = LN * P with LN = 50 of the 64 lanes at work.
//
// Users of the reference set round decimal grids (NPXLS 1000, 500, 200 ...; fast/conf.py) that are not 64 P.  The
// factorisation of fmc_wavefft.h does not need 64 = 8 x 8: with LN = L0 * L1 lanes holding the inputs
//     N = P x L1 x L0 :  in-register radix-P  ->  LDS exchange  ->  in-register radix-L1
//                        ->  LDS exchange  ->  L0-term sums for the WANTED outputs only
//   X[x] = sum_k c[k] w_N^{kx},  k = l + LN j (l < LN),  x = a + P b  (a < P, b < LN)
//        = sum_l w_N^{l a} w_LN^{l b} Z_l[a],            Z_l[a] = sum_j c[l + LN j] w_P^{ja}        (stage 1)
//   l = l0 + L0 l1, b = b0 + L1 b1 (b0 < L1):
//        = sum_l0 w_LN^{l0 b} U[a][l0][b0],   U[a][l0][b0] = sum_l1 w_L1^{l1 b0} T_{l0 + L0 l1}[a]  (stage 2a)
//   with T_l[a] = w_N^{l a} Z_l[a]; the last sum (stage 2b) only for the window outputs, on all 64 lanes.
// LN = 50 = 5 x 10 serves N = 100, 150, ..., 500, 600, ..., 1000, 1200, 1400, 1600 (P = 2^k times 1, 3, 5, 7, 9): 78 % of
// the lanes carry inputs, the stage-2a butterflies (P L0 of radix 10) and the window sums are spread over all 64, and the
// pruned sums have 5 terms instead of 8.  Sizes outside both families keep the chirp-z kernels (fmc_bluestein.h).
//
// Same executor interface as fmc_wavefft.h, so emu_wavefft.cpp runs the index arithmetic on the host.
#pragma once
#include "fmc_core.h"
#include "fmc_wavefft.h"

namespace fmc {

template <class R, int P, int LN = MR_LN>
struct MrGeom {
  static_assert(LN == 50, "mixed-radix lanes: 50 = 5 x 10");
  static constexpr int L0 = 5;                       // terms of the pruned sums (stage 2b)
  static constexpr int L1 = LN / L0;                 // in-register radix of stage 2a
  static constexpr int N = LN * P;
  static constexpr int NBF = P * L0;                 // stage-2a butterflies per row: (a, l0), owner q = a L0 + l0
  static constexpr int NB = (NBF + WAVE - 1) / WAVE; // per lane (q = lane + 64 jj)
  static constexpr int VN = (NB * L1 > P) ? NB * L1 : P;
  // exchange-1 image E[a][l] at a SE + l: the owners of 32 consecutive butterflies read (q / 5) SE + q % 5 + 5 m, and
  // SE = 5 (mod 32) makes that q (mod 32): conflict-free (8-byte elements, 32 element banks per half-wave)
  static constexpr int SE = 69;
  // exchange-2 image F[a][b0][l0] at (a L0 + l0) + NBF b0, dense: the owner of butterfly q writes q + NBF b0 (consecutive
  // lanes, consecutive elements) and window output x = a + P b reads L0 (x mod P L1) + m: stride 5 is a unit mod 32.
  static constexpr int XELEMS = (P * SE > L1 * NBF) ? P * SE : L1 * NBF;     // 8-byte elements per wave
  static_assert(VN <= lane_regs_vn(P), "LaneRegs too narrow");
};

// Tables:  tw1[a*64 + l] = w_N^{l a} (l < LN; the other lanes idle),  om[m*omS + oi] = w_LN^{m b(oi)}, m < L0 (row 0 never
// read), b(oi) = ((lo + oi) / P) mod 50.  The finished sums take the sign (-1)^(lo+oi+osign) by a sign-bit xor: the
// output-side fftshift (N even) times w_N^{-(N/2)^2} = -1 when the FULL row length is 2 (mod 4) (osign = mr_osign(N)).
template <class R, int P, int NS, int B0M = 0x3FF, class Exec>
FMC_HD void pruned_row_fft_mr(Exec& ex, typename Xch<R>::E* xbuf, const cpx<R>* tw1, const cpx<R>* om, int omS, int lo, int Np,
                              int osign = 0) {
  using G = MrGeom<R, P>;
  using X = Xch<R>;
  using E = typename X::E;
  constexpr int NC = X::NC;
  constexpr int L0 = G::L0, L1 = G::L1;
  const int nslots = (Np + WAVE - 1) / WAVE;
  // ---- stage 1: radix-P in registers, twiddle
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
  // ---- exchange 1: lane l < LN stores T_l[a]; owner q = lane + 64 jj = a L0 + l0 loads T_{l0 + L0 m}[a], m < L1
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
#pragma unroll
      for (int a = 0; a < P; ++a) ex.st(xbuf + a * G::SE + lane, X::pack(r.v[a], c));
    });
    ex.sync();
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
#pragma unroll
      for (int jj = 0; jj < G::NB; ++jj) {
        const int q = lane + WAVE * jj;
        if (q < G::NBF) {
          const E* e = xbuf + (q / L0) * G::SE + (q % L0);
#pragma unroll
          for (int m = 0; m < L1; ++m) X::unpack(r.v[jj * L1 + m], ex.ld(e + L0 * m), c);
        }
      }
    });
    ex.sync();
  }
  // ---- stage 2a: radix-L1, natural order
  ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
#pragma unroll
    for (int jj = 0; jj < G::NB; ++jj) {
      cpx<R> t[L1];
#pragma unroll
      for (int m = 0; m < L1; ++m) t[m] = r.v[jj * L1 + m];
      dft_reg<L1, R>(t);
#pragma unroll
      for (int b0 = 0; b0 < L1; ++b0) r.v[jj * L1 + b0] = t[b0];
    }
  });
  // ---- exchange 2 + stage 2b (pruned): L0-term sums for the window outputs
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
#pragma unroll
      for (int jj = 0; jj < G::NB; ++jj) {
        const int q = lane + WAVE * jj;
        if (q < G::NBF) {
#pragma unroll
          for (int b0 = 0; b0 < L1; ++b0)
            if ((B0M >> b0) & 1) ex.st(xbuf + q + G::NBF * b0, X::pack(r.v[jj * L1 + b0], c));   // planes no output reads: neither stored nor computed
        }
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
            const E* f = xbuf + L0 * (x % (P * L1));       // = L0 a + NBF b0
            X::first(r.xr[s], r.xi[s], ex.ld(f), c);
#pragma unroll
            for (int m = 1; m < L0; ++m) X::acc(r.xr[s], r.xi[s], om[m * omS + oi], ex.ld(f + m), c);
            if (c == NC - 1) {
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

// osign of a row of N points (S sub-rows included): 1 when N = 2 (mod 4)
FMC_HD constexpr int mr_osign(int N) { return (N % 4 == 2) ? 1 : 0; }

// Host-side tables (float64 trigonometry by `cs(turns, &c, &s)`).
template <class R, class CosSin>
inline void build_tw1_mr(cpx<R>* tw1, int P, CosSin cs) {
  const int N = MR_LN * P;
  for (int a = 0; a < P; ++a)
    for (int l = 0; l < WAVE; ++l) {
      double c = 1.0, s = 0.0;
      if (l < MR_LN) cs((double)((l * a) % N) / N, &c, &s);
      tw1[a * WAVE + l] = mk<R>((R)c, (R)(-s));
    }
}
template <class R, class CosSin>
inline void build_om_mr(cpx<R>* om, int omS, int P, int lo, int Np, CosSin cs) {
  for (int m = 0; m < 5; ++m)
    for (int oi = 0; oi < omS; ++oi) {
      if (oi >= Np) { om[m * omS + oi] = mk<R>((R)0, (R)0); continue; }
      const int x = lo + oi;
      const int b = (x / P) % MR_LN;
      double c, s;
      cs((double)((m * b) % MR_LN) / (double)MR_LN, &c, &s);
      om[m * omS + oi] = mk<R>((R)c, (R)(-s));
    }
}

}  // namespace fmc
