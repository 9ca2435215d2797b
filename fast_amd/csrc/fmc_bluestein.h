// fmc_bluestein.h -- arbitrary row length N on the wave pipeline: chirp-z (Bluestein) transform, window outputs only.
//
// The reference auto-sizes its grid to arbitrary even N (fast/fast.py:176-211: 164 for the shipped example) and
// users set e.g. NPXLS 1000; the wave kernels of fmc_wavefft.h need N = 64 P.  With a = k + h, b = p - h (h = N // 2,
// numpy's fftshift on both sides, fast/funcs.py:213-215) the wanted outputs are
//     out[p] = sum_k in[k] w_N^{a b},      a b = (a^2 + b^2 - (b - a)^2) / 2
//            = w_2N^{b^2} sum_k (in[k] w_2N^{a^2}) w_2N^{-(b - a)^2}
// i.e. a correlation of the pre-chirped row u[k] = in[k] w_2N^{(k+h)^2} with the chirp c(n) = w_2N^{-n^2}, needed for
// the Np window outputs only: a cyclic convolution of length M = 64 P >= N + Np - 1,
//     y = IDFT_M( DFT_M(u) . DFT_M(v) ),   v[m] = c(m + lo - 2h) for m < Np,  c(m - M + lo - 2h) for m > M - N,
//     out[lo + t] = w_2N^{b_t^2} y[t],  t < Np.
// One wavefront does the whole row in registers and LDS:
//     forward DFT_M with ALL outputs (full_row_fft: the pipeline of fmc_wavefft.h with a real third radix-8 stage
//     instead of the pruned 8-term sums)  ->  multiply by V^ = DFT_M(v), conjugate  ->  one more LDS exchange back to
//     the k = lane + 64 j layout  ->  pruned_row_fft for the window [0, Np)  ->  conjugate, multiply by post[t].
// (IDFT(z) = conj(DFT(conj z)) / M; the 1 / M lives in `post`.)  About 2.7 plain rows of work for any N.
//
// Rows longer than the largest M (N + Np - 1 > 2048) are cut into SB blocks of B inputs (B a multiple of 64, B + Np - 1 <= M):
//     out[p] = post[p] conj( sum_j Y_j[t] ),   Y_j = the chirp-z row of block j: inputs k in [j B, (j + 1) B) pre-chirped with
//     the GLOBAL pre[k], chirp kernel v_j[m] = c(m + lo - 2h - j B) (its own V^_j table),
// the window sums accumulated in registers over the blocks (the convolution is linear in the input).  With B a multiple of
// 64 lane l of block j continues generator stream l where block j - 1 left it.  2200^2, 2816^2 ... 4096-sized grids that are
// neither 64 P S nor 50 P S run SB = 3 ... 5 blocks on the M = 1024 pipeline instead of the O(N^2 Np) direct kernels.
//
// Like fmc_wavefft.h the per-lane phases are written against an executor, so that emu_wavefft.cpp runs the same index
// arithmetic on the host.
#pragma once
#include "fmc_core.h"
#include "fmc_wavefft.h"

namespace fmc {

// LDS images of the two extra exchanges (8-byte elements, conflict-free by construction; searched with
// tools/lds_bank_check.py under the lane-group rules of MI355X_MICROARCH.md):
//   full stage 2b : F[a][b0][l0] at a + FL l0 + FBF b0, written by lane (l0 = lane & 7, i = lane >> 3) for a = i + 8 jj,
//                   read by lane (i = lane & 7, b0 = lane >> 3): 16-lane write groups and 32-lane read groups hit
//                   distinct banks for FL = P + 2 and FBF = 8 P + 24 (46 for P = 4);
//   re-layout     : Z[x] at swz(x), written in the (a, b0; b1) layout of stage 2b, read as x = lane + 64 j.
template <class R, int P>
struct BluGeom {
  using G = WaveGeom<R, P>;
  static constexpr int M = WAVE * P;
  static constexpr int FL = P + 2;
  static constexpr int FBF = (P == 4) ? 46 : 8 * P + 24;
  static constexpr int X1 = G::XELEMS > 8 * FBF ? G::XELEMS : 8 * FBF;
  static constexpr int XELEMS = X1 > M ? X1 : M;
  // (round 6: also 12 and 28 -- M = 768, 1792 -- where 1024 / 2048 are a third / a seventh more work; in stage 2 their last group of
  // eight butterflies runs on 4 of every 8 lanes.  20 (M = 1280) works -- the lane emulator covers it -- but its rows spill at eight
  // waves and lose to M = 1536: not instantiated in the library)
  static_assert(P == 4 || P == 8 || P == 12 || P == 16 || P == 20 || P == 24 || P == 28 || P == 32, "chirp-z sizes: M = 256, 512, 768, 1024, (1280,) 1536, 1792, 2048");
  static FMC_HD int swz(int x) {
    if (P == 16) return x ^ (((x >> 3) & 3) << 2);
    if (P == 32) return x ^ (((x >> 4) & 3) << 2);
    return x;
  }
};

// Forward DFT_M of the P values per lane (k = lane + 64 j), ALL outputs, left in registers:
// lane (i = lane & 7, b0 = lane >> 3), slot jj * 8 + b1  <->  X[(i + 8 jj) + P (b0 + 8 b1)]   (lanes with i + 8 jj >= P idle).
//   twf[b0 * 8 + l0] = w_64^{l0 b0}
template <class R, int P, int NS, class Exec>
FMC_HD void full_row_fft(Exec& ex, typename Xch<R>::E* xbuf, const cpx<R>* tw1, const cpx<R>* twf) {
  using G = WaveGeom<R, P>;
  using B = BluGeom<R, P>;
  using X = Xch<R>;
  constexpr int NC = X::NC;
  // ---- stage 1: radix-P in registers, twiddle
  ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
    cpx<R> z[P];
#pragma unroll
    for (int j = 0; j < P; ++j) z[j] = r.v[j];
    dft_reg<P, R>(z);
    r.v[0] = z[0];
#pragma unroll
    for (int a = 1; a < P; ++a) r.v[a] = cmul(z[a], tw1[a * WAVE + lane]);
  });
  // ---- exchange 1 + stage 2a (as in pruned_row_fft)
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
          for (int m = 0; m < 8; ++m) X::unpack(r.v[jj * 8 + m], ex.ld(xbuf + (i + 8 * jj) * G::SE + l0 + 8 * m), c);
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
      for (int b0 = 0; b0 < 8; ++b0) r.v[jj * 8 + b0] = t[brev(b0, 3)];
    }
  });
  // ---- exchange 2 (full image) + stage 2b as a real radix-8 stage
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
      const int l0 = lane & 7, i = lane >> 3;
#pragma unroll
      for (int jj = 0; jj < G::NB; ++jj)
        if ((P % 8 == 0) || i + 8 * jj < P) {
#pragma unroll
          for (int b0 = 0; b0 < 8; ++b0) ex.st(xbuf + (i + 8 * jj) + B::FL * l0 + B::FBF * b0, X::pack(r.v[jj * 8 + b0], c));
        }
    });
    ex.sync();
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
      const int i = lane & 7, b0 = lane >> 3;
#pragma unroll
      for (int jj = 0; jj < G::NB; ++jj)
        if ((P % 8 == 0) || i + 8 * jj < P) {
#pragma unroll
          for (int l0 = 0; l0 < 8; ++l0) X::unpack(r.v[jj * 8 + l0], ex.ld(xbuf + (i + 8 * jj) + B::FL * l0 + B::FBF * b0), c);
        }
    });
    ex.sync();
  }
  ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
    const int i = lane & 7, b0 = lane >> 3;
#pragma unroll
    for (int jj = 0; jj < G::NB; ++jj)
      if ((P % 8 == 0) || i + 8 * jj < P) {
        cpx<R> t[8];
        t[0] = r.v[jj * 8];
#pragma unroll
        for (int l0 = 1; l0 < 8; ++l0) t[l0] = cmul(r.v[jj * 8 + l0], twf[b0 * 8 + l0]);
        fft_dif<8, R>(t);
#pragma unroll
        for (int b1 = 0; b1 < 8; ++b1) r.v[jj * 8 + b1] = t[brev(b1, 3)];
      }
  });
}

// The whole chirp-z row: on entry r.v[j] = u[lane + 64 j] (pre-chirped, zero beyond N); on return slot s of lane l
// holds Y[t], t = l + 64 s < Np, with out[lo + t] = post[t] * conj(Y[t]).
//   vhat: DFT_M(v) in natural order (global memory / L2);  om: tables of pruned_row_fft for the window [0, Np) (run without the fftshift sign).
template <class R, int P, int NS, class Exec>
FMC_HD void bluestein_row(Exec& ex, typename Xch<R>::E* xbuf, const cpx<R>* tw1, const cpx<R>* om, int omS,
                          const cpx<R>* twf, const cpx<R>* vhat, int Np) {
  using G = WaveGeom<R, P>;
  using B = BluGeom<R, P>;
  using X = Xch<R>;
  constexpr int NC = X::NC;
  full_row_fft<R, P, NS>(ex, xbuf, tw1, twf);
  // pointwise product with V^, conjugated (the inverse transform is done as conj(DFT(conj .)))
  ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
    const int i = lane & 7, b0 = lane >> 3;
#pragma unroll
    for (int jj = 0; jj < G::NB; ++jj)
      if ((P % 8 == 0) || i + 8 * jj < P) {
#pragma unroll
        for (int b1 = 0; b1 < 8; ++b1) {
          const cpx<R> v = vhat[(i + 8 * jj) + P * (b0 + 8 * b1)];
          const cpx<R> u = r.v[jj * 8 + b1];
          r.v[jj * 8 + b1] = mk<R>(u.x * v.x - u.y * v.y, -(u.x * v.y + u.y * v.x));
        }
        ex.loadfence();      // one group of 8 table loads in flight at a time: P of them would hold 4 P registers
      }
  });
  // back to the input layout of the pipeline: x = lane + 64 j
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
      const int i = lane & 7, b0 = lane >> 3;
#pragma unroll
      for (int jj = 0; jj < G::NB; ++jj)
        if ((P % 8 == 0) || i + 8 * jj < P) {
#pragma unroll
          for (int b1 = 0; b1 < 8; ++b1)
            ex.st(xbuf + B::swz((i + 8 * jj) + P * (b0 + 8 * b1)), X::pack(r.v[jj * 8 + b1], c));
        }
    });
    ex.sync();
    ex.each([&](int lane, LaneRegs<R, P, NS>& r) {
#pragma unroll
      for (int j = 0; j < P; ++j) X::unpack(r.v[j], ex.ld(xbuf + B::swz(lane + WAVE * j)), c);
    });
    ex.sync();
  }
  pruned_row_fft<R, P, NS>(ex, xbuf, tw1, om, omS, 0, Np, -1);
}

// ---------------------------------------------------------------- host-side tables (float64 trigonometry by `cs`)
// pre[k]  = w_2N^{(k+h)^2}, k < N, zero up to pre_len  (input chirp, with the input-side fftshift; GLOBAL index k)
// vhat    = [SB][M]: DFT_M(v_j), v_j the chirp kernel of the window for input block j (see the header; SB = 1, B = N: one block)
// post[t] = w_2N^{(lo - h + t)^2} / M, t < Np          (output chirp, with the output-side fftshift and the 1 / M of the IDFT)
// twf[b0 * 8 + l0] = w_64^{l0 b0}
// Returns false when M < B + Np - 1 (the cyclic convolution would alias) or the blocks do not cover the row.
template <class R, class CosSin>
inline bool build_blu_tables(int N, int Np, int lo, int P, cpx<R>* pre, cpx<R>* vhat, cpx<R>* post, int post_len, cpx<R>* twf,
                             CosSin cs, int B = 0, int SB = 1, int pre_len = 0) {
  const int M = WAVE * P, h = N / 2;
  if (B <= 0) B = N;
  if (pre_len <= 0) pre_len = M;
  if (M < B + Np - 1 || (long long)B * SB < N || B > M) return false;
  auto chirp = [&](long long n, double sign, double* c, double* s) {   // exp(sign * i pi n^2 / N)
    const long long q = (n * n) % (2LL * N);
    double cc, ss;
    cs((double)q / (2.0 * N), &cc, &ss);
    *c = cc;
    *s = sign * ss;
  };
  for (int k = 0; k < pre_len; ++k) {
    if (k < N) {
      double c, s;
      chirp(k + h, -1.0, &c, &s);
      pre[k] = mk<R>((R)c, (R)s);
    } else {
      pre[k] = mk<R>((R)0, (R)0);
    }
  }
  for (int t = 0; t < post_len; ++t) {
    if (t < Np) {
      double c, s;
      chirp(lo - h + t, -1.0, &c, &s);
      post[t] = mk<R>((R)(c / M), (R)(s / M));
    } else {
      post[t] = mk<R>((R)0, (R)0);
    }
  }
  // v_j and its DFT (naive O(M^2) in long double with exact twiddle indices: once per problem)
  struct LD { long double x, y; };
  LD* v = new LD[M];
  LD* w = new LD[M];
  for (int m = 0; m < M; ++m) {
    double c, s;
    cs((double)m / M, &c, &s);
    w[m].x = c;
    w[m].y = -s;                                  // w_M^m
  }
  for (int jb = 0; jb < SB; ++jb) {
    const long long off = (long long)lo - 2 * h - (long long)jb * B;     // lag of output t = 0 against the block's first input
    for (int m = 0; m < M; ++m) {
      v[m].x = v[m].y = 0;
      long long n = 0;
      bool used = false;
      if (m < Np) { n = (long long)m + off; used = true; }
      else if (m > M - B) { n = (long long)m - M + off; used = true; }
      if (used) {
        double c, s;
        chirp(n < 0 ? -n : n, +1.0, &c, &s);
        v[m].x = c;
        v[m].y = s;
      }
    }
    for (int q = 0; q < M; ++q) {
      long double ar = 0, ai = 0;
      int e = 0;
      for (int m = 0; m < M; ++m) {
        ar += v[m].x * w[e].x - v[m].y * w[e].y;
        ai += v[m].x * w[e].y + v[m].y * w[e].x;
        e += q;
        if (e >= M) e -= M;
      }
      vhat[(size_t)jb * M + q] = mk<R>((R)ar, (R)ai);
    }
  }
  delete[] v;
  delete[] w;
  for (int b0 = 0; b0 < 8; ++b0)
    for (int l0 = 0; l0 < 8; ++l0) {
      double c, s;
      cs((double)((l0 * b0) % 64) / 64.0, &c, &s);
      twf[b0 * 8 + l0] = mk<R>((R)c, (R)(-s));
    }
  return true;
}

// ---------------------------------------------------------------- chirp-z rows on the PACKED 256-point pipeline (round 6)
// The row above costs 3.4-4.4 plain rows: one wavefront per row, a full forward transform of M >= N + Np - 1 points with P = M / 64
// values per lane (M from seven sizes), a product, a pruned inverse.  The convolution is linear in the input, so a row can be cut into
// blocks of PBZ_B = 128 inputs, each a cyclic convolution of length 256 >= 128 + Np - 1 (Np <= 129):
//     y = IDFT_256( sum_j DFT_256(u_j) . V^_j ),      u_j = the block's pre-chirped inputs, zero-padded;  V^_j = DFT_256(v_j)
// -- the PRODUCT SPECTRA are accumulated (sixteen complex registers per lane) and ONE inverse transform per row follows.  Forward and
// inverse are packed_row_fft<L0 = 1>: FOUR rows per wavefront, sixteen lanes per row, sixteen values per lane whatever N; its output
// layout (lane a, register b: x = a + 16 b) IS its input layout (lane q, register j: k = q + 16 j), so nothing moves between the two.
// ceil(N / 128) + 1 transforms of 256 points per row: about 2.3 packed rows of work.  packed_row_fft applies the output-side
// fftshift sign (-1)^x of the plain 256-point grid to the planes it keeps: the tables carry it (build_pbz_tables).
constexpr int PBZ_B = 128, PBZ_M = 256;
// one block: on entry r.v[j] = u[k0 + q + 16 j], j < 8, r.v[8 ... 15] = 0;  acc[b] += X[a + 16 b] V^_j[a + 16 b]
template <class R, class Exec, class AccOf>
FMC_HD void pbz_block(Exec& ex, typename Xch<R>::E* xbuf, const cpx<R>* tw, const cpx<R>* vhat_j, AccOf acc_of) {
  packed_row_fft<R, 1, 1, 0xFFFF>(ex, xbuf, tw, (const cpx<R>*)nullptr, 0, 0, 0);
  ex.each([&](int lane, LaneRegs<R, 16, 1>& r) {
    cpx<R>* acc = acc_of(lane);
    const cpx<R>* v = vhat_j + (lane & 15);
#pragma unroll
    for (int b = 0; b < 16; ++b) acc[b] = cfma(r.v[b], ex.ld(v + 16 * b), acc[b]);
  });
}
// after the last block: r.v[p], p < NPL, of lane (g, a) = Y'[t], t = a + 16 p, with out[lo + t] = post[t] conj(Y'[t])
template <class R, int NPL, class Exec, class AccOf>
FMC_HD void pbz_finish(Exec& ex, typename Xch<R>::E* xbuf, const cpx<R>* tw, AccOf acc_of) {
  ex.each([&](int lane, LaneRegs<R, 16, 1>& r) {
    const cpx<R>* acc = acc_of(lane);
#pragma unroll
    for (int b = 0; b < 16; ++b) r.v[b] = mk<R>(acc[b].x, -acc[b].y);
  });
  packed_row_fft<R, 1, 1, (1 << NPL) - 1>(ex, xbuf, tw, (const cpx<R>*)nullptr, 0, 0, 0);
}
// pre[k] (SB * 128 entries, zero beyond N), vhat [SB][256], post[t] (128 entries): build_blu_tables with M = 256, B = 128, and the
// plane signs of packed_row_fft folded in.  SB = ceil(N / 128).  false: the window does not fit (Np > 129).
template <class R, class CosSin>
inline bool build_pbz_tables(int N, int Np, int lo, cpx<R>* pre, cpx<R>* vhat, cpx<R>* post, CosSin cs) {
  const int SB = (N + PBZ_B - 1) / PBZ_B;
  cpx<R> twf[64];
  if (!build_blu_tables<R>(N, Np, lo, PBZ_M / WAVE, pre, vhat, post, 128, twf, cs, PBZ_B, SB, SB * PBZ_B)) return false;
  for (int jb = 0; jb < SB; ++jb)
    for (int x = 1; x < PBZ_M; x += 2) { cpx<R>& v = vhat[(size_t)jb * PBZ_M + x]; v = mk<R>(-v.x, -v.y); }
  for (int t = 1; t < 128; t += 2) post[t] = mk<R>(-post[t].x, -post[t].y);
  return true;
}

// Blocks of the chirp-z row when no single M holds it: the M = 1024 pipeline (P = 16), B = the largest multiple of 64 with
// B + Np - 1 <= 1024.  Returns the number of blocks (0: Np too wide).
FMC_HD constexpr int blu_block_len(int Np) { return ((1024 - Np + 1) / 64) * 64; }
FMC_HD constexpr int blu_blocks(int N, int Np) {
  const int B = blu_block_len(Np);
  return (B < 64 || Np > 256) ? 0 : (N + B - 1) / B;
}

}  // namespace fmc
