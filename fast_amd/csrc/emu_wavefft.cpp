// emu_wavefft.cpp -- host-side lane emulation of fmc_wavefft.h (no GPU needed).
// Runs every per-lane phase of pruned_row_fft in a loop over 64 lanes, with the LDS images in
// ordinary arrays, and compares the window outputs with a naive O(N^2) DFT in long double,
// including the fftshift semantics of the reference (fast/funcs.py:213-215).
// Exit code 0 = all cases within tolerance.  Driven by tests/test_emu_wavefft.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "fmc_core.h"
#include "fmc_wavefft.h"
#include "fmc_bluestein.h"
#include "fmc_mrfft.h"

using namespace fmc;

template <class R, int P, int NS>
struct HostExec {
  LaneRegs<R, P, NS> regs[WAVE];
  template <class F> void each(F f) {
    for (int l = 0; l < WAVE; ++l) f(l, regs[l]);
  }
  void sync() {}
  static void loadfence() {}
  template <class T> static void pin(T&) {}
  template <class E> static E ld(const E* p) { return *p; }
  template <class E> static void st(E* p, E v) { *p = v; }
};

static void cs_turns(double t, double* c, double* s) {
  *c = std::cos(2.0 * M_PI * t);
  *s = std::sin(2.0 * M_PI * t);
}

static int g_dense_r16 = 0;       // P = 16: 1 = the 16 x 4 lane factorisation (pruned_row_fft_d16r), 2 = with the centred plane set

template <class R, int P, int NS>
static double run_case(int lo, int Np, unsigned seed) {
  constexpr int N = WAVE * P;
  using G = WaveGeom<R, P>;
  using E = typename Xch<R>::E;
  std::mt19937_64 gen(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> inr(N), ini(N);
  for (int k = 0; k < N; ++k) { inr[k] = nd(gen); ini[k] = nd(gen); }

  const int omS = NS * WAVE;
  std::vector<cpx<R>> tw1((size_t)P * WAVE), om((size_t)8 * omS);
  build_tw1<R>(tw1.data(), P, cs_turns);
  build_om<R>(om.data(), omS, P, lo, Np, cs_turns);
  // The GPU kernels do not stage row 0 of either table (w^0 = 1; fmc_kernels.h: WaveLds hands the rows VIRTUAL bases one row below
  // the LDS block): the invariant "no row reads tw1[0][.] or om[0][.]" is checked here, where the same code runs on the host --
  // row 0 poisoned with NaN must leave every output finite
  for (int l = 0; l < WAVE; ++l) tw1[l] = mk<R>((R)NAN, (R)NAN);
  for (int oi = 0; oi < omS; ++oi) om[oi] = mk<R>((R)NAN, (R)NAN);
  std::vector<E> xbuf(G::XELEMS);

  static HostExec<R, P, NS> ex;
  // the kernels fold the input-side fftshift sign (-1)^k into the spectrum amplitude
  for (int l = 0; l < WAVE; ++l)
    for (int j = 0; j < P; ++j) {
      const int k = l + WAVE * j;
      const double sg = (k & 1) ? -1.0 : 1.0;
      ex.regs[l].v[j] = mk<R>((R)(sg * inr[k]), (R)(sg * ini[k]));
    }
  if constexpr (P == 16) {
    if (g_dense_r16 == 1) pruned_row_fft_d16r<R, NS>(ex, xbuf.data(), tw1.data(), om.data(), omS, lo, Np);
    else if (g_dense_r16 == 2) pruned_row_fft_d16r<R, NS, D16R_CENTRE_MASK>(ex, xbuf.data(), tw1.data(), om.data(), omS, lo, Np);
    else pruned_row_fft<R, P, NS>(ex, xbuf.data(), tw1.data(), om.data(), omS, lo, Np);
  } else {
    pruned_row_fft<R, P, NS>(ex, xbuf.data(), tw1.data(), om.data(), omS, lo, Np);
  }

  // reference: fftshift(fft(fftshift(in)))[p], p = lo + oi   (even N: h = N/2 both ways)
  double worst = 0.0, scale = 0.0;
  const int h = N / 2;
  for (int oi = 0; oi < Np; ++oi) {
    const int p = lo + oi;
    long double sr = 0, si = 0;
    for (int k = 0; k < N; ++k) {
      // out[p] = sum_k in[k] w^{(p-h)(k+h)}
      const long long e = (((long long)(p - h) * (k + h)) % N + N) % N;
      const long double a = -2.0L * M_PIl * (long double)e / N;
      const long double c = cosl(a), s = sinl(a);
      sr += inr[k] * c - ini[k] * s;
      si += inr[k] * s + ini[k] * c;
    }
    const int l = oi % WAVE, s = oi / WAVE;
    const double gr = ex.regs[l].xr[s], gi = ex.regs[l].xi[s];
    worst = std::fmax(worst, std::fmax(std::fabs(gr - (double)sr), std::fabs(gi - (double)si)));
    scale = std::fmax(scale, std::fmax(std::fabs((double)sr), std::fabs((double)si)));
  }
  return worst / scale;
}

template <class R, int P, int NS>
static int sweep(const char* name, double tol) {
  constexpr int N = WAVE * P;
  int bad = 0;
  const int cases[][2] = {{(N - 82) / 2, 82}, {0, 64 * NS < N ? 64 * NS : N}, {N - 5, 5}, {(N - 23) / 2, 23},
                          {(N - 1) / 2, 1}, {7, 64 * NS - 3 < N - 7 ? 64 * NS - 3 : N - 7}, {(N - 128) / 2, NS >= 2 ? 128 : 64}};
  for (auto& c : cases) {
    if (c[1] > 64 * NS || c[0] + c[1] > N || c[1] < 1) continue;
    const double err = run_case<R, P, NS>(c[0], c[1], 1234u + c[0]);
    const bool ok = err <= tol;
    std::printf("%s P=%d NS=%d lo=%d Np=%d relerr=%.3e %s\n", name, P, NS, c[0], c[1], err, ok ? "ok" : "FAIL");
    bad += !ok;
  }
  return bad;
}

// Chirp-z row of fmc_bluestein.h: any N (odd included) with M = 64 P >= N + Np - 1, against the naive DFT with
// numpy's fftshift on both sides: out[p] = sum_k in[k] w_N^{(p - h)(k + h)}, h = N // 2.
template <class R, int P, int NS>
static double run_blu_case(int N, int lo, int Np, unsigned seed) {
  constexpr int M = WAVE * P;
  using B = BluGeom<R, P>;
  using E = typename Xch<R>::E;
  std::mt19937_64 gen(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> inr(N), ini(N);
  for (int k = 0; k < N; ++k) { inr[k] = nd(gen); ini[k] = nd(gen); }
  const int omS = NS * WAVE;
  std::vector<cpx<R>> tw1((size_t)P * WAVE), om((size_t)8 * omS), pre(M), vhat(M), post(omS), twf(64);
  build_tw1<R>(tw1.data(), P, cs_turns);
  build_om<R>(om.data(), omS, P, 0, Np, cs_turns);
  if (!build_blu_tables<R>(N, Np, lo, P, pre.data(), vhat.data(), post.data(), omS, twf.data(), cs_turns)) return 1e30;
  std::vector<E> xbuf(B::XELEMS);
  static HostExec<R, P, NS> ex;
  for (int l = 0; l < WAVE; ++l)
    for (int j = 0; j < P; ++j) {
      const int k = l + WAVE * j;
      ex.regs[l].v[j] = k < N ? cmul(mk<R>((R)inr[k], (R)ini[k]), pre[k]) : mk<R>((R)0, (R)0);
    }
  bluestein_row<R, P, NS>(ex, xbuf.data(), tw1.data(), om.data(), omS, twf.data(), vhat.data(), Np);
  double worst = 0.0, scale = 0.0;
  const int h = N / 2;
  for (int oi = 0; oi < Np; ++oi) {
    const int p = lo + oi;
    long double sr = 0, si = 0;
    for (int k = 0; k < N; ++k) {
      const long long e = (((long long)(p - h) * (k + h)) % N + N) % N;
      const long double a = -2.0L * M_PIl * (long double)e / N;
      const long double c = cosl(a), s2 = sinl(a);
      sr += inr[k] * c - ini[k] * s2;
      si += inr[k] * s2 + ini[k] * c;
    }
    const int l = oi % WAVE, s = oi / WAVE;
    const double yr = ex.regs[l].xr[s], yi = ex.regs[l].xi[s];
    const double gr = post[oi].x * yr + post[oi].y * yi, gi = post[oi].y * yr - post[oi].x * yi;   // post * conj(Y)
    worst = std::fmax(worst, std::fmax(std::fabs(gr - (double)sr), std::fabs(gi - (double)si)));
    scale = std::fmax(scale, std::fmax(std::fabs((double)sr), std::fabs((double)si)));
  }
  return worst / scale;
}

// Chirp-z rows on the packed 256-point pipeline (fmc_bluestein.h: pbz_block / pbz_finish): four rows of any length N per wavefront in
// blocks of 128 inputs, against the naive DFT with numpy's fftshift on both sides.
template <class R, int NPL>
static double run_pbz_case(int N, int lo, int Np, unsigned seed) {
  using E = typename Xch<R>::E;
  std::mt19937_64 gen(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  const int G = 4, SB = (N + PBZ_B - 1) / PBZ_B;
  std::vector<double> inr((size_t)G * N), ini((size_t)G * N);
  for (size_t k = 0; k < inr.size(); ++k) { inr[k] = nd(gen); ini[k] = nd(gen); }
  std::vector<cpx<R>> tw(256), pre((size_t)SB * PBZ_B), vhat((size_t)SB * PBZ_M), post(128);
  build_tw1_pk<R>(tw.data(), 16, cs_turns);
  if (!build_pbz_tables<R>(N, Np, lo, pre.data(), vhat.data(), post.data(), cs_turns)) return 1e30;
  std::vector<E> xbuf(D16_XELEMS);
  static HostExec<R, 16, 1> ex;
  static cpx<R> accs[WAVE][16];
  for (int l = 0; l < WAVE; ++l)
    for (int b = 0; b < 16; ++b) accs[l][b] = mk<R>((R)0, (R)0);
  auto acc_of = [&](int l) { return accs[l]; };
  for (int jb = 0; jb < SB; ++jb) {
    for (int l = 0; l < WAVE; ++l) {
      const int g = l >> 4, q = l & 15;
      for (int j = 0; j < 16; ++j) {
        const int k = jb * PBZ_B + q + 16 * j;
        ex.regs[l].v[j] = (j < 8 && k < N) ? cmul(mk<R>((R)inr[(size_t)g * N + k], (R)ini[(size_t)g * N + k]), pre[k]) : mk<R>((R)0, (R)0);
      }
    }
    pbz_block<R>(ex, xbuf.data(), tw.data(), vhat.data() + (size_t)jb * PBZ_M, acc_of);
  }
  pbz_finish<R, NPL>(ex, xbuf.data(), tw.data(), acc_of);
  double worst = 0.0, scale = 0.0;
  const int h = N / 2;
  for (int g = 0; g < G; ++g)
    for (int oi = 0; oi < Np; ++oi) {
      const int p = lo + oi;
      long double sr = 0, si = 0;
      for (int k = 0; k < N; ++k) {
        const long long e = (((long long)(p - h) * (k + h)) % N + N) % N;
        const long double a = -2.0L * M_PIl * (long double)e / N;
        const long double c = cosl(a), s2 = sinl(a);
        sr += inr[(size_t)g * N + k] * c - ini[(size_t)g * N + k] * s2;
        si += inr[(size_t)g * N + k] * s2 + ini[(size_t)g * N + k] * c;
      }
      const int l = 16 * g + (oi & 15), pl = oi >> 4;
      const double yr = ex.regs[l].v[pl].x, yi = ex.regs[l].v[pl].y;
      const double gr = post[oi].x * yr + post[oi].y * yi, gi = post[oi].y * yr - post[oi].x * yi;   // post * conj(Y)
      worst = std::fmax(worst, std::fmax(std::fabs(gr - (double)sr), std::fabs(gi - (double)si)));
      scale = std::fmax(scale, std::fmax(std::fabs((double)sr), std::fabs((double)si)));
    }
  return worst / scale;
}
template <class R, int NPL>
static int sweep_pbz(const char* name, double tol) {
  int bad = 0;
  const int cases[][3] = {{164, 41, 82}, {49, 13, 23}, {100, 0, 64}, {1002, 460, 82}, {998, 902, 96}, {333, 100, 128}, {97, 96, 1}, {1111, 0, 40},
                          {128, 23, 82}, {129, 0, 96}, {2050, 984, 82}, {640, 500, 128}};
  for (auto& c : cases) {
    const int N = c[0], lo = c[1], Np = c[2];
    if (Np > 16 * NPL || Np < 1 || lo < 0 || lo + Np > N) continue;
    const double err = run_pbz_case<R, NPL>(N, lo, Np, 31u + N);
    const bool ok = err <= tol;
    std::printf("%s packed chirp-z planes=%d N=%d lo=%d Np=%d relerr=%.3e %s\n", name, NPL, N, lo, Np, err, ok ? "ok" : "FAIL");
    bad += !ok;
  }
  return bad;
}

// The same row cut into input blocks (fmc_bluestein.h header): SB chirp-z rows of B inputs on the M = 1024 pipeline, window
// sums accumulated over the blocks -- the form grids longer than the largest M take (2200, 2816, ... <= 4096).
template <class R, int NS>
static double run_blu_blocked_case(int N, int lo, int Np, unsigned seed) {
  constexpr int P = 16, M = WAVE * P;
  using B_ = BluGeom<R, P>;
  using E = typename Xch<R>::E;
  const int B = blu_block_len(Np), SB = blu_blocks(N, Np);
  std::mt19937_64 gen(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> inr(N), ini(N);
  for (int k = 0; k < N; ++k) { inr[k] = nd(gen); ini[k] = nd(gen); }
  const int omS = NS * WAVE;
  std::vector<cpx<R>> tw1((size_t)P * WAVE), om((size_t)8 * omS), pre((size_t)SB * B), vhat((size_t)SB * M), post(omS), twf(64);
  build_tw1<R>(tw1.data(), P, cs_turns);
  build_om<R>(om.data(), omS, P, 0, Np, cs_turns);
  if (!build_blu_tables<R>(N, Np, lo, P, pre.data(), vhat.data(), post.data(), omS, twf.data(), cs_turns, B, SB, SB * B)) return 1e30;
  std::vector<E> xbuf(B_::XELEMS);
  static HostExec<R, P, NS> ex;
  std::vector<double> accr((size_t)WAVE * NS, 0.0), acci((size_t)WAVE * NS, 0.0);
  for (int jb = 0; jb < SB; ++jb) {
    for (int l = 0; l < WAVE; ++l)
      for (int j = 0; j < P; ++j) {
        const int kl = l + WAVE * j, k = jb * B + kl;
        ex.regs[l].v[j] = (kl < B && k < N) ? cmul(mk<R>((R)inr[k], (R)ini[k]), pre[k]) : mk<R>((R)0, (R)0);
      }
    bluestein_row<R, P, NS>(ex, xbuf.data(), tw1.data(), om.data(), omS, twf.data(), vhat.data() + (size_t)jb * M, Np);
    for (int l = 0; l < WAVE; ++l)
      for (int s = 0; s < NS; ++s) { accr[l * NS + s] += ex.regs[l].xr[s]; acci[l * NS + s] += ex.regs[l].xi[s]; }
  }
  double worst = 0.0, scale = 0.0;
  const int h = N / 2;
  for (int oi = 0; oi < Np; ++oi) {
    const int p = lo + oi;
    long double sr = 0, si = 0;
    for (int k = 0; k < N; ++k) {
      const long long e = (((long long)(p - h) * (k + h)) % N + N) % N;
      const long double a = -2.0L * M_PIl * (long double)e / N;
      const long double c = cosl(a), s2 = sinl(a);
      sr += inr[k] * c - ini[k] * s2;
      si += inr[k] * s2 + ini[k] * c;
    }
    const int l = oi % WAVE, s = oi / WAVE;
    const double yr = accr[l * NS + s], yi = acci[l * NS + s];
    const double gr = post[oi].x * yr + post[oi].y * yi, gi = post[oi].y * yr - post[oi].x * yi;   // post * conj(sum Y_j)
    worst = std::fmax(worst, std::fmax(std::fabs(gr - (double)sr), std::fabs(gi - (double)si)));
    scale = std::fmax(scale, std::fmax(std::fabs((double)sr), std::fabs((double)si)));
  }
  return worst / scale;
}

template <class R, int NS>
static int sweep_blu_blocked(const char* name, double tol) {
  int bad = 0;
  const int cases[][3] = {{2200, 1059, 82}, {2816, 0, 82}, {4000, 3918, 82}, {2049, 1000, 127}, {4095, 2000, 64 * NS}, {1971, 944, 82},
                          {1100, 500, 82}, {2200, 1000, 200}};
  for (auto& c : cases) {
    const int N = c[0], lo = c[1], Np = c[2];
    if (Np > 64 * NS || lo < 0 || lo + Np > N || blu_blocks(N, Np) < 2) continue;
    const double err = run_blu_blocked_case<R, NS>(N, lo, Np, 7u + N);
    const bool ok = err <= tol;
    std::printf("%s blocked chirp-z NS=%d N=%d lo=%d Np=%d blocks=%d x %d relerr=%.3e %s\n", name, NS, N, lo, Np, blu_blocks(N, Np),
                blu_block_len(Np), err, ok ? "ok" : "FAIL");
    bad += !ok;
  }
  return bad;
}

template <class R, int P, int NS>
static int sweep_blu(const char* name, double tol) {
  constexpr int M = WAVE * P;
  int bad = 0;
  // (N, lo, Np): the reference's auto-sized 164 with its 82-pixel window, odd N, windows at both ends, the largest N that fits
  const int cases[][3] = {{164, 41, 82}, {49, 13, 23}, {100, 0, 64}, {M - 82 + 1, (M - 82 + 1 - 82) / 2, 82}, {1000, 459, 82},
                          {M - 127, M - 127 - 128, 128}, {M / 2 + 1, 3, 64 * NS < M / 2 ? 64 * NS : M / 2}, {333, 100, 129}, {97, 96, 1}};
  for (auto& c : cases) {
    const int N = c[0], lo = c[1], Np = c[2];
    if (Np > 64 * NS || Np < 1 || lo < 0 || lo + Np > N || N + Np - 1 > M || N < 2) continue;
    const double err = run_blu_case<R, P, NS>(N, lo, Np, 99u + N);
    const bool ok = err <= tol;
    std::printf("%s chirp-z P=%d NS=%d N=%d lo=%d Np=%d relerr=%.3e %s\n", name, P, NS, N, lo, Np, err, ok ? "ok" : "FAIL");
    bad += !ok;
  }
  return bad;
}

// 50-lane mixed-radix row of fmc_mrfft.h: N = 50 P against the naive DFT with numpy's fftshift on both sides (N even).
template <class R, int P, int NS>
static double run_mr_case(int lo, int Np, unsigned seed) {
  constexpr int N = MR_LN * P;
  using G = MrGeom<R, P>;
  using E = typename Xch<R>::E;
  std::mt19937_64 gen(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> inr(N), ini(N);
  for (int k = 0; k < N; ++k) { inr[k] = nd(gen); ini[k] = nd(gen); }
  const int omS = NS * WAVE;
  std::vector<cpx<R>> tw1((size_t)P * WAVE), om((size_t)G::L0 * omS);
  build_tw1_mr<R>(tw1.data(), P, cs_turns);
  build_om_mr<R>(om.data(), omS, P, lo, Np, cs_turns);
  std::vector<E> xbuf(G::XELEMS);
  static HostExec<R, P, NS> ex;
  for (int l = 0; l < WAVE; ++l)
    for (int j = 0; j < P; ++j) {
      const int k = l + MR_LN * j;
      const double sg = (k & 1) ? -1.0 : 1.0;
      // idle lanes carry garbage on the GPU (nothing reads their exchange-1 column): poison them here
      ex.regs[l].v[j] = l < MR_LN ? mk<R>((R)(sg * inr[k]), (R)(sg * ini[k])) : mk<R>((R)1e3, (R)-1e3);
    }
  pruned_row_fft_mr<R, P, NS>(ex, xbuf.data(), tw1.data(), om.data(), omS, lo, Np, mr_osign(N));
  double worst = 0.0, scale = 0.0;
  const int h = N / 2;
  for (int oi = 0; oi < Np; ++oi) {
    const int p = lo + oi;
    long double sr = 0, si = 0;
    for (int k = 0; k < N; ++k) {
      const long long e = (((long long)(p - h) * (k + h)) % N + N) % N;
      const long double a = -2.0L * M_PIl * (long double)e / N;
      const long double c = cosl(a), s = sinl(a);
      sr += inr[k] * c - ini[k] * s;
      si += inr[k] * s + ini[k] * c;
    }
    const int l = oi % WAVE, s = oi / WAVE;
    const double gr = ex.regs[l].xr[s], gi = ex.regs[l].xi[s];
    worst = std::fmax(worst, std::fmax(std::fabs(gr - (double)sr), std::fabs(gi - (double)si)));
    scale = std::fmax(scale, std::fmax(std::fabs((double)sr), std::fabs((double)si)));
  }
  return worst / scale;
}

template <class R, int P, int NS>
static int sweep_mr(const char* name, double tol) {
  constexpr int N = MR_LN * P;
  int bad = 0;
  const int cases[][2] = {{(N - 82) / 2, 82}, {0, 64 * NS < N ? 64 * NS : N}, {N - 5, 5}, {(N - 23) / 2, 23},
                          {(N - 1) / 2, 1}, {7, 64 * NS - 3 < N - 7 ? 64 * NS - 3 : N - 7}, {(N - 128) / 2, NS >= 2 ? 128 : 64},
                          {N - 64 * NS > 0 ? N - 64 * NS : 0, 64 * NS < N ? 64 * NS : N}};
  for (auto& c : cases) {
    if (c[1] > 64 * NS || c[0] < 0 || c[0] + c[1] > N || c[1] < 1) continue;
    const double err = run_mr_case<R, P, NS>(c[0], c[1], 4321u + c[0]);
    const bool ok = err <= tol;
    std::printf("%s 50-lane P=%d NS=%d N=%d lo=%d Np=%d relerr=%.3e %s\n", name, P, NS, N, c[0], c[1], err, ok ? "ok" : "FAIL");
    bad += !ok;
  }
  return bad;
}


// Packed rows of fmc_wavefft.h (N = 256: four rows per wave, N = 512: two): every row of the wave against the naive DFT.
template <class R, int L0, int NSL, int B0M>
static double run_pk_case(int lo, int Np, unsigned seed) {
  constexpr int L = pk_lanes(L0), N = 16 * L, G = WAVE / L;
  using E = typename Xch<R>::E;
  std::mt19937_64 gen(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> inr(G * N), ini(G * N);
  for (auto& v : inr) v = nd(gen);
  for (auto& v : ini) v = nd(gen);
  const int omS = (Np + 7) & ~7;
  std::vector<cpx<R>> tw1((size_t)16 * L), om((size_t)2 * omS);
  build_tw1_pk<R>(tw1.data(), L, cs_turns);
  build_om_pk<R>(om.data(), omS, L, lo, Np, cs_turns);
  std::vector<E> xbuf(D16_XELEMS);
  static HostExec<R, 16, NSL> ex;
  for (int l = 0; l < WAVE; ++l)
    for (int j = 0; j < 16; ++j) {
      const int g = l / L, k = l % L + L * j;
      const double sg = (k & 1) ? -1.0 : 1.0;
      ex.regs[l].v[j] = mk<R>((R)(sg * inr[g * N + k]), (R)(sg * ini[g * N + k]));
    }
  packed_row_fft<R, L0, NSL, B0M>(ex, xbuf.data(), tw1.data(), om.data(), omS, lo, Np);
  std::vector<double> gr((size_t)G * Np, 1e300), gi((size_t)G * Np, 1e300);
  for (int l = 0; l < WAVE; ++l)
    packed_outputs<R, L0, NSL, B0M>(l, ex.regs[l], lo, Np, [&](int oi, R re, R im) { gr[(l / L) * Np + oi] = re; gi[(l / L) * Np + oi] = im; });
  double worst = 0.0, scale = 0.0;
  const int h = N / 2;
  for (int g = 0; g < G; ++g)
    for (int oi = 0; oi < Np; ++oi) {
      const int p = lo + oi;
      long double sr = 0, si = 0;
      for (int k = 0; k < N; ++k) {
        const long long e = (((long long)(p - h) * (k + h)) % N + N) % N;
        const long double a = -2.0L * M_PIl * (long double)e / N;
        const long double c = cosl(a), s = sinl(a);
        sr += inr[g * N + k] * c - ini[g * N + k] * s;
        si += inr[g * N + k] * s + ini[g * N + k] * c;
      }
      worst = std::fmax(worst, std::fmax(std::fabs(gr[g * Np + oi] - (double)sr), std::fabs(gi[g * Np + oi] - (double)si)));
      scale = std::fmax(scale, std::fmax(std::fabs((double)sr), std::fabs((double)si)));
    }
  return worst / scale;
}
template <class R, int L0>
static int sweep_pk(const char* name, double tol) {
  constexpr int L = pk_lanes(L0), N = 16 * L, WALL = N < 256 ? N : 256, NSC = 96 / L, NSA = WALL / L;
  int bad = 0;
  auto report = [&](const char* what, int lo, int Np, double err) {
    const bool ok = err <= tol;
    std::printf("%s packed N=%d %s lo=%d Np=%d relerr=%.3e %s\n", name, N, what, lo, Np, err, ok ? "ok" : "FAIL");
    bad += !ok;
  };
  for (int Np : {82, 96, 23, 1, 64, 95})       // centred windows: six planes
    for (int lo : {(N - Np) / 2, (N - Np) / 2 + (Np < 90 ? 3 : 0)})
      report("centred planes", lo, Np, run_pk_case<R, L0, NSC, pk_centre_mask<L0>()>(lo, Np, 4321u + Np + lo));
  const int cases[][2] = {{(N - 82) / 2, 82}, {0, WALL}, {N - 5, 5}, {5, WALL - 6}, {(N - 128) / 2, 128}, {N - WALL, WALL}, {3, 97}, {N / 2 - 60, 121}};
  for (auto& c : cases) report("all planes", c[0], c[1], run_pk_case<R, L0, NSA, pk_all_mask<L0>()>(c[0], c[1], 99u + c[0] + c[1]));
  return bad;
}

// Packed sub-rows (fmc_wavefft.h: pks_accumulate): G rows of N = S * M points per wave, S passes, against the naive DFT.
template <class R, int L0, int S, int NPL = 6>
static double run_pks_case(int Np, int shift, unsigned seed) {
  constexpr int L = pk_lanes(L0), M = 16 * L, N = S * M, G = WAVE / L, FIRST = pks_first_plane(L0, S, NPL), SPAN = pks_span(NPL);
  using E = typename Xch<R>::E;
  std::mt19937_64 gen(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> inr(G * N), ini(G * N);
  for (auto& v : inr) v = nd(gen);
  for (auto& v : ini) v = nd(gen);
  const int lo = (N - Np) / 2 + shift;
  std::vector<cpx<R>> tw1((size_t)16 * L), pcw((size_t)S * SPAN);
  build_tw1_pk<R>(tw1.data(), L, cs_turns);
  build_pcw<R>(pcw.data(), N, S, cs_turns, NPL);
  std::vector<E> xbuf(D16_XELEMS);
  static HostExec<R, 16, pks_nm<L0, NPL>()> ex;
  pks_clear<R, L0, NPL>(ex);
  for (int sp = 0; sp < S; ++sp) {
    for (int l = 0; l < WAVE; ++l)
      for (int j = 0; j < 16; ++j) {
        const int g = l / L, k = sp + S * (l % L + L * j);
        const double sg = (k & 1) ? -1.0 : 1.0;        // the input-side fftshift sign, folded into the colouring table on the device
        ex.regs[l].v[j] = mk<R>((R)(sg * inr[g * N + k]), (R)(sg * ini[g * N + k]));
      }
    packed_row_fft<R, L0, pks_nm<L0, NPL>(), pks_plane_mask(L0, S, NPL)>(ex, xbuf.data(), tw1.data(), (const cpx<R>*)nullptr, 0, 0, Np);
    pks_accumulate<R, L0, FIRST, NPL>(ex, pcw.data() + sp * SPAN);
  }
  std::vector<double> gr((size_t)G * Np, 1e300), gi((size_t)G * Np, 1e300);
  for (int l = 0; l < WAVE; ++l)
    pks_outputs<R, L0, NPL>(l, ex.regs[l], N, lo, Np, [&](int oi, R re, R im) { gr[(l / L) * Np + oi] = re; gi[(l / L) * Np + oi] = im; });
  double worst = 0.0, scale = 0.0;
  const int h = N / 2;
  for (int g = 0; g < G; ++g)
    for (int oi = 0; oi < Np; ++oi) {
      const int p = lo + oi;
      long double sr = 0, si = 0;
      for (int k = 0; k < N; ++k) {
        const long long e = (((long long)(p - h) * (k + h)) % N + N) % N;
        const long double a = -2.0L * M_PIl * (long double)e / N;
        const long double c = cosl(a), sn = sinl(a);
        sr += inr[g * N + k] * c - ini[g * N + k] * sn;
        si += inr[g * N + k] * sn + ini[g * N + k] * c;
      }
      worst = std::fmax(worst, std::fmax(std::fabs(gr[g * Np + oi] - (double)sr), std::fabs(gi[g * Np + oi] - (double)si)));
      scale = std::fmax(scale, std::fmax(std::fabs((double)sr), std::fabs((double)si)));
    }
  return worst / scale;
}
// all sixteen planes of 256-point sub-rows: centred windows of up to 256 pixels
template <class R, int S>
static int sweep_pks16(const char* name, double tol) {
  int bad = 0;
  for (int Np : {256, 129, 200, 82, 1})
    for (int shift : {0, Np < 250 ? 3 : 0, Np < 200 ? -27 : 0}) {
      const double err = run_pks_case<R, 1, S, 16>(Np, shift, 1616u + Np + shift + S);
      const bool ok = err <= tol;
      std::printf("%s packed sub-rows, sixteen planes N=%d (S=%d) shift=%d Np=%d relerr=%.3e %s\n", name, S * 256, S, shift, Np, err, ok ? "ok" : "FAIL");
      bad += !ok;
    }
  return bad;
}
// eight planes: centred windows of up to 128 pixels
template <class R, int L0, int S>
static int sweep_pks8(const char* name, double tol) {
  int bad = 0;
  for (int Np : {128, 97, 110, 82, 1})
    for (int shift : {0, Np < 120 ? 4 : 0, Np < 100 ? -13 : 0}) {
      const double err = run_pks_case<R, L0, S, 8>(Np, shift, 555u + Np + shift + S);
      const bool ok = err <= tol;
      std::printf("%s packed sub-rows, eight planes N=%d (S=%d) shift=%d Np=%d relerr=%.3e %s\n", name, S * 16 * pk_lanes(L0), S, shift, Np, err, ok ? "ok" : "FAIL");
      bad += !ok;
    }
  return bad;
}
template <class R, int L0, int S>
static int sweep_pks(const char* name, double tol) {
  int bad = 0;
  for (int Np : {82, 96, 23, 1, 64, 95, 40})
    for (int shift : {0, Np < 90 ? 3 : 0, Np < 80 ? -5 : 0}) {
      const double err = run_pks_case<R, L0, S>(Np, shift, 777u + Np + shift + S);
      const bool ok = err <= tol;
      std::printf("%s packed sub-rows N=%d (S=%d) shift=%d Np=%d relerr=%.3e %s\n", name, S * 16 * pk_lanes(L0), S, shift, Np, err, ok ? "ok" : "FAIL");
      bad += !ok;
    }
  return bad;
}

// Packed sub-rows of SIXTY-FOUR points (pks64_pass): eight rows of N = S * 64 points per wave, eight lanes each, S passes, against the naive DFT.
template <class R, int S, int NPL = 6>
static double run_pks64_case(int Np, int shift, unsigned seed) {
  constexpr int M = 64, N = S * M, G = 8;
  using E = typename Xch<R>::E;
  std::mt19937_64 gen(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> inr(G * N), ini(G * N);
  for (auto& v : inr) v = nd(gen);
  for (auto& v : ini) v = nd(gen);
  const int lo = (N - Np) / 2 + shift;
  std::vector<cpx<R>> tw(64), pcw((size_t)S * pks_span(NPL));
  build_tw64<R>(tw.data(), cs_turns);
  build_pcw<R>(pcw.data(), N, S, cs_turns, NPL);
  std::vector<E> xbuf(D16_XELEMS);
  static HostExec<R, 16, 2> ex;
  pks_clear<R, -1, NPL>(ex);
  for (int sp = 0; sp < S; ++sp) {
    for (int l = 0; l < WAVE; ++l)
      for (int j = 0; j < 8; ++j) {
        const int g = l / 8, k = sp + S * (l % 8 + 8 * j);
        const double sg = (k & 1) ? -1.0 : 1.0;
        ex.regs[l].v[j] = mk<R>((R)(sg * inr[g * N + k]), (R)(sg * ini[g * N + k]));
      }
    pks64_pass<R, NPL>(ex, xbuf.data(), tw.data(), pcw.data() + sp * pks_span(NPL));
  }
  std::vector<double> gr((size_t)G * Np, 1e300), gi((size_t)G * Np, 1e300);
  for (int l = 0; l < WAVE; ++l)
    pks_outputs<R, -1, NPL>(l, ex.regs[l], N, lo, Np, [&](int oi, R re, R im) { gr[(l / 8) * Np + oi] = re; gi[(l / 8) * Np + oi] = im; });
  double worst = 0.0, scale = 0.0;
  const int h = N / 2;
  for (int g = 0; g < G; ++g)
    for (int oi = 0; oi < Np; ++oi) {
      const int p = lo + oi;
      long double sr = 0, si = 0;
      for (int k = 0; k < N; ++k) {
        const long long e = (((long long)(p - h) * (k + h)) % N + N) % N;
        const long double a = -2.0L * M_PIl * (long double)e / N;
        const long double c = cosl(a), sn = sinl(a);
        sr += inr[g * N + k] * c - ini[g * N + k] * sn;
        si += inr[g * N + k] * sn + ini[g * N + k] * c;
      }
      worst = std::fmax(worst, std::fmax(std::fabs(gr[g * Np + oi] - (double)sr), std::fabs(gi[g * Np + oi] - (double)si)));
      scale = std::fmax(scale, std::fmax(std::fabs((double)sr), std::fabs((double)si)));
    }
  return worst / scale;
}
template <class R, int S>
static int sweep_pks64_8(const char* name, double tol) {
  int bad = 0;
  for (int Np : {128, 97, 110, 82, 1})
    for (int shift : {0, Np < 120 ? 4 : 0, Np < 100 ? -13 : 0}) {
      const double err = run_pks64_case<R, S, 8>(Np, shift, 8181u + Np + shift + S);
      const bool ok = err <= tol;
      std::printf("%s packed sub-rows of 64, eight planes N=%d (S=%d) shift=%d Np=%d relerr=%.3e %s\n", name, S * 64, S, shift, Np, err, ok ? "ok" : "FAIL");
      bad += !ok;
    }
  return bad;
}
template <class R, int S>
static int sweep_pks64(const char* name, double tol) {
  int bad = 0;
  for (int Np : {82, 96, 23, 1, 64, 95, 40})
    for (int shift : {0, Np < 90 ? 3 : 0, Np < 80 ? -5 : 0}) {
      const double err = run_pks64_case<R, S>(Np, shift, 4242u + Np + shift + S);
      const bool ok = err <= tol;
      std::printf("%s packed sub-rows of 64 N=%d (S=%d) shift=%d Np=%d relerr=%.3e %s\n", name, S * 64, S, shift, Np, err, ok ? "ok" : "FAIL");
      bad += !ok;
    }
  return bad;
}

int main() {
  int bad = 0;
  bad += sweep_pks64<double, 9>("f64", 1e-13);
  bad += sweep_pks64<double, 3>("f64", 1e-13);
  bad += sweep_pks64<double, 5>("f64", 1e-13);
  bad += sweep_pks64<double, 7>("f64", 1e-13);
  bad += sweep_pks64<float, 9>("f32", 2e-5);
  bad += sweep_pks<double, 1, 3>("f64", 1e-13);
  bad += sweep_pks<double, 1, 5>("f64", 1e-13);
  bad += sweep_pks<double, 1, 6>("f64", 1e-13);
  bad += sweep_pks<double, 1, 7>("f64", 1e-13);
  bad += sweep_pks<double, 0, 3>("f64", 1e-13);
  bad += sweep_pks<double, 0, 5>("f64", 1e-13);
  bad += sweep_pks<double, 0, 7>("f64", 1e-13);
  bad += sweep_pks<double, 0, 9>("f64", 1e-13);
  // the counts the run-time kernels take (fmc_core.h: pks_rt): the arithmetic knows the count's parity only
  bad += sweep_pks16<double, 5>("f64", 1e-13);
  bad += sweep_pks16<double, 6>("f64", 1e-13);
  bad += sweep_pks8<double, 1, 5>("f64", 1e-13);
  bad += sweep_pks8<double, 1, 6>("f64", 1e-13);
  bad += sweep_pks8<double, 0, 7>("f64", 1e-13);
  bad += sweep_pks8<double, 0, 15>("f64", 1e-13);
  bad += sweep_pks64_8<double, 9>("f64", 1e-13);
  bad += sweep_pks64_8<double, 21>("f64", 1e-13);
  bad += sweep_pks64<double, 11>("f64", 1e-13);
  bad += sweep_pks64<double, 33>("f64", 1e-13);
  bad += sweep_pks<double, 0, 31>("f64", 1e-13);
  bad += sweep_pks<double, 1, 11>("f64", 1e-13);
  bad += sweep_pks<double, 0, 15>("f64", 1e-13);
  bad += sweep_pks<double, 0, 27>("f64", 1e-13);
  bad += sweep_pks<double, 1, 9>("f64", 1e-13);
  bad += sweep_pks<double, 1, 10>("f64", 1e-13);
  bad += sweep_pks<double, 1, 15>("f64", 1e-13);
  bad += sweep_pks<float, 1, 5>("f32", 2e-5);
  bad += sweep_pks<float, 0, 7>("f32", 2e-5);
  bad += sweep_pk<double, 0>("f64", 1e-13);
  bad += sweep_pk<float, 0>("f32", 2e-5);
  bad += sweep_pk<double, 1>("f64", 1e-13);
  bad += sweep_pk<double, 2>("f64", 1e-13);
  bad += sweep_pk<float, 1>("f32", 2e-5);
  bad += sweep_pk<float, 2>("f32", 2e-5);
  bad += sweep_mr<double, 2, 2>("f64", 1e-13);
  bad += sweep_mr<double, 3, 2>("f64", 1e-13);
  bad += sweep_mr<double, 4, 2>("f64", 1e-13);
  bad += sweep_mr<double, 5, 2>("f64", 1e-13);
  bad += sweep_mr<double, 6, 2>("f64", 1e-13);
  bad += sweep_mr<double, 7, 2>("f64", 1e-13);
  bad += sweep_mr<double, 8, 2>("f64", 1e-13);
  bad += sweep_mr<double, 9, 2>("f64", 1e-13);
  bad += sweep_mr<double, 10, 2>("f64", 1e-13);
  bad += sweep_mr<double, 10, 4>("f64", 1e-13);
  bad += sweep_mr<double, 12, 2>("f64", 1e-13);
  bad += sweep_mr<double, 14, 2>("f64", 1e-13);
  bad += sweep_mr<double, 16, 2>("f64", 1e-13);
  bad += sweep_mr<double, 18, 2>("f64", 1e-13);
  bad += sweep_mr<double, 20, 2>("f64", 1e-13);
  bad += sweep_mr<double, 20, 4>("f64", 1e-13);
  bad += sweep_mr<double, 24, 2>("f64", 1e-13);
  bad += sweep_mr<double, 28, 2>("f64", 1e-13);
  bad += sweep_mr<double, 32, 2>("f64", 1e-13);
  bad += sweep_mr<float, 5, 2>("f32", 2e-5);
  bad += sweep_mr<float, 20, 2>("f32", 2e-5);
  bad += sweep_mr<float, 24, 4>("f32", 2e-5);
  bad += sweep_pbz<double, 6>("f64", 1e-12);
  bad += sweep_pbz<double, 8>("f64", 1e-12);
  bad += sweep_blu<double, 4, 2>("f64", 1e-12);
  bad += sweep_blu<double, 8, 2>("f64", 1e-12);
  bad += sweep_blu<double, 8, 4>("f64", 1e-12);
  bad += sweep_blu<double, 16, 2>("f64", 1e-12);
  bad += sweep_blu<double, 16, 4>("f64", 1e-12);
  bad += sweep_blu<double, 12, 2>("f64", 1e-12);
  bad += sweep_blu<double, 20, 2>("f64", 1e-12);
  bad += sweep_blu<double, 28, 2>("f64", 1e-12);
  bad += sweep_blu<double, 24, 2>("f64", 1e-12);
  bad += sweep_blu<double, 32, 2>("f64", 1e-12);
  bad += sweep_blu_blocked<double, 2>("f64", 1e-12);
  bad += sweep_blu_blocked<double, 4>("f64", 1e-12);
  bad += sweep_blu_blocked<float, 2>("f32", 2e-4);
  bad += sweep_blu<float, 4, 2>("f32", 1e-4);
  bad += sweep_blu<float, 16, 2>("f32", 1e-4);
  bad += sweep_blu<float, 24, 4>("f32", 1e-4);
  bad += sweep<double, 2, 2>("f64", 1e-13);
  bad += sweep<double, 4, 2>("f64", 1e-13);
  bad += sweep<double, 4, 4>("f64", 1e-13);
  bad += sweep<float, 2, 2>("f32", 2e-5);
  bad += sweep<float, 4, 4>("f32", 2e-5);
  bad += sweep<double, 3, 2>("f64", 1e-13);
  bad += sweep<double, 5, 2>("f64", 1e-13);
  bad += sweep<double, 6, 2>("f64", 1e-13);
  bad += sweep<double, 7, 2>("f64", 1e-13);
  bad += sweep<double, 9, 2>("f64", 1e-13);
  bad += sweep<double, 14, 2>("f64", 1e-13);
  bad += sweep<double, 18, 2>("f64", 1e-13);
  bad += sweep<double, 28, 2>("f64", 1e-13);
  bad += sweep<float, 18, 2>("f32", 2e-5);
  bad += sweep<double, 10, 2>("f64", 1e-13);
  bad += sweep<double, 12, 2>("f64", 1e-13);
  bad += sweep<double, 20, 2>("f64", 1e-13);
  bad += sweep<double, 24, 2>("f64", 1e-13);
  bad += sweep<float, 12, 2>("f32", 2e-5);
  bad += sweep<float, 24, 2>("f32", 2e-5);
  bad += sweep<double, 8, 2>("f64", 1e-13);
  bad += sweep<double, 16, 2>("f64", 1e-13);
  g_dense_r16 = 1;
  bad += sweep<double, 16, 2>("f64 dense 16x4", 1e-13);
  bad += sweep<float, 16, 2>("f32 dense 16x4", 2e-5);
  g_dense_r16 = 2;      // the plane set of a centred window: the centred cases of the sweep
  for (int Np : {82, 96, 23, 1, 64})
    for (int lo : {(1024 - Np) / 2, (1024 - Np) / 2 + (Np < 90 ? 3 : 0)}) {
      const double err = run_case<double, 16, 2>(lo, Np, 777u + Np);
      std::printf("f64 dense 16x4 centred planes lo=%d Np=%d relerr=%.3e %s\n", lo, Np, err, err <= 1e-13 ? "ok" : "FAIL");
      bad += !(err <= 1e-13);
    }
  g_dense_r16 = 0;
  bad += sweep<double, 32, 2>("f64", 1e-13);
  bad += sweep<double, 8, 8>("f64", 1e-13);
  bad += sweep<double, 16, 16>("f64", 1e-13);
  bad += sweep<float, 8, 2>("f32", 2e-5);
  bad += sweep<float, 16, 2>("f32", 2e-5);
  bad += sweep<float, 32, 2>("f32", 2e-5);
  bad += sweep<float, 16, 16>("f32", 2e-5);
  std::printf("%s\n", bad ? "EMU FAILED" : "EMU OK");
  return bad ? 1 : 0;
}
