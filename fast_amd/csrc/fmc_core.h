// fmc_core.h -- arithmetic shared by the gfx950 kernels and their host-side lane emulation.
//
// Everything here is __host__ __device__ so that tests/emu (a host program that runs the
// per-lane phases of the wave pipeline in a loop over 64 lanes) executes the same index
// arithmetic as the GPU.  Hardware transcendental intrinsics are confined to fmc_kernels.hip.
#pragma once
#include <stdint.h>
#include <utility>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FMC_HD __host__ __device__ __forceinline__
#else
#define FMC_HD inline
#endif

namespace fmc {

constexpr int WAVE = 64;

template <class R>
struct cpx {
  R x, y;
};

template <class R> FMC_HD cpx<R> mk(R x, R y) { cpx<R> r; r.x = x; r.y = y; return r; }
template <class R> FMC_HD cpx<R> operator+(cpx<R> a, cpx<R> b) { return mk<R>(a.x + b.x, a.y + b.y); }
template <class R> FMC_HD cpx<R> operator-(cpx<R> a, cpx<R> b) { return mk<R>(a.x - b.x, a.y - b.y); }
template <class R> FMC_HD cpx<R> cmul(cpx<R> a, cpx<R> b) { return mk<R>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <class R> FMC_HD cpx<R> cfma(cpx<R> a, cpx<R> b, cpx<R> c) {  // a*b + c
  return mk<R>(c.x + a.x * b.x - a.y * b.y, c.y + a.x * b.y + a.y * b.x);
}
template <class R> FMC_HD cpx<R> cscale(cpx<R> a, R s) { return mk<R>(a.x * s, a.y * s); }

// cos(2*pi*k/64), k = 0..16, to double precision.
FMC_HD constexpr double cos64_quarter(int k) {
  switch (k) {
    case 0: return 1.0;
    case 1: return 0.9951847266721968862448;
    case 2: return 0.9807852804032304491262;
    case 3: return 0.9569403357322088649358;
    case 4: return 0.9238795325112867561282;
    case 5: return 0.8819212643483550297128;
    case 6: return 0.8314696123025452370788;
    case 7: return 0.7730104533627369608109;
    case 8: return 0.7071067811865475244008;
    case 9: return 0.6343932841636454982152;
    case 10: return 0.5555702330196022247428;
    case 11: return 0.4713967368259976485564;
    case 12: return 0.3826834323650897717285;
    case 13: return 0.2902846772544623676362;
    case 14: return 0.1950903220161282678483;
    case 15: return 0.0980171403295606019942;
    default: return 0.0;
  }
}
// cos / sin of 2*pi*k/64 for any integer k (symmetry of the quarter table).
FMC_HD constexpr double cos64(int k) {
  k &= 63;
  if (k > 32) k = 64 - k;           // cos(2pi - t) = cos t
  return k <= 16 ? cos64_quarter(k) : -cos64_quarter(32 - k);
}
FMC_HD constexpr double sin64(int k) { return cos64(k - 16); }

// Multiply d by w_P^k = exp(-2*pi*i*k/P), P | 64, k a value the optimiser sees as constant
// after unrolling: the trivial cases cost no multiplies.
template <class R>
FMC_HD cpx<R> mul_tw(cpx<R> d, int k, int P) {
  const int k64 = (k * (64 / P)) & 63;
  if (k64 == 0) return d;
  if (k64 == 16) return mk<R>(d.y, -d.x);    // * (-i)
  if (k64 == 32) return mk<R>(-d.x, -d.y);   // * (-1)
  if (k64 == 48) return mk<R>(-d.y, d.x);    // * (+i)
  const R h = (R)0.7071067811865475244008;
  if (k64 == 8) return mk<R>((d.x + d.y) * h, (d.y - d.x) * h);     // (1 - i)/sqrt2
  if (k64 == 24) return mk<R>((d.y - d.x) * h, -(d.x + d.y) * h);   // (-1 - i)/sqrt2
  if (k64 == 40) return mk<R>(-(d.x + d.y) * h, (d.x - d.y) * h);   // (-1 + i)/sqrt2
  if (k64 == 56) return mk<R>((d.x - d.y) * h, (d.x + d.y) * h);    // (1 + i)/sqrt2
  const R c = (R)cos64(k64), s = (R)(-sin64(k64));                  // w = c + i s
  return mk<R>(d.x * c - d.y * s, d.x * s + d.y * c);
}

FMC_HD constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }
FMC_HD constexpr int brev(int a, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((a >> i) & 1) << (bits - 1 - i);
  return r;
}

// f(integral_constant<int, 0>), ..., f(integral_constant<int, N-1>): a loop whose index is a constant expression
template <class F, int... Is>
FMC_HD void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
FMC_HD void static_for(F&& f) { static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// In-register forward DFT of size P (power of two <= 64), decimation in frequency.
// On return v[brev(a)] = sum_j v_in[j] * exp(-2*pi*i*j*a/P).
//
// FMA form (Linzer & Feig; the form FFTW's FMA codelets take): the real factor of a twiddle is not multiplied in where the
// twiddle is applied but carried as a PENDING SCALE of that element and folded into the next butterfly that reads it,
//     u + w  ->  fma(r, w, u),   r = scale(w) / scale(u)  (a compile-time constant; r = 1: a plain add),
// so that   w_8-type twiddles (1 -+ i) / sqrt 2 cost two adds (pending scale 1 / sqrt 2) instead of two adds and two multiplies,
//           general twiddles c - i s cost two FMAs, c (d.x + t d.y, d.y - t d.x) with t = s / c (or the same with s factored
//           out when |s| > |c|), instead of two multiplies and two FMAs.
// Every pending scale is 1 again after the last stage (the even element of each final pair never carries one; checked at
// compile time).  A radix-16 butterfly costs 148 float64 instructions instead of 168, and no result is less accurate: each
// output sees the same or fewer roundings.
FMC_HD constexpr double dif_abs(double x) { return x < 0 ? -x : x; }
FMC_HD constexpr double dif_tw_scale(int k64) {      // real factor of w_64^{k64} that is left pending
  k64 &= 63;
  if ((k64 & 15) == 0) return 1.0;
  if ((k64 & 7) == 0) return 0.7071067811865475244008;
  const double c = cos64(k64), s = sin64(k64);
  return dif_abs(c) >= dif_abs(s) ? c : s;
}
template <int P>
struct DifPlan {
  static constexpr int L = ilog2(P);
  double sc[L + 1][P];      // sc[t][idx]: pending scale of element idx before stage t (sc[L]: after the last stage)
  constexpr DifPlan() : sc{} {
    for (int i = 0; i < P; ++i) sc[0][i] = 1.0;
    for (int t = 0; t < L; ++t) {
      const int S = P >> (t + 1);
      for (int blk = 0; blk < P; blk += 2 * S)
        for (int i = 0; i < S; ++i) {
          const double su = sc[t][blk + i];
          sc[t + 1][blk + i] = su;
          sc[t + 1][blk + i + S] = su * dif_tw_scale(i * (P / (2 * S)) * (64 / P));
        }
    }
  }
  constexpr bool unit_outputs() const {
    for (int i = 0; i < P; ++i)
      if (sc[L][i] != 1.0) return false;
    return true;
  }
};

// d * w_64^{k64} WITHOUT the real factor dif_tw_scale(k64)
template <int K64, class R>
FMC_HD cpx<R> mul_tw_unscaled(cpx<R> d) {
  constexpr int k64 = K64 & 63;
  if constexpr (k64 == 0) return d;
  else if constexpr (k64 == 16) return mk<R>(d.y, -d.x);
  else if constexpr (k64 == 32) return mk<R>(-d.x, -d.y);
  else if constexpr (k64 == 48) return mk<R>(-d.y, d.x);
  else if constexpr (k64 == 8) return mk<R>(d.x + d.y, d.y - d.x);        // (1 - i)
  else if constexpr (k64 == 24) return mk<R>(d.y - d.x, -d.x - d.y);      // (-1 - i)
  else if constexpr (k64 == 40) return mk<R>(-d.x - d.y, d.x - d.y);      // (-1 + i)
  else if constexpr (k64 == 56) return mk<R>(d.x - d.y, d.x + d.y);       // (1 + i)
  else {
    constexpr double c = cos64(k64), s = sin64(k64);                      // w = c - i s:  d w = (d.x c + d.y s, d.y c - d.x s)
    if constexpr (dif_abs(c) >= dif_abs(s)) {
      constexpr R t = (R)(s / c);
      return mk<R>(d.x + t * d.y, d.y - t * d.x);                          // c (...)
    } else {
      constexpr R t = (R)(c / s);
      return mk<R>(t * d.x + d.y, t * d.y - d.x);                          // s (...)
    }
  }
}

#ifndef FMC_DIF_FMA
#define FMC_DIF_FMA 1
#endif
template <int P, class R>
FMC_HD void fft_dif(cpx<R> (&v)[P]) {
#if FMC_DIF_FMA
  if constexpr (P >= 4 && 64 % P == 0) {
    constexpr DifPlan<P> plan{};
    static_assert(plan.unit_outputs(), "a pending scale survives the last stage");
    static_for<DifPlan<P>::L>([&](auto T) {
      constexpr int t = decltype(T)::value;
      constexpr int S = P >> (t + 1);
      static_for<P / 2>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        constexpr int blk = (q / S) * 2 * S, i = q % S;
        constexpr double ratio = plan.sc[t][blk + i + S] / plan.sc[t][blk + i];
        const cpx<R> u = v[blk + i], w = v[blk + i + S];
        cpx<R> d;
        if constexpr (ratio == 1.0) {
          v[blk + i] = u + w;
          d = u - w;
        } else {
          constexpr R r = (R)ratio;
          v[blk + i] = mk<R>(u.x + r * w.x, u.y + r * w.y);
          d = mk<R>(u.x - r * w.x, u.y - r * w.y);
        }
        v[blk + i + S] = mul_tw_unscaled<i * (P / (2 * S)) * (64 / P), R>(d);
      });
    });
    return;
  }
#endif
#pragma unroll
  for (int S = P / 2; S >= 1; S >>= 1) {
#pragma unroll
    for (int blk = 0; blk < P; blk += 2 * S) {
#pragma unroll
      for (int i = 0; i < S; ++i) {
        const cpx<R> u = v[blk + i], w = v[blk + i + S];
        v[blk + i] = u + w;
        v[blk + i + S] = mul_tw<R>(u - w, i * (P / (2 * S)), P);
      }
    }
  }
}

// ---------------------------------------------------------------- in-register DFT of any supported size
// cos / sin of 2*pi*num/den evaluated by the compiler (constant arguments after unrolling):
// reduction to the first octant, then 12-term Taylor series (error < 1e-17).
FMC_HD constexpr double taylor_sin(double x) {
  double term = x, sum = x;
  for (int k = 1; k < 12; ++k) { term *= -x * x / ((2 * k) * (2 * k + 1)); sum += term; }
  return sum;
}
FMC_HD constexpr double taylor_cos(double x) {
  double term = 1.0, sum = 1.0;
  for (int k = 1; k < 12; ++k) { term *= -x * x / ((2 * k - 1) * (2 * k)); sum += term; }
  return sum;
}
FMC_HD constexpr double cos_frac(int num, int den) {   // cos(2 pi num / den)
  num %= den;
  if (num < 0) num += den;
  if (2 * num > den) num = den - num;                                   // cos(2pi - t) = cos t, t now in [0, pi]
  double sign = 1.0;
  long long n = num, d = den;
  if (4 * n > d) { n = d - 2 * n; d = 2 * d; sign = -1.0; }             // cos(t) = -cos(pi - t), now in [0, pi/2]
  const double t = 6.283185307179586476925286766559 * (double)n / (double)d;
  return sign * ((8 * n > d) ? taylor_sin(1.5707963267948966192313216916398 - t) : taylor_cos(t));
}
FMC_HD constexpr double sin_frac(int num, int den) { return cos_frac(4 * num - den, 4 * den); }   // sin t = cos(t - pi/2)

FMC_HD constexpr bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// Natural-order forward DFT of P values held in registers.  Powers of two: radix-2 DIF network;
// P = Q*2^k with Q = 3, 5, 7, 9: one Cooley-Tukey level (Q strided power-of-two sub-transforms,
// twiddle, M radix-Q butterflies with compile-time constants).
template <int P, class R>
FMC_HD void dft_reg(cpx<R> (&v)[P]) {
  if constexpr (is_pow2(P)) {
    fft_dif<P, R>(v);
    cpx<R> t[P];
#pragma unroll
    for (int a = 0; a < P; ++a) t[a] = v[brev(a, ilog2(P))];
#pragma unroll
    for (int a = 0; a < P; ++a) v[a] = t[a];
  } else {
    constexpr int M = P & -P;      // largest power of two dividing P
    constexpr int Q = P / M;       // odd part: one direct radix-Q stage (Q^2 complex MACs per butterfly)
    static_assert(Q == 3 || Q == 5 || Q == 7 || Q == 9, "supported sizes: 2^k times 1, 3, 5, 7 or 9");
    cpx<R> y[Q][M];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      cpx<R> t[M];
#pragma unroll
      for (int m = 0; m < M; ++m) t[m] = v[q + Q * m];
      if constexpr (M > 1) fft_dif<M, R>(t);
#pragma unroll
      for (int a2 = 0; a2 < M; ++a2) y[q][a2] = t[brev(a2, ilog2(M))];
    }
    // loop indices as template constants, so that every twiddle is a compile-time literal
    static_for<M>([&](auto A2) {
      constexpr int a2 = decltype(A2)::value;
      cpx<R> u[Q];
      static_for<Q>([&](auto Qi) {
        constexpr int q = decltype(Qi)::value;
        constexpr int e = (q * a2) % P;
        if constexpr (e == 0) u[q] = y[q][a2];
        else {
          constexpr double wr = cos_frac(e, P), wi = -sin_frac(e, P);
          u[q] = cmul(y[q][a2], mk<R>((R)wr, (R)wi));
        }
      });
      if constexpr (Q == 5) {
        // radix-5 butterfly in 36 real operations instead of the 72 of the direct form (round 4: the 50-lane rows -- 1000 = 50 x 20 --
        // spent a third of their float64 instructions in direct radix-5 butterflies):  t1 = u1 + u4, t2 = u2 + u3, t3 = u1 - u4,
        // t4 = u2 - u3;  a1 = u0 + c1 t1 + c2 t2, a2 = u0 + c2 t1 + c1 t2, b1 = s1 t3 + s2 t4, b2 = s2 t3 - s1 t4;
        // y0 = u0 + t1 + t2, y1 = a1 - i b1, y4 = a1 + i b1, y2 = a2 - i b2, y3 = a2 + i b2
        constexpr R c1 = (R)cos_frac(1, 5), c2 = (R)cos_frac(2, 5), s1 = (R)sin_frac(1, 5), s2 = (R)sin_frac(2, 5);
        const cpx<R> t1 = u[1] + u[4], t2 = u[2] + u[3], t3 = u[1] - u[4], t4 = u[2] - u[3];
        const cpx<R> p1 = mk<R>(u[0].x + c1 * t1.x + c2 * t2.x, u[0].y + c1 * t1.y + c2 * t2.y);
        const cpx<R> p2 = mk<R>(u[0].x + c2 * t1.x + c1 * t2.x, u[0].y + c2 * t1.y + c1 * t2.y);
        const cpx<R> q1 = mk<R>(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
        const cpx<R> q2 = mk<R>(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
        v[a2 + M * 0] = u[0] + (t1 + t2);
        v[a2 + M * 1] = mk<R>(p1.x + q1.y, p1.y - q1.x);
        v[a2 + M * 4] = mk<R>(p1.x - q1.y, p1.y + q1.x);
        v[a2 + M * 2] = mk<R>(p2.x + q2.y, p2.y - q2.x);
        v[a2 + M * 3] = mk<R>(p2.x - q2.y, p2.y + q2.x);
      } else if constexpr (Q == 3) {
        // radix-3: t = u1 + u2, a = u0 - t / 2, b = (sqrt 3 / 2)(u1 - u2);  y0 = u0 + t, y1 = a - i b, y2 = a + i b
        constexpr R s = (R)sin_frac(1, 3);
        const cpx<R> t = u[1] + u[2], d = u[1] - u[2];
        const cpx<R> a = mk<R>(u[0].x - (R)0.5 * t.x, u[0].y - (R)0.5 * t.y), b = mk<R>(s * d.x, s * d.y);
        v[a2 + M * 0] = u[0] + t;
        v[a2 + M * 1] = mk<R>(a.x + b.y, a.y - b.x);
        v[a2 + M * 2] = mk<R>(a.x - b.y, a.y + b.x);
      } else {
        // radix-7 / radix-9 (round 5; the direct form costs Q (Q - 1) complex multiply-adds = 168 / 288 real operations, the rows of
        // 448 / 896 / 1792 and 576 / 1152 spent a third of their float64 instructions there): the same symmetric form as the
        // radix-5 butterfly -- t_k = u_k + u_{Q-k}, d_k = u_k - u_{Q-k} (k = 1 ... H = (Q - 1) / 2); a_j = u0 + sum_k cos(2 pi j k / Q) t_k,
        // b_j = sum_k sin(2 pi j k / Q) d_k;  y0 = u0 + sum_k t_k, y_j = a_j - i b_j, y_{Q-j} = a_j + i b_j: 66 / 104 real operations
        constexpr int H = (Q - 1) / 2;
        cpx<R> t[H], d[H];
        static_for<H>([&](auto K) {
          constexpr int k = decltype(K)::value + 1;
          t[k - 1] = u[k] + u[Q - k];
          d[k - 1] = u[k] - u[Q - k];
        });
        cpx<R> y0 = u[0];
        static_for<H>([&](auto K) { y0 = y0 + t[decltype(K)::value]; });
        v[a2 + M * 0] = y0;
        static_for<H>([&](auto J) {
          constexpr int j = decltype(J)::value + 1;
          cpx<R> a = u[0], b = mk<R>((R)0, (R)0);
          static_for<H>([&](auto K) {
            constexpr int k = decltype(K)::value + 1;
            constexpr R c = (R)cos_frac((j * k) % Q, Q), sn = (R)sin_frac((j * k) % Q, Q);
            a = mk<R>(a.x + c * t[k - 1].x, a.y + c * t[k - 1].y);
            if constexpr (k == 1) b = mk<R>(sn * d[0].x, sn * d[0].y);
            else b = mk<R>(b.x + sn * d[k - 1].x, b.y + sn * d[k - 1].y);
          });
          v[a2 + M * j] = mk<R>(a.x + b.y, a.y - b.x);
          v[a2 + M * (Q - j)] = mk<R>(a.x - b.y, a.y + b.x);
        });
      }
    });
  }
}

// ---------------------------------------------------------------- Philox4x32-10
struct u32x4 {
  uint32_t a, b, c, d;
};

FMC_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c);
template <int ROUNDS>
FMC_HD u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0;
    const uint64_t p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  u32x4 o; o.a = c0; o.b = c1; o.c = c2; o.d = c3;
  return o;
}
FMC_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return philox4x32<10>(c0, c1, c2, c3, k0, k1);
}

// ---------------------------------------------------------------- xoshiro128+ (Blackman & Vigna)
// Short per-(realisation, row, lane) streams: the 128-bit state is one Philox4x32-10 block, then
// each 32-bit output costs ~9 cheap VALU ops instead of a quarter of a Philox block.  The "+"
// scrambler's weak low bits are dropped by the u32 -> float32 conversion in Box-Muller.
FMC_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FMC_NO_BITOP3)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);     // truth table of a ^ b ^ c
#else
  return a ^ b ^ c;
#endif
}

struct xoshiro128p {
  uint32_t s0, s1, s2, s3;
  FMC_HD void seed(u32x4 x) {
    s0 = x.a; s1 = x.b; s2 = x.c; s3 = x.d;
    if ((s0 | s1 | s2 | s3) == 0u) s0 = 1u;
  }
  // the five xors of the reference formulation (s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t) as three 3-input xors
  // of the OLD state words and one 2-input xor: on gfx950 a 3-input xor is one v_bitop3_b32
  FMC_HD void advance() {
    const uint32_t t = s1 << 9;
    const uint32_t n1 = xor3(s1, s2, s0);
    const uint32_t n0 = xor3(s0, s3, s1);
    const uint32_t n2 = xor3(s2, s0, t);
    const uint32_t x3 = s3 ^ s1;
    s0 = n0; s1 = n1; s2 = n2;
    s3 = (x3 << 11) | (x3 >> 21);
  }
  FMC_HD uint32_t next() {
    const uint32_t r = s0 + s3;
    advance();
    return r;
  }
  // Two words from ONE state advance: a = s0 + s3 (the "+" scrambler) and b = s1 + s2.  Over the period the state
  // visits every non-zero 128-bit value once, so (a, b) is jointly equidistributed (each pair has 2^64 preimages);
  // the streams here are 2 P <= 64 steps long, seeded by independent Philox blocks.
  FMC_HD void next2(uint32_t& a, uint32_t& b) {
    a = s0 + s3;
    b = s1 + s2;
    advance();
  }
  // Four words from ONE state advance (the float64 generator, round 5): (a, b) as next2 -- the float32 draw's words -- and
  //   a2 = ((a mod 2^24) 0x9E3779 + s0) | 1,   b2 = (b mod 2^24) 0x85EBCB + s1          (mod 2^32)
  // -- each sum scrambled once more by a 24-bit multiply and combined with one of its terms (one v_mad_u32_u24 each: a 24-bit
  // multiply-add issues faster than the rotate of a "++" scrambler; tools/ubench).  (a, a2 >> 1) <-> (s0 up to one bit, s3) and
  // (b, b2) <-> (s1, s2) are bijections, so the four words are jointly equidistributed over the period like the state itself;
  // a2 and b2 only supply the bits BELOW the 32 / 24 leading ones of the uniform and of the angle (fmc_gen64.h), and a2 comes out
  // odd because the uniform wants it so.  Three plain instructions instead of the second stream's eight + the splice, and no
  // second Philox block per lane and row.
  FMC_HD void next4(uint32_t& a, uint32_t& b, uint32_t& a2, uint32_t& b2) {
    a = s0 + s3;
    b = s1 + s2;
    a2 = ((a & 0x00FFFFFFu) * 0x009E3779u + s0) | 1u;
    b2 = (b & 0x00FFFFFFu) * 0x0085EBCBu + s1;
    advance();
  }
};

// Streams of the device generator (counter word 1).
//   STREAM_SCREEN: counter word 0 = ky*SL + L, SL = 64*spec_split(N), L = kx mod SL; the xoshiro stream seeded by that
//                  block yields, for j = 0, 1, ..., the two words of coefficient (ky, kx = L + SL j).
// Generator layout: a grid of N columns is drawn as 64*spec_split(N) streams per row, stream L = kx mod (64 S)
// yielding coefficient j <-> kx = L + 64 S j.  S > 1 where the wave kernels transform a row as S interleaved
// sub-rows (kx = s mod S), so that lane l of pass s reads ONE stream (L = s + S l) sequentially.
// P of the in-register radix stage: 2^k times 1, 3, 5, 7 or 9, 2 <= P <= 32
FMC_HD constexpr bool mr_supported_P(int P) {
  if (P < 2 || P > 32) return false;
  const int odd = P / (P & -P);
  return odd == 1 || odd == 3 || odd == 5 || odd == 7 || odd == 9;
}
// Wave-family grids N = 64 q whose q is not such a P: S = 2 ... 4 sub-rows (run-time S in the kernels) when that leaves
// 7 <= P <= 24 -- 1344 = 3 x 448, 1728 = 3 x 576, 1920 = 3 x 640, 2304 = 2 x 1152, 2560 = 2 x 1280, 2688 = 3 x 896,
// 3072 = 2 x 1536, 3456 = 3 x 1152, 3584 = 4 x 896, 3840 = 3 x 1280.  0 otherwise.
FMC_HD constexpr int wave_rt_split(int N) {
  if (N % 64 != 0 || N == 2048 || N == 4096) return 0;
  const int q = N / 64;
  if (mr_supported_P(q)) return 0;
  const int smax = N > 4096 ? 8 : 4;        // beyond 4096: up to eight sub-rows (8192 = 8 x 1024, 7168 = 7 x 1024, 6400 = 5 x 1280, ...)
  for (int S = 2; S <= smax; ++S)
    if (q % S == 0 && q / S >= 7 && q / S <= 24 && mr_supported_P(q / S)) return S;
  return 0;
}
FMC_HD constexpr int spec_split(int N) { return N == 4096 ? 4 : (N == 2048 ? 2 : (wave_rt_split(N) ? wave_rt_split(N) : 1)); }
// Grid sizes of the 50-lane family (fmc_mrfft.h): N = 50 P S -- S interleaved sub-rows (kx = s mod S) of 50 P points, P =
// 2^k times 1, 3, 5, 7 or 9.  S = 1 for P <= 24 (100, 150, ..., 1000, 1200); larger grids take the smallest S <= 5 that
// leaves 7 <= P <= 24 (1400 = 2 x 700, 1500 = 3 x 500, 1600 = 2 x 800, 2000 = 2 x 1000, 2500 = 5 x 500, 3000 = 3 x 1000,
// 4000 = 4 x 1000, ...).  Sizes of the wave family (N = 64 P') stay there.  Their rows are drawn as 50 S streams, stream
// L = kx mod 50 S, whichever kernel family transforms them (lane l of sub-row s reads stream s + S l sequentially).
constexpr int MR_LN = 50;
FMC_HD constexpr int mr_split(int N) {       // 0: not a size of the family
  if (N < 2 * MR_LN || N % MR_LN != 0) return 0;
  if (N == 2048 || N == 4096 || (N % 64 == 0 && (mr_supported_P(N / 64) || wave_rt_split(N)))) return 0;
  const int q = N / MR_LN;
  if (q <= 24) return mr_supported_P(q) ? 1 : 0;
  const int smax = N > 4096 ? 8 : 5;        // beyond 4096: up to eight sub-rows (6400 = 8 x 800, 7000 = 7 x 1000, 8000 = 8 x 1000, ...)
  for (int S = 2; S <= smax; ++S)
    if (q % S == 0 && q / S >= 7 && q / S <= 24 && mr_supported_P(q / S)) return S;
  return 0;
}
FMC_HD constexpr bool mr_supported(int N) { return mr_split(N) > 0; }
// Grids of the packed rows (fmc_wavefft.h: packed_row_fft): 128 = 16 x 8 (eight rows per wavefront), 256 = 16 x 16 (four),
// 512 = 16 x 32 (two).  Their rows are drawn as N / 16 streams of sixteen advances (stream q = kx mod N / 16): one seeding
// block per lane serves the 8 / 4 / 2 rows a wavefront transforms at once.
FMC_HD constexpr bool pk_grid(int N) { return N == 128 || N == 256 || N == 512; }
// Grids of the packed SUB-ROWS (round 6; fmc_wavefft.h: pks_accumulate, fmc_kernels.h: k_rows_pks): N = S M with M = 256 (S = 3, 5,
// 6, 7: 768, 1280, 1536, 1792), M = 128 (S = 3, 5, 7, 9: 384, 640, 896, 1152) or M = 64 (S = 3, 5, 7, 9: 192, 320, 448, 576) -- a row is transformed as S interleaved sub-rows
// (kx = s mod S) of M points, FOUR / EIGHT / EIGHT rows at a time on the packed pipeline of the 256 / 128-point grid (64: its own 8 x 8 form), and the window
// outputs are combined by decimation in time in registers.  They replace the one-row-per-wave kernels with N / 64 = 10 ... 28
// values per lane for the device generator (up to 338 registers, one or two waves per SIMD: 0.61-0.79 of the 1024-point row's
// rate per pixel; 1536: 0.93).  Their rows are drawn as N / 16 streams of sixteen advances (64-point sub-rows: N / 8 streams of eight),
// stream t = kx mod SL: lane q of sub-row s reads ONE stream (t = s + S q) sequentially.
FMC_HD constexpr int pks_ct(int N) {         // the sub-row count is a template argument of these grids' kernels
  return N == 768 ? 3 : (N == 1280 ? 5 : (N == 1536 ? 6 : (N == 1792 ? 7 : (N == 640 ? 5 : (N == 896 ? 7 : (N == 1152 ? 9 :
         (N == 576 ? 9 : (N == 448 ? 7 : (N == 320 ? 5 : (N == 192 ? 3 : (N == 384 ? 3 : 0)))))))))));
}
// EVERY other multiple of 64 up to 8192 (the library's largest grid) that has no faster form (128, 256, 512: packed rows; 1024: the dense P = 16 row; 2048 / 4096: pks_p16 below) takes
// the same kernels with the sub-row count at RUN TIME: N = S x 256 (2304 ... 8192, S = 9 ... 32, either parity), else S x 128 (odd
// S = 11 ... 63: 1408 ... 8064), else S x 64 (odd S = 11 ... 127: 704 ... 8128; the table of pks_accumulate goes through the LDS one
// pass at a time).  These were the grids of wave_rt_split (two to four
// one-row-per-wave sub-rows of 7 ... 24 values per lane: 0.58-0.72 of the 1024-point row's rate per pixel), of the chirp-z family
// (704, 832, 960, 1088, ...: about a third) and 1600 / 3200 of the 50-lane family; those families keep their host-coefficient rows.
FMC_HD constexpr int pks_rt(int N) {
  if (N % 64 != 0 || N < 192 || N > 8192 || pk_grid(N) || N == 1024 || N == 2048 || N == 4096 || pks_ct(N)) return 0;
  if (N % 256 == 0) return N / 256;
  if (N % 128 == 0) return N / 128;
  return N / 64;
}
FMC_HD constexpr int pks_split(int N) { return pks_ct(N) ? pks_ct(N) : pks_rt(N); }
// 1024, 2048, 4096 = 4, 8, 16 x 256: the grids of the P = 16 rows draw 64 S' streams per row (S' = 1, 2, 4) -- exactly the N / 16
// streams of sixteen draws the packed sub-rows read, stream t = kx mod N / 16 -- so the packed sub-rows can serve them WITHOUT a
// change of the generator layout (fastmc.hip: pks_p16_from decides which of them they do serve).
FMC_HD constexpr int pks_p16(int N) { return N == 1024 ? 4 : (N == 2048 ? 8 : (N == 4096 ? 16 : 0)); }
FMC_HD constexpr int pks_count(int N) { return pks_split(N) ? pks_split(N) : pks_p16(N); }
// M = 16 * pk_lanes(L0): 256 (L0 = 1) or 128 (L0 = 0); -1: sub-rows of SIXTY-FOUR points (192, 320, 448, 576 = 3, 5, 7, 9 x 64: eight
// rows per wavefront, eight lanes per sub-row, eight draws per generator stream: fmc_wavefft.h: pks64_pass)
FMC_HD constexpr int pks_L0(int N) { return N % 256 == 0 ? 1 : (N % 128 == 0 ? 0 : -1); }
FMC_HD constexpr bool pks_grid(int N) { return pks_split(N) != 0; }
// Generator streams per row: coefficient (ky, kx) is draw kx / SL of stream kx mod SL.
FMC_HD constexpr int stream_lanes(int N) {
  // the packed grids and the sub-row grids of 256 / 128 points: sixteen draws per stream; the sub-row grids of 64 points: eight
  return pks_grid(N) && pks_L0(N) < 0 ? N / 8 : ((pk_grid(N) || pks_grid(N)) ? N / 16 : (mr_supported(N) ? MR_LN * mr_split(N) : WAVE * spec_split(N)));
}
constexpr uint32_t STREAM_SCREEN = 0;
constexpr uint32_t STREAM_LOGAMP = 1;   // counter words 2,3 = global iteration index
constexpr uint32_t STREAM_SUBHARM = 2;  // counter word 0 = mode-pair index m in [0,14)

// ---------------------------------------------------------------- index helpers
// Pupil-window output p (fft-shifted index) of an N-point transform of fft-shifted input k:
//   out[p] = sum_k in[k] * w_N^{((p - h)(k + h)) mod N},  h = N//2   (numpy fftshift both sides)
FMC_HD int shifted_exponent_step(int p, int N) {  // q = (p - h) mod N
  const int h = N / 2;
  int q = (p - h) % N;
  return q < 0 ? q + N : q;
}

}  // namespace fmc
