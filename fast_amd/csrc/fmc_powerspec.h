// fmc_powerspec.h -- AO-residual phase power spectrum on the GPU.
//
// One thread per spectrum pixel, float64 throughout, the reference's operation order kept so
// that the grid matches numpy to ~1e-13 relative:
//   Fast.compute_powerspec                         fast/fast.py:445-492
//   funcs.turb_powerspectrum_vonKarman             fast/funcs.py:138-173
//   ao_power_spectra.G_AO_PAOLA                    fast/ao_power_spectra.py:225-270
//   ao_power_spectra.Jol_alias_openloop            fast/ao_power_spectra.py:163-223  (only where lf_mask != 0)
//   ao_power_spectra.Jol_noise_openloop            fast/ao_power_spectra.py:148-161
//   ao_power_spectra.logamp_powerspec              fast/ao_power_spectra.py:272-301
// The 2-D Simpson integrals (funcs.py:100-115) are weighted sums  sum_i w_i sum_j w_j P[i][j]
// with the 1-D weights w supplied by the host: each block reduces one row, a second kernel
// reduces the rows, both in a fixed order (deterministic).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace fmc {

constexpr int PS_MAX_LAYERS = 64;
constexpr int PS_THREADS = 256;
constexpr int PS_NQ = 6;   // aniso_servo, alias, noise, fitting, phs_var, logamp_var (+ L per-layer)

struct PsArgs {
  int N, L, ao_mode, alias;
  double dx, wvl, L0, l0, noise, d_wfs, t_loop, t_exp, dth_x, dth_y;
  const double* cn2;
  const double* h;
  const double* wind;          // [L][2]
  const double* mask;          // [N][N] host-supplied mask (mask_mode 0) or null
  int mask_mode;               // 0 supplied, 1 zonal, 2 modal radial cut, 3 modal Zernike (ao_power_spectra.py:119-141)
  int zmax;                    // Zernike modes 1..zmax (mask_mode 3)
  double modal_mult, D_zern;
  const int* noll_n;           // [max(zmax,4)] radial order of Noll mode j = index + 1
  const int* noll_m;           // [max(zmax,4)] signed azimuthal order
  double* mask_out;            // [N][N] or null
  const double* pfilter;       // [N][N] or null
  const double* lgs_z;         // [N][N] or null
  const double* w;             // [N] Simpson weights
  double* powerspec;           // [N][N]
  double* per_layer;           // [L][N][N] or null
  double* logamp_ps;           // [N][N] or null
  double* rowsums;             // [N][PS_NQ + L]
  // the terms of the assembly, as the reference keeps them on the object (fast/fast.py:448-472); any may be null
  double* turb;                // [L][N][N] turb_powerspec   (funcs.turb_powerspectrum_vonKarman)
  double* g_ao;                // [L][N][N] G_ao             (ao_power_spectra.G_AO_PAOLA)
  double* alias_out;           // [L][N][N] alias_powerspec  (Jol_alias_openloop)
  double* noise_out;           // [N][N]    noise_powerspec  (Jol_noise_openloop)
};

__device__ __forceinline__ double np_sinc(double x) {   // numpy.sinc
  const double y = M_PI * (x == 0.0 ? 1.0e-20 : x);
  return sin(y) / y;
}

// 0.033 exp(-k^2/km^2) / (k^2 + k0^2)^(11/6); the caller multiplies by cn2 and zeroes infinities.
__device__ __forceinline__ double vk_base(double fabs_, double km, double k0) {
  return 0.033 * exp(-(fabs_ * fabs_) / (km * km)) / pow(fabs_ * fabs_ + k0 * k0, 11 / 6.0);
}
__device__ __forceinline__ double vk_layer(double base, double cn2) {
  const double v = cn2 * base;
  return isinf(v) ? 0.0 : v;
}

// sum_{j=1..n_noll} |Z~_j(kappa)|^2, centre forced to 1 by the caller
// (ao_power_spectra.zernike_ft 10-21, zernike_squared_filter 54-76).
__device__ __forceinline__ double zernike_sq(double fabs_, double fx, double fy, double D, int n_noll,
                                             const int* nn, const int* mm) {
  const double phi = atan2(fy, fx);
  const double x = fabs_ * D / 2;
  double out = 0.0;
  for (int j = 1; j <= n_noll; ++j) {
    const int n = nn[j - 1], m = mm[j - 1];
    const double rad = 2 * jn(n + 1, x) / x;
    double z2;
    if (m == 0) z2 = (n + 1) * (rad * rad);
    else if ((j & 1) == 0) { const double t = rad * cos(m * phi); z2 = 2 * (n + 1) * (t * t); }
    else { const double t = rad * sin(m * phi); z2 = 2 * (n + 1) * (t * t); }
    out = out + z2;
  }
  return out;
}

__global__ __launch_bounds__(PS_THREADS) void k_powerspec(PsArgs A) {
  // per-wave partial row sums: [0, PS_NQ) written once after the pixel loop, [PS_NQ, PS_NQ + L) the per-layer sums, which
  // lane 0 of each wave accumulates pixel step by pixel step (a per-thread array indexed by the layer would live in scratch
  // memory: 576 B per lane in round 2)
  __shared__ double s_red[PS_THREADS / 64][PS_NQ + PS_MAX_LAYERS];
  const int N = A.N, L = A.L;
  const int iy = blockIdx.x;
  const int mid = N / 2;   // int(N/2.) for both axes
  const double TWO_PI = 2.0 * M_PI;
  const double df = TWO_PI / (N * A.dx);
  const double k = TWO_PI / A.wvl;
  const double c2pk2 = 2 * M_PI * (k * k);
  const double km = 5.92 / A.l0;
  const double k0 = TWO_PI / A.L0;
  const double c_la = TWO_PI * ((TWO_PI / A.wvl) * (TWO_PI / A.wvl));
  const double fy = (iy - N / 2.0) * df;
  const int nq = PS_NQ + L;

  double q[PS_NQ];
#pragma unroll
  for (int i = 0; i < PS_NQ; ++i) q[i] = 0.0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0)
    for (int l = 0; l < L; ++l) s_red[wv][PS_NQ + l] = 0.0;

  // every lane takes part in every step (the per-layer sums are reduced over the wave inside it): lanes beyond the row
  // repeat its last pixel with weight 0 and store nothing
  for (int ix0 = 0; ix0 < N; ix0 += blockDim.x) {
    const bool active = ix0 + (int)threadIdx.x < N;
    const int ix = active ? ix0 + (int)threadIdx.x : N - 1;
    const double wj = active ? A.w[ix] : 0.0;
    const double fx = (ix - N / 2.0) * df;
    const double fabs_ = sqrt(fx * fx + fy * fy);
    const size_t pix = (size_t)iy * N + ix;
    const bool centre = (iy == mid && ix == mid);
    double mask;
    if (A.mask_mode == 0) mask = A.mask[pix];
    else {
      const double fmax = M_PI / A.d_wfs;
      const bool wfs = fabs(fx) <= fmax && fabs(fy) <= fmax;
      double dm;
      if (A.mask_mode == 1) dm = wfs ? 1.0 : 0.0;
      else if (A.mask_mode == 2) dm = (fabs_ <= fmax * A.modal_mult) ? 1.0 : 0.0;
      else dm = centre ? 1.0 : zernike_sq(fabs_, fx, fy, A.D_zern, A.zmax, A.noll_n, A.noll_m);
      mask = (wfs ? 1.0 : 0.0) * (dm < 1 ? dm : 1.0);
    }
    if (A.mask_out && active) A.mask_out[pix] = mask;
    const double base = vk_base(fabs_, km, k0);

    double noise_ps = 0.0;
    if (A.noise > 0.0 && A.ao_mode != 0) {
      if (!centre) {
        const double sx = np_sinc(A.d_wfs * fx / TWO_PI), sy = np_sinc(A.d_wfs * fy / TWO_PI);
        noise_ps = A.noise / (fabs_ * fabs_ * (sx * sx) * (sy * sy));
      }
      noise_ps = mask * noise_ps;
    }

    double ps = 0.0, la = 0.0, gt_sum = 0.0, alias_tot = 0.0;
    const bool do_alias = A.alias && A.ao_mode != 0 && mask != 0.0;
    const double term_0 = (fx * fx) * (fy * fy) / (fabs_ * fabs_ * fabs_ * fabs_);

    // sum over the 120 shifted von Karman spectra (layer-independent part; the reference's
    // per-layer term_2 = cn2_l * this), special-cased centre row / column / pixel (208-213)
    double alias_base = 0.0;
    if (do_alias) {
      for (int sl = -5; sl <= 5; ++sl) {
        const double fys = fy - TWO_PI * sl / A.d_wfs;
        for (int sk = -5; sk <= 5; ++sk) {
          if (sl == 0 && sk == 0) continue;
          const double fxs = fx - TWO_PI * sk / A.d_wfs;
          const double fabs_s = sqrt(fxs * fxs + fys * fys);
          double term_2 = vk_base(fabs_s, km, k0);
          if (isinf(term_2)) term_2 = 0.0;
          double mult;
          if ((sl == 0 && iy == mid) || (sk == 0 && ix == mid)) mult = term_2;
          else if (centre) mult = 0.0;
          else {
            const double t = fx / fys + fy / fxs;
            mult = (t * t) * term_2 * term_0;
          }
          alias_base += mult;
        }
      }
    }

    for (int l = 0; l < L; ++l) {
      const double cn2 = A.cn2[l];
      const double turb = vk_layer(base, cn2);
      const double vx = A.wind[2 * l], vy = A.wind[2 * l + 1];
      const double v_k = fx * vx + fy * vy;
      double G = 1.0;
      if (A.ao_mode != 0) {
        const double drx = A.dth_x / 206265.0 * A.h[l], dry = A.dth_y / 206265.0 * A.h[l];
        const double dr_k = fx * drx + fy * dry;
        const double s = np_sinc(A.t_exp * v_k / TWO_PI);
        const double aniso = 1 - 2 * cos(dr_k - A.t_loop * v_k) * s + s * s;
        if (A.ao_mode == 3) {
          const double aniso_lgs = 1 - 2 * cos(-A.t_loop * v_k) * s + s * s;
          const double Z = A.lgs_z ? A.lgs_z[pix] : (centre ? 1.0 : zernike_sq(fabs_, fx, fy, A.D_zern, 4, A.noll_n, A.noll_m));
          G = mask * (Z * aniso + (1 - Z) * aniso_lgs) + (1 - mask);
        } else {
          G = aniso * mask + (1 - mask);
        }
      }
      double alias = 0.0;
      if (do_alias) {
        const double sc = np_sinc(A.t_exp * v_k / TWO_PI);
        alias = (cn2 * alias_base) * ((sc * sc) * mask);
        if (isnan(alias)) alias = 0.0;
      }
      const double pl = c2pk2 * (turb * G + alias) + noise_ps / L;
      if (active) {
        if (A.per_layer) A.per_layer[(size_t)l * N * N + pix] = pl;
        if (A.turb) A.turb[(size_t)l * N * N + pix] = turb;
        if (A.g_ao) A.g_ao[(size_t)l * N * N + pix] = G;
        if (A.alias_out) A.alias_out[(size_t)l * N * N + pix] = alias;
      }
      ps += pl;
      gt_sum += G * turb;
      alias_tot += alias * c2pk2;
      const double sn = sin(A.wvl * A.h[l] * (fabs_ * fabs_) / (4 * M_PI));
      double lal = turb * c_la;
      lal *= sn * sn;
      if (A.pfilter) lal *= A.pfilter[pix];
      la += lal;
      const double wl = wave_sum(wj * pl);           // fixed-order reduction over the 64 pixels of this step
      if (lane == 0) s_red[wv][PS_NQ + l] += wl;
    }
    if (active) {
      A.powerspec[pix] = ps;
      if (A.logamp_ps) A.logamp_ps[pix] = la;
      if (A.noise_out) A.noise_out[pix] = noise_ps;
    }
    q[0] += wj * (gt_sum * mask * c2pk2);
    q[1] += wj * alias_tot;
    q[2] += wj * noise_ps;
    q[3] += wj * (ps * (1 - mask));
    q[4] += wj * ps;
    q[5] += wj * la;
  }
  // block reduction in a fixed order
#pragma unroll
  for (int i = 0; i < PS_NQ; ++i) {
    const double v = wave_sum(q[i]);
    if (lane == 0) s_red[wv][i] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nq; i += blockDim.x) {
    double v = 0.0;
    for (int w2 = 0; w2 < PS_THREADS / 64; ++w2) v += s_red[w2][i];
    A.rowsums[(size_t)iy * nq + i] = v;
  }
}

// scalars[i] = sum_iy w[iy] * rowsums[iy][i]: one block per quantity, fixed-shape tree (deterministic)
__global__ __launch_bounds__(256) void k_ps_scalars(const double* rowsums, const double* w, int N, int nq, double* scalars) {
  __shared__ double s_part[4];
  const int i = blockIdx.x;
  double v = 0.0;
  for (int iy = threadIdx.x; iy < N; iy += blockDim.x) v += w[iy] * rowsums[(size_t)iy * nq + i];
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) scalars[i] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

}  // namespace fmc
