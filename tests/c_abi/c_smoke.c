/* Plain-C client of include/fastmc.h: no Python, no C++ -- the boundary a cgo / JNI / ctypes binding would use.
 *   gcc -std=c99 -O1 -I include tests/c_abi/c_smoke.c -o c_smoke -L fast_amd -lfastmc -Wl,-rpath,$PWD/fast_amd -lm
 * Prints the mean coupled power of 2000 iterations on a flat-ish spectrum and checks it against the
 * same run split in two calls (results depend only on the seed and the global realisation index). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fastmc.h"

#define CHECK(call)                                                                 \
  do {                                                                              \
    int rc_ = (call);                                                               \
    if (rc_ < 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, fastmc_last_error()); return 1; } \
  } while (0)

int main(void) {
  const int N = 256, Np = 40, n_real = 1000;
  int ndev = 0;
  CHECK(fastmc_device_count(&ndev));
  if (ndev < 1) { fprintf(stderr, "no device\n"); return 2; }
  double* ps = malloc(sizeof(double) * N * N);
  double* W = malloc(sizeof(double) * Np * Np);
  double* out = malloc(sizeof(double) * 2 * n_real);
  double* out2 = malloc(sizeof(double) * 2 * n_real);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      const double fx = (i - N / 2) * 0.25, fy = (j - N / 2) * 0.25, k2 = fx * fx + fy * fy;
      ps[i * N + j] = 2e-3 * pow(k2 + 0.04, -11.0 / 6.0);      /* von Karman-like, finite at the origin */
    }
  for (int i = 0; i < Np * Np; ++i) W[i] = 1.0;
  fastmc_t* h = NULL;
  CHECK(fastmc_create(&h, 0, N, Np, FASTMC_F64));
  CHECK(fastmc_set_spectrum(h, ps, 0.25));
  CHECK(fastmc_set_pupil(h, W, (N - Np) / 2, 0.01));
  CHECK(fastmc_run(h, 1234u, 0, n_real, NULL, 0.01, 0, out));
  /* the same realisations in two calls: [0, 400) and [400, 1000) */
  CHECK(fastmc_run(h, 1234u, 0, 400, NULL, 0.01, 0, out2));
  double* tail = malloc(sizeof(double) * 2 * 600);
  CHECK(fastmc_run(h, 1234u, 400, 600, NULL, 0.01, 0, tail));
  double mean = 0.0, maxdiff = 0.0;
  for (int i = 0; i < 2 * n_real; ++i) mean += out[i] / (2 * n_real);
  for (int s = 0; s < 2; ++s) {            /* layout: [Re-screen results | Im-screen results] of the call's range */
    for (int i = 0; i < 400; ++i) maxdiff = fmax(maxdiff, fabs(out[s * n_real + i] - out2[s * 400 + i]));
    for (int i = 0; i < 600; ++i) maxdiff = fmax(maxdiff, fabs(out[s * n_real + 400 + i] - tail[s * 600 + i]));
  }
  int64_t bins[66];
  CHECK(fastmc_histogram(h, -60.0, 10.0, 64, bins));
  long long total = 0;
  for (int i = 0; i < 66; ++i) total += bins[i];
  if (fastmc_run(h, 1u, 0, -5, NULL, 0.0, 0, out) >= 0) { fprintf(stderr, "negative count accepted\n"); return 3; }
  /* kernel families: 256 = 64 x 4 runs the wave FFT kernels; the direct family gives the same powers to rounding */
  if (fastmc_kernel_path(h, -1) != 1) { fprintf(stderr, "256^2 is not on the wave kernels\n"); return 5; }
  CHECK(fastmc_kernel_path(h, 0));
  CHECK(fastmc_run(h, 1234u, 0, 400, NULL, 0.01, 0, tail));
  double famdiff = 0.0;
  for (int i = 0; i < 800; ++i) famdiff = fmax(famdiff, fabs(tail[i] - out2[i]) / out2[i]);
  if (fastmc_kernel_path(h, 3) >= 0) { fprintf(stderr, "50-lane kernels accepted for N = 256\n"); return 6; }
  CHECK(fastmc_kernel_path(h, 1));
  /* one process, one device: communicator of one rank, all-gather + histogram all-reduce through RCCL (or a refusal
   * with FASTMC_ECOMM when librccl is absent / disabled, which a caller answers with the host copy it already has) */
  CHECK(fastmc_run(h, 1234u, 0, 400, NULL, 0.01, 0, out2));
  fastmc_t* group[1];
  group[0] = h;
  int world = -1, rank = -2;
  const int rc_comm = fastmc_comm_init_all(group, 1);
  double gathered_diff = -1.0;
  long long gtotal = -1;
  if (rc_comm >= 0) {
    CHECK(fastmc_comm_world(h, &world, &rank));
    double* all = malloc(sizeof(double) * 800);
    int64_t gb[66];
    CHECK(fastmc_comm_gather_all(group, 1, 800, all, gb, -60.0, 10.0, 64));
    gathered_diff = 0.0;
    for (int i = 0; i < 800; ++i) gathered_diff = fmax(gathered_diff, fabs(all[i] - out2[i]));
    gtotal = 0;
    for (int i = 0; i < 66; ++i) gtotal += gb[i];
    free(all);
    fastmc_comm_destroy(h);
  }
  fastmc_destroy(h);
  /* a round decimal grid: 200 = 50 x 4 runs the 50-lane kernels */
  fastmc_t* h2 = NULL;
  CHECK(fastmc_create(&h2, 0, 200, 40, FASTMC_F64));
  const int path200 = fastmc_kernel_path(h2, -1);
  fastmc_destroy(h2);
  /* the generator at the reference's precision (what fast_amd.Fast selects) on a 1024-point row, and the clock the row kernel ran in */
  {
    const int NB = 1024, NpB = 82;
    fastmc_t* hb = NULL;
    double* psb = malloc(sizeof(double) * NB * NB);
    double* Wb = malloc(sizeof(double) * NpB * NpB);
    double r64[16], r32[16], ghz = 0.0, span_us = 0.0;
    for (int i = 0; i < NB * NB; ++i) psb[i] = 1e-4;
    for (int i = 0; i < NpB * NpB; ++i) Wb[i] = 1.0;
    CHECK(fastmc_create(&hb, 0, NB, NpB, FASTMC_F64));
    if (fastmc_last_clock(hb, &ghz, &span_us) != FASTMC_ESTATE) { fprintf(stderr, "a clock before any launch\n"); return 9; }
    CHECK(fastmc_set_spectrum(hb, psb, 0.1));
    CHECK(fastmc_set_pupil(hb, Wb, (NB - NpB) / 2, 0.01));
    CHECK(fastmc_run(hb, 7, 0, 8, NULL, 0.0, 0, r64));                 /* a float64 handle draws at float64 precision as created */
    CHECK(fastmc_last_clock(hb, &ghz, &span_us));
    CHECK(fastmc_set_rng_precision(hb, FASTMC_F32));                   /* the opt-in float32 draw */
    CHECK(fastmc_run(hb, 7, 0, 8, NULL, 0.0, 0, r32));
    {
      double again[16];
      CHECK(fastmc_set_rng_precision(hb, FASTMC_F64));
      CHECK(fastmc_run(hb, 7, 0, 8, NULL, 0.0, 0, again));
      for (int i = 0; i < 16; ++i) if (again[i] != r64[i]) { fprintf(stderr, "the default generator is not FASTMC_F64\n"); return 9; }
    }
    double d = 0.0;
    for (int i = 0; i < 16; ++i) d = fmax(d, fabs(r64[i] / r32[i] - 1.0));
    printf("float64 generator vs float32 draw, same seed: max rel diff %.3g; row-kernel clock %.3f GHz over %.1f us\n", d, ghz, span_us);
    if (!(d > 0.0 && d < 1e-3) || !(ghz > 1.0 && ghz < 2.6) || !(span_us > 1.0)) return 9;
    fastmc_destroy(hb);
    free(psb); free(Wb);
  }
  printf("families: wave-vs-direct %.3g, path(200)=%d; comm rc=%d world=%d rank=%d gathered_diff=%.3g hist=%lld\n", famdiff, path200,
         rc_comm, world, rank, gathered_diff, gtotal);
  if (famdiff > 1e-9 || path200 != 3) return 7;
  if (rc_comm >= 0 && (world != 1 || rank != 0 || gathered_diff != 0.0 || gtotal != 800)) return 8;
  printf("C-ABI OK mean=%.6f split_maxdiff=%.3g hist_total=%lld last_error_after_bad_call=\"%s\"\n", mean, maxdiff, total,
         fastmc_last_error());
  return (mean > 0.0 && mean <= 1.1 && maxdiff == 0.0 && total == 1200) ? 0 : 4;
}
