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
  fastmc_destroy(h);
  printf("C-ABI OK mean=%.6f split_maxdiff=%.3g hist_total=%lld last_error_after_bad_call=\"%s\"\n", mean, maxdiff, total,
         fastmc_last_error());
  return (mean > 0.0 && mean <= 1.1 && maxdiff == 0.0 && total == 1200) ? 0 : 4;
}
