// emu_gen64.cpp -- host execution of the fast float64 Box-Muller of fmc_gen64.h (the arithmetic the row kernels run in
// MODE 2), for tests/test_emu_gen64.py:  emu_gen64 <in.bin> <out.bin>
//   in : n records of four uint32 (a, b, a2, b2);   out : n records of two float64 (re, im), amp = 1.
// Only the 1 / sqrt seed differs from the device (a correctly rounded float32 one here, v_rsq_f32 there); the cubic
// step after it forgets the difference.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fmc_gen64.h"

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  const size_t n = (size_t)bytes / 16;
  std::vector<uint32_t> w(4 * n);
  if (fread(w.data(), 16, n, f) != n) { fprintf(stderr, "short read\n"); return 1; }
  fclose(f);
  std::vector<fmc::Gen64Entry> tab(fmc::GEN64_LOG_ENTRIES + fmc::GEN64_TRIG_ENTRIES);
  fmc::gen64_build_table(tab.data());
  std::vector<double> out(2 * n);
  for (size_t i = 0; i < n; ++i)
    fmc::box_muller_f64_fast(w[4 * i], w[4 * i + 1], w[4 * i + 2] | 1u, w[4 * i + 3], 1.0, tab.data(), out[2 * i], out[2 * i + 1]);
  f = fopen(argv[2], "wb");
  if (!f) { perror(argv[2]); return 1; }
  fwrite(out.data(), 16, n, f);
  fclose(f);
  return 0;
}
