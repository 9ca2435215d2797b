// Gate for moving the row transform's radix-16 stage onto the FP64 matrix pipe.  Run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o ubench_mfma64 ubench_mfma64.hip && ./ubench_mfma64
// (i)  cycles per back-to-back v_mfma_f64_16x16x4_f64 (and the 4x4x4 four-block form) on one SIMD, 1 and 4 accumulators,
//      one wave per SIMD and four waves per SIMD;
// (ii) does a float64 VALU stream of OTHER waves of the same SIMD keep its rate beside it?  Sixteen waves per workgroup = four
//      per SIMD; role = bit SHIFT of the wave index (waves w and w + 4 share a SIMD: bit 2 or 3 puts two waves of each role
//      on every SIMD); and the same two streams interleaved in every wave (G fma per MFMA).
// A DFT-16 over the 64 columns of a 1024-point row is 64 of these MFMAs (4 real products x 4 column blocks x 4 k blocks) in
// place of 148 float64 VALU instructions: it pays only if 64 MFMAs cost the SIMD's VALU issue well under 148 x 4 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 1024

typedef double d4 __attribute__((ext_vector_type(4)));

enum { VALU = 0, MFMA16 = 1, MFMA4 = 2 };

// MODE 0: every wave VALU; 1: every wave MFMA (NACC accumulators); 2: role split by bit SHIFT; 3: every wave 8 MFMA + G*8 fma
template <int KIND, int MODE, int SHIFT, int NACC, int G>
__global__ __launch_bounds__(1024) void k(uint64_t* out, double seed, int waves) {
  const int w = threadIdx.x >> 6;
  if (w >= waves) return;
  double d[8];
  for (int i = 0; i < 8; ++i) d[i] = 1.0 + (threadIdx.x * 8 + i) * 1e-12 * seed;
  d4 acc[4];
  double acc1[4];
  for (int i = 0; i < 4; ++i) { acc[i] = d4{0, 0, 0, 0}; acc1[i] = 0; }
  const double a = 1.0 + threadIdx.x * 1e-9 * seed, b = 0.5 + threadIdx.x * 1e-10 * seed;
  const bool roleB = ((w >> SHIFT) & 1) != 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; ++it) {
    const bool do_valu = MODE == 0 || (MODE == 2 && !roleB);
    const bool do_mfma = MODE == 1 || (MODE == 2 && roleB);
    if (do_valu) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = fma(d[i], 1.0000001, 0.5);
    }
    if (do_mfma) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == MFMA16) acc[i % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i % NACC], 0, 0, 0);
        if (KIND == MFMA4) acc1[i % NACC] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i % NACC], 0, 0, 0);
      }
    }
    if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == MFMA16) acc[i % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i % NACC], 0, 0, 0);
        if (KIND == MFMA4) acc1[i % NACC] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i % NACC], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep the order as written: one MFMA, then its G fillers
#pragma unroll
        for (int g = 0; g < G; ++g) d[(i + g) & 7] = fma(d[(i + g) & 7], 1.0000001, 0.5);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  double r = 0;
  for (int i = 0; i < 8; ++i) r += d[i];
  for (int i = 0; i < 4; ++i) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w + acc1[i];
  if (r == 0.12345) out[1] = 1;
  if (blockIdx.x == 128 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND, int MODE, int SHIFT, int NACC, int G>
float time_one(uint64_t* d, int waves, uint64_t* cyc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, MODE, SHIFT, NACC, G>), dim3(256), dim3(1024), 0, 0, d, 1.0, waves);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, MODE, SHIFT, NACC, G>), dim3(256), dim3(1024), 0, 0, d, 1.0, waves);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(cyc, d, 8, hipMemcpyDeviceToHost);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms;
}

template <int KIND>
void suite(const char* name, uint64_t* d) {
  uint64_t c;
  printf("== %s\n", name);
  // (i) back-to-back issue.  s_memtime ticks at the shader clock here (MI355X_MICROARCH.md, cycle constants).
  float ms = time_one<KIND, 1, 0, 1, 0>(d, 4, &c);
  printf("(i) one wave per SIMD, 1 accumulator : %.3f ms, %.1f cycles per MFMA (stamped), %.1f at 2.4 GHz (wall)\n", ms, c / (8.0 * ITER), ms * 2.4e6 / (8.0 * ITER));
  ms = time_one<KIND, 1, 0, 4, 0>(d, 4, &c);
  printf("(i) one wave per SIMD, 4 accumulators: %.3f ms, %.1f cycles per MFMA (stamped), %.1f at 2.4 GHz (wall)\n", ms, c / (8.0 * ITER), ms * 2.4e6 / (8.0 * ITER));
  ms = time_one<KIND, 1, 0, 4, 0>(d, 16, &c);
  printf("(i) four waves per SIMD, 4 accumulators: %.3f ms, %.1f SIMD cycles per MFMA (stamped / 4), %.1f at 2.4 GHz (wall / 4)\n", ms, c / (32.0 * ITER), ms * 2.4e6 / (32.0 * ITER));
  // (ii) co-execution: per SIMD the split runs half the VALU work and half the MFMA work
  const float v = time_one<KIND, 0, 0, 4, 0>(d, 16, &c);
  printf("(ii) all sixteen waves VALU (32 fma per iteration): %.3f ms, %.2f SIMD cycles per fma (stamped / 4)\n", v, c / (4.0 * 32.0 * ITER));
  const float m = time_one<KIND, 1, 0, 4, 0>(d, 16, &c);
  const float s2 = time_one<KIND, 2, 2, 4, 0>(d, 16, &c), s3 = time_one<KIND, 2, 3, 4, 0>(d, 16, &c);
  const float s0 = time_one<KIND, 2, 0, 4, 0>(d, 16, &c);
  printf("(ii) all VALU %.3f ms, all MFMA %.3f ms; split by wave bit 2 / 3 / 0: %.3f / %.3f / %.3f ms  (serial %.3f, full overlap %.3f)\n", v, m, s2, s3, s0,
         (v + m) / 2, (v > m ? v : m) / 2);
  // interleaved in every wave: 8 MFMA + 8 G fma per iteration against 8 MFMA alone (m) and 8 G fma alone (v * G / 4)
  const float i1 = time_one<KIND, 3, 0, 4, 1>(d, 16, &c), i2 = time_one<KIND, 3, 0, 4, 2>(d, 16, &c), i4 = time_one<KIND, 3, 0, 4, 4>(d, 16, &c),
              i8 = time_one<KIND, 3, 0, 4, 8>(d, 16, &c), i16 = time_one<KIND, 3, 0, 4, 16>(d, 16, &c);
  printf("(ii) interleaved, G fma per MFMA, four waves per SIMD: G=1 %.3f  G=2 %.3f  G=4 %.3f  G=8 %.3f  G=16 %.3f ms\n", i1, i2, i4, i8, i16);
  printf("      the fma alone would take              : G=1 %.3f  G=2 %.3f  G=4 %.3f  G=8 %.3f  G=16 %.3f ms;  the MFMA alone %.3f ms\n", v / 4, v / 2, v, 2 * v, 4 * v, m);
  const float j4 = time_one<KIND, 3, 0, 4, 4>(d, 4, &c), j8 = time_one<KIND, 3, 0, 4, 8>(d, 4, &c), j16 = time_one<KIND, 3, 0, 4, 16>(d, 4, &c);
  const float v1 = time_one<KIND, 0, 0, 4, 0>(d, 4, &c), m1 = time_one<KIND, 1, 0, 4, 0>(d, 4, &c);
  printf("(ii) interleaved, ONE wave per SIMD: G=4 %.3f  G=8 %.3f  G=16 %.3f ms; fma alone G=4 %.3f, MFMA alone %.3f ms\n", j4, j8, j16, v1, m1);
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 16);
  hipMemset(d, 0, 16);
  suite<MFMA16>("v_mfma_f64_16x16x4_f64 (2048 flop per wave-instruction)", d);
  suite<MFMA4>("v_mfma_f64_4x4x4_4b_f64 (512 flop per wave-instruction)", d);
  return 0;
}
