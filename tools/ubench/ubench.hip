// Instruction-throughput microbenchmarks for gfx950 (run on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 -o ubench ubench.hip && ./ubench
// Each kernel issues ITER x 8 independent ops per lane; reports cycles per wave-instruction per SIMD
// at 1, 2, 4 waves per SIMD (grid = 256 CUs x 4 SIMDs x waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096

template <int OP>
__global__ void k(uint32_t* out, uint32_t seed) {
  uint32_t a[8];
  float f[8];
  double d[8];
  uint64_t q[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 8 + i; f[i] = 1.0f + a[i] * 1e-9f; d[i] = 1.0 + a[i] * 1e-12; q[i] = a[i]; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) f[i] = fmaf(f[i], 1.0000001f, 0.5f);
      if (OP == 1) d[i] = fma(d[i], 1.0000001, 0.5);
      if (OP == 2) a[i] = a[i] * 0xD2511F53u + 1u;                         // v_mul_lo_u32 (+add)
      if (OP == 3) a[i] = __umulhi(a[i], 0xD2511F53u) ^ (uint32_t)it;       // v_mul_hi_u32
      if (OP == 4) { q[i] = (uint64_t)(uint32_t)q[i] * 0xD2511F53u + (q[i] >> 32); }   // v_mad_u64_u32
      if (OP == 5) f[i] = __builtin_amdgcn_logf(f[i]) + 2.0f;
      if (OP == 6) f[i] = __builtin_amdgcn_sinf(f[i]) + 0.3f;
      if (OP == 7) f[i] = __builtin_amdgcn_sqrtf(f[i]) + 1.0f;
      if (OP == 8) a[i] = (a[i] ^ (a[i] >> 7)) + 0x9E3779B9u;               // xor/shift/add: 3 plain ops
      if (OP == 9) d[i] = d[i] + 1.0000001;                                 // v_add_f64
      if (OP == 10) d[i] = d[i] * 1.0000001;                                // v_mul_f64
      if (OP == 11) a[i] = __builtin_amdgcn_alignbit(a[i], a[i], 13) + 1u;  // rotate + add
      if (OP == 12) a[i] = __umul24(a[i], 0x51F53u) + 1u;    // v_mul_u32_u24 / v_mad_u32_u24
      if (OP == 13) d[i] = (double)(float)d[i] + 1.0;                        // cvt f64<->f32
      if (OP == 14) { f[i] = (float)a[i]; a[i] = __float_as_uint(f[i]) + 3u; }           // v_cvt_f32_u32 + add
      if (OP == 15) { d[i] = (double)f[i]; f[i] = __uint_as_float((uint32_t)((uint64_t)__double_as_longlong(d[i]) >> 32)); }  // v_cvt_f64_f32
      if (OP == 16) f[i] = __builtin_amdgcn_cosf(f[i]) + 0.3f;
      if (OP == 17) {   // one complex coefficient as the rows kernel draws it: 2 xoshiro128+ words, Box-Muller, to f64, scale
        uint32_t s0 = a[i], s1 = a[(i + 1) & 7], s2 = a[(i + 2) & 7], s3 = a[(i + 3) & 7];
        uint32_t w[2];
        for (int k2 = 0; k2 < 2; ++k2) {
          w[k2] = s0 + s3;
          const uint32_t t = s1 << 9;
          s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t;
          s3 = __builtin_amdgcn_alignbit(s3, s3, 21);
        }
        a[i] = s0; a[(i + 1) & 7] = s1; a[(i + 2) & 7] = s2; a[(i + 3) & 7] = s3;
        const float u = fmaf((float)w[0], 2.3283064365386963e-10f, 1.1641532182693481e-10f);
        const float t = fmaf((float)w[1], 2.3283064365386963e-10f, 1.1641532182693481e-10f);
        const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));
        d[i] = d[i] * (double)(r * __builtin_amdgcn_cosf(t)) + d[(i + 1) & 7] * (double)(r * __builtin_amdgcn_sinf(t));
      }
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; ++i) r ^= a[i] ^ __float_as_uint(f[i]) ^ (uint32_t)__double_as_longlong(d[i]) ^ (uint32_t)q[i];
  if (r == 0x12345678u) out[0] = r;
}

template <int OP>
void run(const char* name, int ops_per_iter) {
  uint32_t* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps : {1, 2, 4}) {
    const int blocks = 256, threads = 64 * 4 * wps;   // one block per CU, waves spread over 4 SIMDs
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)ITER * 8 * ops_per_iter * wps;
    const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
    printf("%-28s waves/SIMD=%d  %.3f ms  ~%.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, wps, ms, cyc);
  }
}

int main() {
  run<0>("v_fma_f32", 1);
  run<1>("v_fma_f64", 1);
  run<9>("v_add_f64", 1);
  run<10>("v_mul_f64", 1);
  run<2>("v_mul_lo_u32+add (2 ops)", 2);
  run<3>("v_mul_hi_u32+xor (2 ops)", 2);
  run<4>("v_mad_u64_u32", 1);
  run<12>("v_mad_u32_u24", 1);
  run<5>("v_log_f32+add (2 ops)", 2);
  run<6>("v_sin_f32+add (2 ops)", 2);
  run<7>("v_sqrt_f32+add (2 ops)", 2);
  run<8>("xor/shift/add (3 ops)", 3);
  run<11>("alignbit+add (2 ops)", 2);
  run<13>("cvt f64->f32->f64 + add (3)", 3);
  run<14>("v_cvt_f32_u32+add (2 ops)", 2);
  run<15>("v_cvt_f64_f32", 1);
  run<16>("v_cos_f32+add (2 ops)", 2);
  run<17>("one complex coefficient", 1);
  return 0;
}
