// fmc_npstream.h -- numpy's Generator.normal stream, drawn on the device (GPU_RNG 'numpy').
//
// The reference draws every coefficient from ONE sequential stream, `funcs._R = numpy.random.default_rng(seed)`
// (fast/funcs.py:21): `_R.normal(0, 1, shape) + 1j * _R.normal(0, 1, shape)` (funcs.py:352-356) after the log-amplitude draws
// (fast.py:123, 639-645).  "Identical RNG seeds" therefore means reproducing that stream: PCG64 (128-bit LCG, XSL-RR output)
// feeding numpy's 256-layer ziggurat, in which a normal consumes ONE 64-bit word 97.8 % of the time and two or more otherwise
// (wedge test, tail loop, restarts) -- so the word a normal starts at depends on every earlier rejection.  Round 3 drew this
// stream on the host (114 iterations/s end to end).  Here it is drawn on the device, per array of n normals (a "segment": the
// real parts of a chunk followed by its imaginary parts, the log-amplitudes ...), all enqueued without a host round trip.
// The three-pass form, for segments of any length:
//
//   classify  one workgroup per TILE of T = 8192 consecutive words.  A PCG64 state can be advanced by any distance in
//             O(log) 128-bit multiply-adds, so every thread jumps to its own 8-word pieces and treats EVERY word as if a
//             normal started there: fast (one word) or slow -- then it runs numpy's slow path to the end on a private copy of
//             the generator and records an EVENT (position, words consumed f, value).  Events come out in position order
//             (threads own consecutive pieces; a block scan assigns the slots).  Then the tile's transfer map: for each
//             entry offset e < K (the first normal of the tile starts e words in, because the previous tile's last normal
//             spilled over) the number of normals that start in the tile and the offset handed to the next tile -- and, per
//             event, for which entry offsets it is a start.  (The slow path diverges: the words of a sub-tile that need it
//             are queued and evaluated one per lane.)
//   scan      one workgroup: every tile's true entry offset and the index of its first normal (see k_nps_scan); the word
//             after the n-th normal = the words this array consumed, and the generator state there (one more jump) for the
//             next segment.
//   emit      per tile, with its entry offset known: the words once more, the skipped ones masked (inside a slow normal's
//             span), a prefix sum for the output index, value = rabs * wi[idx] (or the event's) -> out[], in the order numpy
//             writes them.
//
// and the ONE-PASS form (k_nps_onepass, at the end of this file) when the segment's normals fit a device buffer of their own:
// the same classification, but each tile keeps its values in registers, learns its entry offset and first index from its
// predecessors while they still run (decoupled look-back) and writes its normals itself -- every word generated once.
// A spill beyond K words, more events than a tile holds, or a stream longer than the launch allowed for raises an OVERFLOW flag
// and the caller redoes that chunk with numpy's own draws (never observed: K = 16 is ~8 consecutive rejections).
// The ziggurat tables are not in this source: fast_amd/npnormal.py reads them out of the numpy that is installed through a
// crafted bit generator and checks the restatement against numpy's own stream before the device is trusted with it.
// log1p / exp here are ocml's, numpy's are the host libm's: a tail value can differ in its last bit and an acceptance could flip
// only when the two sides of the test agree to 2^-52 (probability ~1e-16 per slow normal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fmc {

typedef unsigned __int128 u128;

#ifndef FMC_NPS_T
#define FMC_NPS_T 8192
#endif
constexpr int NPS_T = FMC_NPS_T;      // words per tile
constexpr int NPS_THREADS = 256;
constexpr int NPS_WPT = 8;            // consecutive words per thread and sub-tile
constexpr int NPS_SUB = NPS_THREADS * NPS_WPT;      // 2048 words per sub-tile
constexpr int NPS_NSUB = NPS_T / NPS_SUB;           // 8
constexpr int NPS_K = 16;             // entry offsets a tile's map covers
constexpr int NPS_EVCAP = NPS_T / 16; // events per tile (expected T / 45: 180 at T = 8192, sigma 13)

struct NpsEvent {
  uint32_t pos;      // word in the tile at which the slow normal would start | (bit e: it IS a start when the tile is entered at offset e) << 16
  uint32_t f;        // words it consumes
  double v;          // its value
};

struct NpsTables {   // numpy's ziggurat (fast_amd/npnormal.py)
  double wi[256];
  uint64_t ki[256];
  double fi[256];
};

struct NpsJump {     // a^(2^k) and the matching increments of the PCG64 LCG: state -> a state + c  applied 2^k times
  u128 a[64];
  u128 c[64];        // for increment 1 (scaled by the stream's increment at use: c_k(inc) = c_k(1) * inc)
};

struct NpsSegArgs {
  const u128* state;        // generator state at the start of the segment (device)
  u128 inc;
  uint64_t n;               // normals wanted
  int64_t ntiles;           // tiles classified (upper bound of the stream length)
  const NpsTables* tab;
  const NpsJump* jump;
  NpsEvent* events;         // [ntiles][NPS_EVCAP]
  uint32_t* evcount;        // [ntiles]
  uint32_t* maps;           // [ntiles][NPS_K]: normals started | exit offset << 16
  u128* tile_state;         // [ntiles] generator state at the first word of each tile (k_nps_tilestates)
  uint8_t* tile_e;          // [ntiles] entry offset (scan)
  uint64_t* tile_base;      // [ntiles] index of the tile's first normal (scan)
  u128* state_out;          // state after the segment (scan)
  uint64_t* consumed;       // words the segment consumed (scan)
  uint32_t* overflow;       // != 0: give up on the device for this chunk
  uint32_t flags;           // NPS_SCAN_GENERAL
};

__device__ __forceinline__ u128 nps_advance(u128 s, u128 inc, uint64_t dist, const NpsJump* J) {
  for (int k = 0; dist; ++k, dist >>= 1)
    if (dist & 1) s = J->a[k] * s + J->c[k] * inc;
  return s;
}
__device__ __forceinline__ uint64_t nps_next(u128& s, u128 inc) {       // pcg64: step, then XSL-RR of the new state
  s = s * ((((u128)0x2360ED051FC65DA4ull) << 64) | 0x4385DF649FCCF645ull) + inc;
  const uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s, v = hi ^ lo;
  const unsigned rot = (unsigned)(hi >> 58);
  return (v >> rot) | (v << ((64u - rot) & 63u));
}
__device__ __forceinline__ double nps_double(u128& s, u128 inc) { return (double)(nps_next(s, inc) >> 11) * (1.0 / 9007199254740992.0); }

// The fast-path value of the word r -- numpy's `x = rabs * wi[idx]; if (sign) x = -x` with rabs = bits 9 ... 60 -- in five
// instructions: the 52 bits dropped under the exponent of 2^52 are the integer as a double (exact, no u64 -> f64 conversion),
// and the sign bit (bit 8 of r) is XORed in.  rabs comes back for the ziggurat's `rabs < ki[idx]`.
__device__ __forceinline__ double nps_fast_value(uint64_t r, const double* wi, uint64_t& rabs) {
  const uint32_t lo = (uint32_t)r, hi = (uint32_t)(r >> 32);
  const uint32_t mlo = __builtin_amdgcn_alignbit(hi, lo, 9), mhi = (hi >> 9) & 0xfffffu;
  rabs = ((uint64_t)mhi << 32) | mlo;
  const double d = __hiloint2double((int)(mhi | 0x43300000u), (int)mlo) - 4503599627370496.0;
  const double x = __dmul_rn(d, wi[lo & 0xffu]);
  return __hiloint2double(__double2hiint(x) ^ (int)((lo << 23) & 0x80000000u), __double2loint(x));
}

// numpy/random/src/distributions/distributions.c: random_standard_normal, from the word `r` on, on the private generator (s, inc).
// Returns the value; f = words consumed including r.  No FMA contraction in the acceptance tests (numpy's are plain C on x86-64).
__device__ inline double nps_slow(uint64_t r, u128 s, u128 inc, const double* wi, const uint64_t* ki, const double* fi, uint32_t& f) {
  const double R = 3.6541528853610087963519472518, INV_R = 0.27366123732975827203338247596;
  f = 1;
  for (;;) {
    const int idx = (int)(r & 0xff);
    const uint64_t rabs = (r >> 9) & 0x000fffffffffffffull;
    double x = __dmul_rn((double)rabs, wi[idx]);
    if ((r >> 8) & 1) x = -x;
    if (rabs < ki[idx]) return x;
    if (idx == 0) {
      for (;;) {
        const double xx = __dmul_rn(-INV_R, log1p(-nps_double(s, inc)));
        const double yy = -log1p(-nps_double(s, inc));
        f += 2;
        if (__dadd_rn(yy, yy) > __dmul_rn(xx, xx)) return ((rabs >> 8) & 1) ? -__dadd_rn(R, xx) : __dadd_rn(R, xx);
        if (f > 4096) return 0.0;      // (cannot happen; keeps a corrupted table from hanging the device)
      }
    } else {
      const double u = nps_double(s, inc);
      f += 1;
      if (__dadd_rn(__dmul_rn(fi[idx - 1] - fi[idx], u), fi[idx]) < exp(__dmul_rn(__dmul_rn(-0.5, x), x))) return x;
    }
    r = nps_next(s, inc);
    f += 1;
    if (f > 4096) return 0.0;
  }
}

// exclusive block scan of one small integer per thread (256 threads, 4 waves); `total` = the sum
// inclusive sum over the 64 lanes of a wave on the DPP crossbar (row shifts, then the two row broadcasts): VALU only
__device__ __forceinline__ uint32_t nps_wave_scan(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);    // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);    // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);    // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);    // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2 and 3
  return (uint32_t)x;
}
__device__ __forceinline__ uint32_t nps_block_scan(uint32_t v, uint32_t* s_wave, uint32_t& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t inc = nps_wave_scan(v);
  if (lane == 63) s_wave[w] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int i = 0; i < w; ++i) base += s_wave[i];
  total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  __syncthreads();
  return base + inc - v;
}

// The jump table in the LDS with the stream's increment folded in: state -> a[k] state + cinc[k] is 2^k steps.  (Read from global
// memory, the dependent loads of a jump cost a microsecond each.)
struct NpsLdsJump {
  u128 a[64];
  u128 cinc[64];
};
__device__ __forceinline__ void nps_load_jump(NpsLdsJump* L, const NpsJump* J, u128 inc) {      // the caller's barrier follows
  if (threadIdx.x < 64) { L->a[threadIdx.x] = J->a[threadIdx.x]; L->cinc[threadIdx.x] = J->c[threadIdx.x] * inc; }
}
__device__ __forceinline__ u128 nps_advance_lds(u128 s, uint64_t dist, const NpsLdsJump* L) {
  for (int k = 0; dist; ++k, dist >>= 1)
    if (dist & 1) s = L->a[k] * s + L->cinc[k];
  return s;
}

// state at the first word of every tile of a segment (one thread per tile)
__global__ __launch_bounds__(NPS_THREADS) void k_nps_tilestates(NpsSegArgs A) {
  __shared__ NpsLdsJump s_jump;
  nps_load_jump(&s_jump, A.jump, A.inc);
  __syncthreads();
  const int64_t tile = (int64_t)blockIdx.x * NPS_THREADS + threadIdx.x;
  if (tile < A.ntiles) A.tile_state[tile] = nps_advance_lds(*A.state, (uint64_t)tile * NPS_T, &s_jump);
}

// ---------------------------------------------------------------- classify
constexpr int NPS_QCAP = 256;          // slow words of one sub-tile (expected 45, sigma 7)
struct NpsQueued {                     // a slow word waiting for numpy's slow path: the generator right after it, the word, where it was
  u128 s;
  uint64_t r;
  uint32_t pos;                        // in the tile
  uint32_t f;                          // out: words consumed
  double v;                            // out: value
};

__global__ __launch_bounds__(NPS_THREADS) void k_nps_classify(NpsSegArgs A) {
  __shared__ double s_wi[256];
  __shared__ uint64_t s_ki[256];
  __shared__ double s_fi[256];
  __shared__ uint32_t s_ev[NPS_EVCAP];        // pos | f << 16 of the tile's events, in position order
  __shared__ uint32_t s_mask[NPS_EVCAP];      // bit e: the event is a START on the path that enters the tile at offset e
  __shared__ NpsQueued s_q[NPS_QCAP];
  __shared__ uint16_t s_qidx[NPS_SUB];
  __shared__ uint32_t s_wave[4];
  __shared__ NpsLdsJump s_jump;
  __shared__ uint32_t s_qn[2];                // queue length, double-buffered over the sub-tiles
  const int64_t tile = blockIdx.x;
  const int t = threadIdx.x;
  s_wi[t] = A.tab->wi[t]; s_ki[t] = A.tab->ki[t]; s_fi[t] = A.tab->fi[t];
  nps_load_jump(&s_jump, A.jump, A.inc);
  if (t == 0) s_qn[0] = s_qn[1] = 0;
  __syncthreads();
  u128 piece = nps_advance_lds(A.tile_state[tile], (uint64_t)t * NPS_WPT, &s_jump);      // the generator before this thread's piece of the sub-tile
  const u128 a_sub = s_jump.a[11], c_sub = s_jump.cinc[11];                     // 2^11 = NPS_SUB words on
  static_assert(NPS_SUB == 2048, "the sub-tile stride is the 2^11 entry of the jump table");
  NpsEvent* ev_out = A.events + (size_t)tile * NPS_EVCAP;
  uint32_t nev = 0;
  for (int sub = 0; sub < NPS_NSUB; ++sub) {
    // the eight words of this thread's piece: only WHICH are slow (numpy's slow path diverges; it runs below, one queued word per lane)
    uint32_t slowmask = 0, cnt = 0;
    u128 s = piece;
#pragma unroll
    for (int i = 0; i < NPS_WPT; ++i) {
      const uint64_t r = nps_next(s, A.inc);
      const int idx = (int)(r & 0xff);
      const uint64_t rabs = (r >> 9) & 0x000fffffffffffffull;
      if (rabs >= s_ki[idx]) {
        slowmask |= 1u << i; ++cnt;
        const uint32_t q = atomicAdd(&s_qn[sub & 1], 1u);
        if (q < (uint32_t)NPS_QCAP) {
          s_q[q].s = s; s_q[q].r = r; s_q[q].pos = (uint32_t)(sub * NPS_SUB + t * NPS_WPT + i);
          s_qidx[t * NPS_WPT + i] = (uint16_t)q;
        }
      }
    }
    uint32_t total;
    const uint32_t slot0 = nev + nps_block_scan(cnt, s_wave, total);           // (its barriers also publish the queue)
    const uint32_t qn = min(s_qn[sub & 1], (uint32_t)NPS_QCAP);
    if (t == 0) {
      if (s_qn[sub & 1] > (uint32_t)NPS_QCAP) atomicOr(A.overflow, 2u);
      s_qn[(sub + 1) & 1] = 0;           // the next sub-tile's queue (its pushes come after this iteration's last barrier)
    }
    if ((uint32_t)t < qn) {
      uint32_t f;
      s_q[t].v = nps_slow(s_q[t].r, s_q[t].s, A.inc, s_wi, s_ki, s_fi, f);
      s_q[t].f = f;
      if (f >= 0xffffu) atomicOr(A.overflow, 1u);
    }
    __syncthreads();
    uint32_t slot = slot0;
#pragma unroll
    for (int i = 0; i < NPS_WPT; ++i)
      if ((slowmask >> i) & 1u) {
        const uint32_t q = s_qidx[t * NPS_WPT + i];
        if (slot < (uint32_t)NPS_EVCAP && q < (uint32_t)NPS_QCAP) {
          s_ev[slot] = s_q[q].pos | (s_q[q].f << 16);
          ev_out[slot].f = s_q[q].f;
          ev_out[slot].v = s_q[q].v;
        }
        ++slot;
      }
    nev += total;
    __syncthreads();
    piece = a_sub * piece + c_sub;
  }
  if (nev > NPS_EVCAP) { if (t == 0) atomicOr(A.overflow, 2u); nev = NPS_EVCAP; }
  if (t == 0) A.evcount[tile] = nev;
  __syncthreads();
  // transfer map: lane e walks the events from entry offset e and marks the ones that are starts on its path
  if (t < 64) {                         // (the whole first wave, in step: the ballot needs every lane at the same event)
    uint32_t cur = (uint32_t)min(t, NPS_K - 1), count = 0;
    for (uint32_t k = 0; k < nev; ++k) {
      const uint32_t pf = s_ev[k], p = pf & 0xffffu, f = pf >> 16;
      const bool on = p >= cur;         // else: inside an earlier slow normal, not a start
      const uint64_t b = __ballot(on);
      if (t == 0) s_mask[k] = (uint32_t)(b & ((1u << NPS_K) - 1u));
      if (on) {
        count += p - cur + 1;           // the fast starts before it, and this one
        cur = p + f;
      }
    }
    if (t < NPS_K) {
    uint32_t exit_off = 0;
    if (cur < (uint32_t)NPS_T) count += NPS_T - cur; else exit_off = cur - NPS_T;
    if (exit_off >= (uint32_t)NPS_K) { atomicOr(A.overflow, 4u); exit_off = 0; }
    const bool same = (__ballot(exit_off == (uint32_t)__builtin_amdgcn_readfirstlane((int)exit_off)) & 0xffffull) == 0xffffull;
    A.maps[(size_t)tile * NPS_K + t] = count | (exit_off << 16) | (same ? 0x80000000u : 0u);      // (NPS_MAP_CONST: see the scan)
    }
  }
  __syncthreads();
  for (uint32_t k = t; k < nev; k += NPS_THREADS) ev_out[k].pos = (s_ev[k] & 0xffffu) | (s_mask[k] << 16);
}

// ---------------------------------------------------------------- scan (one workgroup)
// Entry offset and first-normal index of every tile.  A tile almost never remembers how it was entered (the paths from the K
// entry offsets merge at the first word that is a start on all of them), which the classifier records in bit 31 of its map: then
// the entry of tile j is the exit of tile j - 1 whatever came before, every load is independent of every other, and the indices
// are a prefix sum (the "fast" path: three rounds of memory latency).  If any tile's exit depends on its entry, or the segment is
// longer than the LDS holds, the maps are composed in order instead: each thread walks a contiguous run of tiles for all K
// entries at once, the runs are chained, and a second walk fills in the tiles (the "general" path: one latency per tile of a run,
// twice).  Both are exact; NPS_SCAN_GENERAL in A.flags forces the second (tests).
constexpr int NPS_SCAN_THREADS = 1024;
constexpr int NPS_SCAN_CAP = 16384;                    // tiles of the fast path (134 M words: a full-size chunk has 12.9 k)
constexpr uint32_t NPS_SCAN_GENERAL = 1u;
constexpr uint32_t NPS_MAP_CONST = 0x80000000u;        // map bit: the tile's exit offset is the same for all K entries
__device__ __forceinline__ uint32_t nps_map_exit(uint32_t m) { return (m >> 16) & 0xffu; }

__global__ __launch_bounds__(NPS_SCAN_THREADS) void k_nps_scan(NpsSegArgs A) {
  constexpr int GENERAL_BYTES = NPS_K * NPS_SCAN_THREADS * 5, FAST_BYTES = NPS_SCAN_CAP * 3;
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[GENERAL_BYTES > FAST_BYTES ? GENERAL_BYTES : FAST_BYTES];
  __shared__ int64_t s_end_tile;
  __shared__ uint32_t s_ev[NPS_EVCAP];
  __shared__ NpsLdsJump s_jump;
  __shared__ uint32_t s_entry[NPS_SCAN_THREADS];
  __shared__ unsigned long long s_base[NPS_SCAN_THREADS];
  __shared__ int s_general;
  const int t = threadIdx.x;
  nps_load_jump(&s_jump, A.jump, A.inc);
  const int64_t per = (A.ntiles + NPS_SCAN_THREADS - 1) / NPS_SCAN_THREADS;
  const int64_t j0 = min((int64_t)t * per, A.ntiles), j1 = min(j0 + per, A.ntiles);
  const bool general0 = A.ntiles > NPS_SCAN_CAP || (A.flags & NPS_SCAN_GENERAL);       // (uniform)
  if (t == 0) { s_end_tile = -1; s_general = general0 ? 1 : 0; }
  __syncthreads();
  if (!general0) {
    uint8_t* s_x = s_raw;                                              // exit offset of tile j (then, after it: normals started in tile j)
    const int nt = (int)A.ntiles;
    bool all_const = true;
    for (int j = t; j < nt; j += 4 * NPS_SCAN_THREADS) {               // four independent loads in flight per lane
      uint32_t m[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int ju = j + u * NPS_SCAN_THREADS; m[u] = ju < nt ? A.maps[(size_t)ju * NPS_K] : NPS_MAP_CONST; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ju = j + u * NPS_SCAN_THREADS;
        if (ju < nt) s_x[ju] = (uint8_t)nps_map_exit(m[u]);
        all_const = all_const && (m[u] & NPS_MAP_CONST);
      }
    }
    if (!all_const) s_general = 1;
    __syncthreads();
  }
  if (!s_general) {
    uint8_t* s_x = s_raw;
    uint16_t* s_c = reinterpret_cast<uint16_t*>(s_raw + NPS_SCAN_CAP);
    const int nt = (int)A.ntiles;
    for (int j = t; j < nt; j += 4 * NPS_SCAN_THREADS) {
      uint32_t m[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ju = j + u * NPS_SCAN_THREADS;
        m[u] = ju < nt ? A.maps[(size_t)ju * NPS_K + (ju ? s_x[ju - 1] : 0u)] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int ju = j + u * NPS_SCAN_THREADS; if (ju < nt) s_c[ju] = (uint16_t)(m[u] & 0xffffu); }
    }
    __syncthreads();
    uint32_t mine = 0;
    for (int64_t j = j0; j < j1; ++j) mine += s_c[j];
    // exclusive prefix over the 1024 threads (at most CAP * T = 2^27 normals: 32 bits)
    uint32_t* s_wave = s_entry;                                         // (free on this path)
    const uint32_t inc = nps_wave_scan(mine);
    if ((t & 63) == 63) s_wave[t >> 6] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < NPS_SCAN_THREADS / 64; ++w) { const uint32_t v = s_wave[w]; if (w < (t >> 6)) before += v; total += v; }
    if (t == 0 && total < A.n) atomicOr(A.overflow, 8u);                // the launch did not cover n normals
    uint64_t base = before + inc - mine;
    for (int64_t j = j0; j < j1; ++j) {
      const uint64_t c = s_c[j];
      A.tile_e[j] = j ? s_x[j - 1] : (uint8_t)0;
      A.tile_base[j] = base;
      if (base < A.n && A.n <= base + c) s_end_tile = j;               // the tile the n-th normal starts in (exactly one)
      base += c;
    }
  } else {
    // maps of contiguous runs of tiles, one per thread: normals started and exit offset for each of the K entry offsets
    auto s_cnt = reinterpret_cast<uint32_t(*)[NPS_SCAN_THREADS]>(s_raw);                                    // [entry offset][run]
    auto s_exit = reinterpret_cast<uint8_t(*)[NPS_SCAN_THREADS]>(s_raw + NPS_K * NPS_SCAN_THREADS * 4);
    __syncthreads();
    {
      // sixteen independent chains (one per entry offset) through this thread's tiles: their loads overlap
      uint32_t cur[NPS_K], cnt[NPS_K];
#pragma unroll
      for (int e = 0; e < NPS_K; ++e) { cur[e] = (uint32_t)e; cnt[e] = 0; }
      for (int64_t j = j0; j < j1; ++j) {
        const uint32_t* mj = A.maps + (size_t)j * NPS_K;
#pragma unroll
        for (int e = 0; e < NPS_K; ++e) {
          const uint32_t m = mj[cur[e]];
          cnt[e] += m & 0xffffu;
          cur[e] = nps_map_exit(m);
        }
      }
#pragma unroll
      for (int e = 0; e < NPS_K; ++e) { s_cnt[e][t] = cnt[e]; s_exit[e][t] = (uint8_t)cur[e]; }
    }
    __syncthreads();
    if (t == 0) {                          // chain the runs (1024 steps in the LDS)
      uint32_t e = 0;
      unsigned long long base = 0;
      for (int r = 0; r < NPS_SCAN_THREADS; ++r) {
        s_entry[r] = e;
        base += s_cnt[e][r];
        s_base[r] = base;
        e = s_exit[e][r];
      }
      if (base < A.n) atomicOr(A.overflow, 8u);
    }
    __syncthreads();
    uint32_t cur = s_entry[t];
    uint64_t base = t ? s_base[t - 1] : 0ull;
    for (int64_t j = j0; j < j1; ++j) {
      const uint32_t m = A.maps[(size_t)j * NPS_K + cur];
      A.tile_e[j] = (uint8_t)cur;
      A.tile_base[j] = base;
      const uint64_t c = m & 0xffffu;
      if (base < A.n && A.n <= base + c) s_end_tile = j;
      base += c;
      cur = nps_map_exit(m);
    }
  }
  __syncthreads();
  // the word after the n-th normal
  const int64_t je = s_end_tile;
  if (je < 0) {
    if (t == 0) { *A.consumed = 0; *A.state_out = *A.state; if (A.n) atomicOr(A.overflow, 8u); }
    return;
  }
  const uint32_t nev = A.evcount[je];
  for (uint32_t k = t; k < nev; k += NPS_SCAN_THREADS) {
    const NpsEvent e = A.events[(size_t)je * NPS_EVCAP + k];
    s_ev[k] = (e.pos & 0xffffu) | (e.f << 16);
  }
  __syncthreads();
  if (t == 0) {
    const uint64_t want = A.n - A.tile_base[je];        // the want-th start of the tile (1-based)
    uint64_t cur = A.tile_e[je], count = 0, end = 0;
    bool done = false;
    for (uint32_t k = 0; k < nev && !done; ++k) {
      const uint64_t p = s_ev[k] & 0xffffu, f = s_ev[k] >> 16;
      if (p < cur) continue;
      if (count + (p - cur) >= want) { end = cur + (want - count); done = true; break; }      // a fast start
      count += p - cur + 1;
      cur = p + f;
      if (count == want) { end = cur; done = true; }
    }
    if (!done) end = cur + (want - count);
    const uint64_t consumed = (uint64_t)je * NPS_T + end;
    *A.consumed = consumed;
    *A.state_out = nps_advance_lds(*A.state, consumed, &s_jump);
  }
}

// ---------------------------------------------------------------- emit: normals [lo, hi) of the segment -> out[idx - lo]
struct NpsEmitRange {      // normals [lo, hi) -> out[0 ... hi - lo), looked for in tiles [tile0, tile0 + ntiles)
  int64_t tile0, ntiles;
  uint64_t lo, hi;
  double* out;
};
__global__ __launch_bounds__(NPS_THREADS) void k_nps_emit(NpsSegArgs A, NpsEmitRange R0, NpsEmitRange R1) {
  // two ranges per launch (the real and the imaginary parts of a batch): twice the workgroups in flight
  const bool second = (int64_t)blockIdx.x >= R0.ntiles;
  const NpsEmitRange R = second ? R1 : R0;
  const int64_t tile0 = R.tile0 - (second ? R0.ntiles : 0);
  const uint64_t lo = R.lo, hi = R.hi;
  double* out = R.out;
  __shared__ double s_wi[256];
  __shared__ uint64_t s_ki[256];
  __shared__ uint32_t s_skip[NPS_T / 32];     // bit p: word p is not the start of a normal
  __shared__ double s_out[NPS_SUB];           // the normals of one sub-tile, in order
  __shared__ uint32_t s_wave[4];
  __shared__ NpsLdsJump s_jump;
  const int64_t tile = tile0 + blockIdx.x;
  if (tile >= A.ntiles) return;
  const uint64_t base = A.tile_base[tile];
  const uint32_t e_in = A.tile_e[tile];
  const uint32_t m = A.maps[(size_t)tile * NPS_K + e_in];
  if (base >= hi || base + (m & 0xffffu) <= lo) return;          // no normal of [lo, hi) starts here (uniform over the workgroup)
  const int t = threadIdx.x;
  s_wi[t] = A.tab->wi[t]; s_ki[t] = A.tab->ki[t];
  for (int i = t; i < NPS_T / 32; i += NPS_THREADS) s_skip[i] = (i == 0) ? ((1u << e_in) - 1u) : 0u;      // the words before the entry offset
  nps_load_jump(&s_jump, A.jump, A.inc);
  __syncthreads();
  // the words inside the slow normals of the true path (classify marked which events start on it)
  const uint32_t nev = A.evcount[tile];
  const NpsEvent* ev = A.events + (size_t)tile * NPS_EVCAP;
  for (uint32_t k = t; k < nev; k += NPS_THREADS) {
    const uint32_t pm = ev[k].pos, f = ev[k].f, p = pm & 0xffffu;
    if ((pm >> (16 + e_in)) & 1u)
      for (uint32_t q = p + 1; q < p + f && q < (uint32_t)NPS_T; ++q) atomicOr(&s_skip[q >> 5], 1u << (q & 31));
  }
  __syncthreads();
  u128 piece = nps_advance_lds(A.tile_state[tile], (uint64_t)t * NPS_WPT, &s_jump);
  const u128 a_sub = s_jump.a[11], c_sub = s_jump.cinc[11];
  uint64_t rank0 = base;         // index of the first normal of the current sub-tile
  uint32_t nev_before = 0;       // events of the earlier sub-tiles
  static_assert(NPS_NSUB % 2 == 0, "sub-tiles are stepped in pairs");
  for (int sub = 0; sub < NPS_NSUB; sub += 2) {
    // TWO sub-tiles per pass: two independent generator chains per lane (a PCG64 step is a dependent chain of a dozen
    // quarter-rate multiplies; with three or four waves per SIMD one chain per lane leaves the VALU idle two cycles in three)
    double val[2][NPS_WPT];
    uint32_t slowmask[2] = {0, 0};
    u128 sc[2];
    const u128 piece_b = a_sub * piece + c_sub;
    sc[0] = piece;
    sc[1] = piece_b;
#pragma unroll
    for (int i = 0; i < NPS_WPT; ++i) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint64_t r = nps_next(sc[c], A.inc);
        const int idx = (int)(r & 0xff);
        uint64_t rabs;
        val[c][i] = nps_fast_value(r, s_wi, rabs);
        if (rabs >= s_ki[idx]) slowmask[c] |= 1u << i;
      }
    }
    piece = a_sub * piece_b + c_sub;     // this thread's piece of sub-tile sub + 2
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint32_t p0 = (uint32_t)((sub + c) * NPS_SUB + t * NPS_WPT);
      const uint32_t startmask = ~(s_skip[p0 >> 5] >> (p0 & 31)) & 0xffu;        // eight consecutive bits of one word (p0 is a multiple of 8)
      const uint32_t nslow = __popc(slowmask[c]), nstart = __popc(startmask);
      // ONE scan for both counts (each sums to at most 2048 per sub-tile: 16 bits apiece)
      uint32_t tot;
      const uint32_t ex = nps_block_scan(nslow | (nstart << 16), s_wave, tot);
      const uint32_t tot_slow = tot & 0xffffu, tot_start = tot >> 16;
      // the sub-tile's normals are one contiguous run of the output (rank0 ...): gathered in the LDS in order, written coalesced
      uint32_t ks = nev_before + (ex & 0xffffu), o = ex >> 16;
#pragma unroll
      for (int i = 0; i < NPS_WPT; ++i) {
        const bool slow = (slowmask[c] >> i) & 1u;
        if ((startmask >> i) & 1u) {
          s_out[o] = slow ? ev[min(ks, (uint32_t)NPS_EVCAP - 1)].v : val[c][i];
          ++o;
        }
        if (slow) ++ks;
      }
      __syncthreads();
      for (uint32_t k = t; k < tot_start; k += NPS_THREADS) {
        const uint64_t g = rank0 + k;
        if (g >= lo && g < hi) out[g - lo] = s_out[k];
      }
      nev_before += tot_slow;
      rank0 += tot_start;
    }
  }
}


// ---------------------------------------------------------------- one pass: classify, chain and write a segment in ONE kernel
// The three passes above generate every word twice (classify, emit: ~0.4 ms each per 2 x 10^8 words, the price of 128-bit LCG steps
// at quarter rate).  When the whole segment fits a device buffer the words are generated ONCE: a workgroup keeps its tile's fast
// values in registers (32 per lane), classifies as above, learns its entry offset and the index of its first normal from its
// predecessors while they are still running (a decoupled look-back, Merrill & Garland 2016), and writes its normals.
//   * tiles are handed out by a ticket in dispatch order, so a workgroup only ever waits for workgroups that already run;
//   * entry offset: a tile whose exit offset is the same for all K entries (NPS_MAP_CONST: virtually all) publishes it straight
//     after its map; a tile that does depend on its entry waits for its predecessor's first -- a chain only through such tiles;
//   * index: with the entry known the tile's count is a number, and the index of its first normal a plain prefix sum:
//     status << 62 | value per tile in one 64-bit word (1: the tile's count, 2: the sum up to and including the tile), looked
//     back over 64 predecessors per step by the first wave.
// Same overflow flags as the three-pass form, same results bit for bit (tests run both).
struct NpsOneArgs {
  double* out;                    // [n] the segment's normals
  uint32_t* xexit;                // [ntiles] bit 31: published; low byte: the tile's exit offset on the stream's path
  unsigned long long* agg;        // [ntiles] status << 62 | normals
  uint32_t* ticket;
};
constexpr unsigned long long NPS_AGG_MASK = (1ull << 62) - 1ull;
constexpr uint32_t NPS1_SPIN_LIMIT = 1u << 20;        // polls of a look-back wait before the tile gives up (each a memory round trip: ~1 s)

// tile states, and the look-back words cleared (one thread per tile)
template <int NSUB>
__global__ __launch_bounds__(NPS_THREADS) void k_nps_tilestates1(NpsSegArgs A, NpsOneArgs O) {
  constexpr int TW = NSUB * NPS_SUB;
  __shared__ NpsLdsJump s_jump;
  nps_load_jump(&s_jump, A.jump, A.inc);
  __syncthreads();
  const int64_t tile = (int64_t)blockIdx.x * NPS_THREADS + threadIdx.x;
  if (tile == 0) *O.ticket = 0u;
  if (tile < A.ntiles) {
    A.tile_state[tile] = nps_advance_lds(*A.state, (uint64_t)tile * TW, &s_jump);
    O.xexit[tile] = 0u;
    O.agg[tile] = 0ull;
  }
}

__device__ __forceinline__ unsigned long long nps_wave_sum64(unsigned long long v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// Five workgroups per CU (96 registers -- five values spill -- and 30 KB of LDS): between generating its words and writing
// them a tile mostly WAITS (its predecessors' words, 1 - 2 us per round trip through memory), and only other tiles fill the VALU.
#ifndef NPS1_WAVES
#define NPS1_WAVES 5
#endif
#if NPS1_WAVES
#define NPS1_OCC __attribute__((amdgpu_waves_per_eu(NPS1_WAVES, NPS1_WAVES)))
#else
#define NPS1_OCC
#endif
#ifndef NPS1_QCAP
#define NPS1_QCAP 192
#endif
template <int NSUB>      // tile = NSUB sub-tiles of 2048 words: 8 values per lane and sub-tile stay in registers
__global__ __launch_bounds__(NPS_THREADS) NPS1_OCC void k_nps_onepass(NpsSegArgs A, NpsOneArgs O) {
  static_assert(NSUB % 2 == 0 && NPS_WPT == 8, "passes of two sub-tiles; eight start bits of a lane lie in one skip word");
  constexpr int TW = NSUB * NPS_SUB, EVCAP1 = TW / 16;
  constexpr int QCAP = NPS1_QCAP;                   // slow words of TWO sub-tiles (expected 90, sigma 9.4)
  __shared__ double s_wi[256];
  __shared__ uint64_t s_ki[256];
  __shared__ double s_fi[256];
  __shared__ uint32_t s_ev[EVCAP1];          // pos | f << 16, in position order
  __shared__ uint32_t s_mask[EVCAP1];        // bit e: a START on the path that enters at offset e
  __shared__ double s_evv[EVCAP1];           // the slow normal's value
  // the slow-word queue (while classifying) and the output staging (while writing) share their bytes
  constexpr int Q_BYTES = QCAP * (int)sizeof(NpsQueued), QIDX_BYTES = 2 * NPS_SUB * 2, OUT_BYTES = NPS_SUB * 8;
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[(Q_BYTES + QIDX_BYTES) > OUT_BYTES ? (Q_BYTES + QIDX_BYTES) : OUT_BYTES];
  NpsQueued* s_q = reinterpret_cast<NpsQueued*>(s_raw);
  uint16_t* s_qidx = reinterpret_cast<uint16_t*>(s_raw + Q_BYTES);        // [2][NPS_SUB]
  double* s_out = reinterpret_cast<double*>(s_raw);
  __shared__ uint32_t s_skip[TW / 32];
  __shared__ uint32_t s_wave[4];
  __shared__ NpsLdsJump s_jump;
  __shared__ uint32_t s_qn[2];
  __shared__ uint32_t s_tile, s_ein, s_cnt;
  __shared__ unsigned long long s_base;
  const int t = threadIdx.x;
  if (t == 0) { s_tile = atomicAdd(O.ticket, 1u); s_qn[0] = s_qn[1] = 0; }
  s_wi[t] = A.tab->wi[t]; s_ki[t] = A.tab->ki[t]; s_fi[t] = A.tab->fi[t];
  nps_load_jump(&s_jump, A.jump, A.inc);
  __syncthreads();
  const int64_t tile = s_tile;
  u128 piece = nps_advance_lds(A.tile_state[tile], (uint64_t)t * NPS_WPT, &s_jump);
  const u128 a_sub = s_jump.a[11], c_sub = s_jump.cinc[11];
  static_assert(NPS_SUB == 2048, "the sub-tile stride is the 2^11 entry of the jump table");
#ifdef NPS1_EXP_TIMES
  const uint64_t T0 = wall_clock64();
  uint64_t T1 = 0, T2 = 0, T3 = 0;
  __shared__ uint64_t s_t1w;
#endif

  // ---- the words, once: fast values kept, slow words evaluated through the queue (two sub-tiles per pass: two chains per lane)
  double val[NSUB][NPS_WPT];
  uint32_t slowmask[NSUB];
  uint32_t nev = 0;
#pragma unroll
  for (int pass = 0; pass < NSUB / 2; ++pass) {
    u128 sc[2];
    sc[0] = piece;
    sc[1] = a_sub * piece + c_sub;
    piece = a_sub * sc[1] + c_sub;
    slowmask[2 * pass] = slowmask[2 * pass + 1] = 0;
#pragma unroll
    for (int i = 0; i < NPS_WPT; ++i) {
      // both chains' steps in ONE basic block (their dependent multiplies interleave), one branch for the rare pushes
      uint64_t r[2], rabs[2];
      bool slow[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) r[c] = nps_next(sc[c], A.inc);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        val[2 * pass + c][i] = nps_fast_value(r[c], s_wi, rabs[c]);
        asm volatile("" : "+v"(val[2 * pass + c][i]));      // the VALUE stays (else the compiler keeps what it is made of: 255 registers)
        slow[c] = rabs[c] >= s_ki[(int)(r[c] & 0xff)];
      }
      if (slow[0] | slow[1]) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (slow[c]) {
            slowmask[2 * pass + c] |= 1u << i;
            const uint32_t q = atomicAdd(&s_qn[pass & 1], 1u);
            if (q < (uint32_t)QCAP) {
              s_q[q].s = sc[c]; s_q[q].r = r[c]; s_q[q].pos = (uint32_t)((2 * pass + c) * NPS_SUB + t * NPS_WPT + i);
              s_qidx[c * NPS_SUB + t * NPS_WPT + i] = (uint16_t)q;
            }
          }
      }
    }
    const uint32_t na = __popc(slowmask[2 * pass]), nb = __popc(slowmask[2 * pass + 1]);
    uint32_t tot;
    const uint32_t ex = nps_block_scan(na | (nb << 16), s_wave, tot);         // (its barriers also publish the queue)
    const uint32_t tot_a = tot & 0xffffu, tot_b = tot >> 16;
    const uint32_t qn = min(s_qn[pass & 1], (uint32_t)QCAP);
    if (t == 0) {
      if (s_qn[pass & 1] > (uint32_t)QCAP) atomicOr(A.overflow, 2u);
      s_qn[(pass + 1) & 1] = 0;
    }
    if ((uint32_t)t < qn) {
      uint32_t f;
      s_q[t].v = nps_slow(s_q[t].r, s_q[t].s, A.inc, s_wi, s_ki, s_fi, f);
      s_q[t].f = f;
      if (f >= 0xffffu) atomicOr(A.overflow, 1u);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      uint32_t slot = nev + (c ? tot_a + (ex >> 16) : (ex & 0xffffu));
#pragma unroll
      for (int i = 0; i < NPS_WPT; ++i)
        if ((slowmask[2 * pass + c] >> i) & 1u) {
          const uint32_t q = s_qidx[c * NPS_SUB + t * NPS_WPT + i];
          if (slot < (uint32_t)EVCAP1 && q < (uint32_t)QCAP) {
            s_ev[slot] = s_q[q].pos | (s_q[q].f << 16);
            s_evv[slot] = s_q[q].v;
          }
          ++slot;
        }
    }
    nev += tot_a + tot_b;
    __syncthreads();
  }
  if (nev > EVCAP1) { if (t == 0) atomicOr(A.overflow, 2u); nev = EVCAP1; }

  // ---- first wave: the tile's transfer map, then its place in the stream from its predecessors
  if (t < 64) {
#ifdef NPS1_EXP_TIMES
    T1 = wall_clock64();
#endif
    // Lane e walks the events from entry offset e (lanes 16 ... 63 repeat lane 15) until the K paths have merged -- at the first
    // word that is a start on all of them, within an event or two.  From there the walk is ONE path, and no longer serial: an
    // event at p is a start unless the last start's span reaches beyond p, which the running maximum M of ALL earlier spans
    // decides for nearly every event at once (M <= p: a start whatever came before); the few with M > p (a slow word within a
    // slow normal's span: 2 %) are settled in order on the scalar unit.  (A serial walk costs ~20 dependent instructions per
    // event on a lone wave: 13 us per tile, as long as generating it.)
    uint32_t cur = (uint32_t)min(t, NPS_K - 1), count = 0;
    uint32_t km = 0;
    bool merged = false;
    for (; km < nev && !merged; ++km) {
      const uint32_t pf = s_ev[km], p = pf & 0xffffu, f = pf >> 16;
      const bool on = p >= cur;
      const uint64_t b = __ballot(on);
      if (t == 0) s_mask[km] = (uint32_t)(b & ((1u << NPS_K) - 1u));
      if (on) { count += p - cur + 1; cur = p + f; }
      merged = __ballot(cur == (uint32_t)__builtin_amdgcn_readfirstlane((int)cur)) == ~0ull;
    }
    if (merged) {
      const uint32_t cur0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
      uint32_t R = cur0;              // where the path stands: the end of the last start's span
      uint32_t skipped = 0;           // words of the tile inside the spans of the starts (not the starts themselves)
      for (uint32_t k0 = km; k0 < nev; k0 += 64) {
        const uint32_t k = k0 + t;
        const bool valid = k < nev;
        const uint32_t pf = valid ? s_ev[k] : 0xffffffffu, p = pf & 0xffffu, reach = p + (pf >> 16);
        uint32_t incl = reach;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
          if (t >= d) incl = max(incl, o);
        }
        uint32_t M = (uint32_t)__shfl_up((int)incl, 1, 64);
        M = t ? max(M, R) : R;
        uint64_t onm = __ballot(valid && M <= p), amb = __ballot(valid && M > p);
        while (amb) {
          const int ka = __builtin_ctzll(amb);
          amb &= amb - 1;
          const uint64_t below = onm & ((1ull << ka) - 1ull);
          const uint32_t Rk = below ? (uint32_t)__builtin_amdgcn_readlane((int)reach, 63 - __builtin_clzll(below)) : R;
          if (Rk <= (uint32_t)__builtin_amdgcn_readlane((int)p, ka)) onm |= 1ull << ka;
        }
        const bool on = (onm >> t) & 1ull;
        if (valid) s_mask[k] = on ? ((1u << NPS_K) - 1u) : 0u;
        uint32_t sk = on ? min(reach, (uint32_t)TW) - p - 1u : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sk += (uint32_t)__shfl_xor((int)sk, d, 64);
        skipped += sk;
        if (onm) R = (uint32_t)__builtin_amdgcn_readlane((int)reach, 63 - __builtin_clzll(onm));
      }
      if (cur0 < (uint32_t)TW) count += (uint32_t)TW - cur0 - skipped;
      cur = max(R, (uint32_t)TW);       // (every word up to the end of the tile accounted for)
    }
    uint32_t exit_off = 0;
    if (cur < (uint32_t)TW) count += TW - cur; else exit_off = cur - TW;
    if (exit_off >= (uint32_t)NPS_K) { atomicOr(A.overflow, 4u); exit_off = 0; }
#if defined(NPS1_EXP_TIMES) && NPS1_EXP_TIMES == 2
    if (t == 0) s_t1w = wall_clock64();
#endif
    const bool same = __ballot(exit_off == (uint32_t)__builtin_amdgcn_readfirstlane((int)exit_off)) == ~0ull;
    uint32_t e_in = 0;
    if (tile > 0) {
      if (same && t == 0) __hip_atomic_store(&O.xexit[tile], 0x80000000u | exit_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (Every wait below is bounded: a predecessor that never publishes -- a fault, not a state this protocol reaches -- must
      // not hang the device.  After NPS1_SPIN_LIMIT polls, ~1 s, the tile raises overflow flag 32, goes on with what it has and
      // publishes, so that its successors end too; the caller redoes the chunk with numpy as for every other flag.)
      uint32_t x, spins = 0;
      for (;;) {
        x = __hip_atomic_load(&O.xexit[tile - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
        if (x >> 31) break;
        if (++spins > NPS1_SPIN_LIMIT) { if (t == 0) atomicOr(A.overflow, 32u); x = 0x80000000u; break; }
        __builtin_amdgcn_s_sleep(4);
      }
      e_in = x & 0xffu;
    }
#ifdef NPS1_EXP_TIMES
    T2 = wall_clock64();
#endif
    const uint32_t c_here = (uint32_t)__shfl((int)count, (int)e_in, 64);
    const uint32_t x_here = (uint32_t)__shfl((int)exit_off, (int)e_in, 64);
    if ((tile == 0 || !same) && t == 0) __hip_atomic_store(&O.xexit[tile], 0x80000000u | x_here, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long base = 0;
    if (tile > 0) {
      if (t == 0) __hip_atomic_store(&O.agg[tile], (1ull << 62) | c_here, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int64_t p = tile - 1;
      uint32_t spins = 0;
      for (;;) {
        if (++spins > NPS1_SPIN_LIMIT) { if (t == 0) atomicOr(A.overflow, 32u); break; }
        const int64_t idx = p - t;
        const unsigned long long v = idx >= 0 ? __hip_atomic_load(&O.agg[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
        const uint32_t st = (uint32_t)(v >> 62);
        const uint64_t pm = __ballot(st == 2u), zm = __ballot(st == 0u);
        if (pm) {
          const int first = __builtin_ctzll(pm);                                   // the nearest predecessor that knows its prefix
          const uint64_t upto = first == 63 ? ~0ull : ((1ull << (first + 1)) - 1ull);
          if (zm & upto) { __builtin_amdgcn_s_sleep(2); continue; }
          base += nps_wave_sum64(t <= first ? (v & NPS_AGG_MASK) : 0ull);
          break;
        }
        if (zm) { __builtin_amdgcn_s_sleep(2); continue; }
        base += nps_wave_sum64(v & NPS_AGG_MASK);
        p -= 64;
      }
    }
    if (t == 0) {
      __hip_atomic_store(&O.agg[tile], (2ull << 62) | (base + c_here), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_ein = e_in; s_cnt = c_here; s_base = base;
    }
#ifdef NPS1_EXP_TIMES
    T3 = wall_clock64();
#endif
  }
  for (int i = t; i < TW / 32; i += NPS_THREADS) s_skip[i] = 0u;
  __syncthreads();
  const uint32_t e_in = s_ein, cnt_tile = s_cnt;
  const uint64_t base = s_base;
  if (tile == 0 && A.n == 0 && t == 0) { *A.consumed = 0; *A.state_out = *A.state; }
  if (tile == A.ntiles - 1 && t == 0 && base + cnt_tile < A.n) atomicOr(A.overflow, 8u);          // the launch did not cover n normals
  if (base >= A.n) return;                                                                         // (uniform) beyond the segment

  // ---- the words that are not starts: before the entry offset, and inside the slow normals of the stream's path
  if (t == 0 && e_in) atomicOr(&s_skip[0], (1u << e_in) - 1u);
  for (uint32_t k = t; k < nev; k += NPS_THREADS) {
    const uint32_t pf = s_ev[k], p = pf & 0xffffu, f = pf >> 16;
    if ((s_mask[k] >> e_in) & 1u)
      for (uint32_t q = p + 1; q < p + f && q < (uint32_t)TW; ++q) atomicOr(&s_skip[q >> 5], 1u << (q & 31));
  }
  __syncthreads();

  // ---- write: each sub-tile's normals are one contiguous run of the output, gathered in the LDS in order, stored coalesced
  uint64_t rank0 = base;
  uint32_t nev_before = 0;
#pragma unroll
  for (int c = 0; c < NSUB; ++c) {
    const uint32_t p0 = (uint32_t)(c * NPS_SUB + t * NPS_WPT);
    const uint32_t startmask = ~(s_skip[p0 >> 5] >> (p0 & 31)) & 0xffu;
    const uint32_t nslow = __popc(slowmask[c]), nstart = __popc(startmask);
    uint32_t tot;
    const uint32_t ex = nps_block_scan(nslow | (nstart << 16), s_wave, tot);
    const uint32_t tot_slow = tot & 0xffffu, tot_start = tot >> 16;
    uint32_t ks = nev_before + (ex & 0xffffu), o = ex >> 16;
#pragma unroll
    for (int i = 0; i < NPS_WPT; ++i) {
      const bool slow = (slowmask[c] >> i) & 1u;
      if ((startmask >> i) & 1u) {
        s_out[o] = slow ? s_evv[min(ks, (uint32_t)EVCAP1 - 1)] : val[c][i];
        ++o;
      }
      if (slow) ++ks;
    }
    __syncthreads();
    for (uint32_t k = t; k < tot_start; k += NPS_THREADS) {
      const uint64_t g = rank0 + k;
      if (g < A.n) O.out[g] = s_out[k];
    }
    nev_before += tot_slow;
    rank0 += tot_start;
  }

#ifdef NPS1_EXP_TIMES
  __syncthreads();
  if (t == 0) {          // start | four 24-bit deltas (100 MHz ticks): generation done, entry offset known, index known, written
    const uint64_t T4 = wall_clock64();
#if NPS1_EXP_TIMES == 2     // generation done, walk done, entry offset known, index known
    const uint64_t d1 = (T1 - T0) & 0xffffff, d2 = (s_t1w - T0) & 0xffffff, d3 = (T2 - T0) & 0xffffff, d4 = (T3 - T0) & 0xffffff;
    (void)T4;
#else
    const uint64_t d1 = (T1 - T0) & 0xffffff, d2 = (T2 - T0) & 0xffffff, d3 = (T3 - T0) & 0xffffff, d4 = (T4 - T0) & 0xffffff;
#endif
    const u128 rec = (u128)(uint32_t)T0 | ((u128)d1 << 32) | ((u128)d2 << 56) | ((u128)d3 << 80) | ((u128)d4 << 104);
    A.tile_state[tile] = rec;
  }
#endif
  // ---- the tile the n-th normal starts in: the word after it = what the segment consumed, and the generator there
  if (t == 0 && base < A.n && A.n <= base + cnt_tile) {
    const uint64_t want = A.n - base;
    uint64_t cur = e_in, count = 0, end = 0;
    bool done = false;
    for (uint32_t k = 0; k < nev && !done; ++k) {
      const uint64_t p = s_ev[k] & 0xffffu, f = s_ev[k] >> 16;
      if (p < cur) continue;
      if (count + (p - cur) >= want) { end = cur + (want - count); done = true; break; }
      count += p - cur + 1;
      cur = p + f;
      if (count == want) { end = cur; done = true; }
    }
    if (!done) end = cur + (want - count);
    const uint64_t consumed = (uint64_t)tile * TW + end;
    *A.consumed = consumed;
    *A.state_out = nps_advance_lds(*A.state, consumed, &s_jump);
  }
}

}  // namespace fmc
