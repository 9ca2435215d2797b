// fastmc.hip -- C-ABI of libfastmc.so (include/fastmc.h): handle, device memory, launches.
// gfx950 only.  No CPU fallback anywhere: every compute entry point needs a device.
#include "../../include/fastmc.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#ifndef FMC_TU
#define FMC_TU 0     // see "Translation units" below
#endif
#include "fmc_kernels.h"
#ifndef FMC_PKS_P16_FROM
#define FMC_PKS_P16_FROM 2048         // (2048 +7 %, 4096 +15 %; 1024 +0.4 %: its dense sixteen-wave row stays -- profiles/r06_ab_packed_subrows.txt section 8)
#endif
#if FMC_TU == 0
#include "fmc_powerspec.h"
#if FMC_TU == 0
#include "fmc_npstream.h"
#endif
#endif

using namespace fmc;

// Translation units.  Alone (FMC_TU undefined) this file is the whole library.  With -DFMC_SPLIT_BUILD the Makefile compiles
// it ten times in parallel and links the objects: unit 0 holds the C-ABI, every host function and the small kernels;
// the kernel families that run_impl launches are explicit instantiations of their dispatch templates in units 1 ... 9
// (FMC_UNITS below): a fresh build takes the time of the slowest unit instead of the sum.
#ifndef FMC_TU
#define FMC_TU 0
#endif

// ------------------------------------------------------------------ errors
#if FMC_TU == 0
thread_local std::string g_err;
#else
extern thread_local std::string g_err;
#endif
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess)                                                                \
      return fail(FASTMC_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__));      \
  } while (0)

// ------------------------------------------------------------------ context
struct TimedSpan {
  hipEvent_t a, b;
  int family;
};

// One of the two steps a handle can have in flight (fastmc_run_queued / fastmc_comm_gather*_queued / fastmc_queue_wait): its own
// timing events, its own pinned host landing buffers and a completion event, so that the kernels of step i + 1 are enqueued
// while step i's results are still on their way to the host -- the device never waits for the host between steps.
struct QueueSlot {
  std::vector<TimedSpan> spans;
  std::vector<hipEvent_t> pool;
  size_t pool_used = 0;
  hipEvent_t ex_a = nullptr, ex_b = nullptr;
  bool ex_recorded = false;
  bool pending = false;
  bool busy = false;              // enqueued and not waited for
  bool stalled = false;           // FASTMC_TEST_STALL_GATHER=1: the queued exchange "never completes" (fastmc_queue_wait blocks until an abort)
  int stall_gen = 0;              //   ... counted from the abort generation of the device when the exchange was queued
  hipEvent_t done = nullptr;      // after the last copy of the step
  double* pinned = nullptr;       // host landing buffer: the step's own vector, or the gathered vectors
  size_t pinned_cap = 0, landed = 0;   // doubles
  unsigned long long* pinned_hist = nullptr;
  size_t hist_cap = 0, hist_landed = 0;
};

struct fastmc_ctx {
  int device = 0, N = 0, Np = 0, lo = 0, precision = 0;
  int precision_req = 0;   // what fastmc_create was asked for (float32 is honoured on the wave family's grids only)
  int path = 0, P = 0, NS = 0, omS = 0;
  int S = 1;               // wave family: sub-rows per row (N = S * 64 * P); spec_split(N)
  int batch = 0;
  double df = 0, dx = 0, wsum = 0;
  bool have_spec = false, have_pupil = false, have_sh = false, have_ps = false;
  int tables_lo = -1;      // window position the wave tables were built for (-1: none)
  hipStream_t stream = nullptr;
  size_t rsz = 8;   // sizeof(R)

  void* amp = nullptr;     // R[N*N]  unsigned (direct family)
  void* amp_s = nullptr;   // R[N*N]  with (-1)^(ky+kx)  (wave family)
  void* amp_p = nullptr;   // grids of the packed sub-rows: amp_s / ampf_s with every row stored sub-row major (fmc_kernels.h: k_make_amp)
  float* ampf_p = nullptr;
  float* ampf = nullptr;   // amp / amp_s times sqrt(2 ln 2) in float32: colouring of the device generator's draws
  float* ampf_s = nullptr; //   (fmc_kernels.h: box_muller_scaled)
  unsigned int* bad = nullptr;   // count of invalid spectrum entries (k_make_amp)
  void* tw = nullptr;      // direct: cpx<R>[N]
  void* tw1 = nullptr;     // wave
  void* om = nullptr;      // wave
  void* cw = nullptr;      // wave, S > 1: [S][omS] combination twiddles
  // chirp-z family (path 2: grid sizes that are not 64 P): M = 64 * blu_P >= N + Np - 1
  bool no_dense = false;   // FASTMC_NO_DENSE16=1 in the environment at create: keep the twelve-wave kernels (A/B)
  int blu_P = 0;           // 0: not eligible
  int blu_SB = 1, blu_B = 0;   // input blocks per row (fmc_bluestein.h): grids whose rows exceed the largest M run SB blocks of B on M = 1024
  int blu_lo = -1;         // window position the tables below were built for
  void* blu_tw1 = nullptr; // tw1 of size M
  void* blu_om = nullptr;  // om for the window [0, Np) of the second transform
  void* blu_twf = nullptr;
  void* blu_pre = nullptr;
  void* blu_vhat = nullptr;
  void* blu_post = nullptr;
  void* pbz_tw = nullptr;   // chirp-z rows on the packed 256-point pipeline (fmc_bluestein.h: build_pbz_tables; windows of up to 128 pixels)
  void* pbz_pre = nullptr;
  void* pbz_vhat = nullptr;
  void* pbz_post = nullptr;
  // 50-lane family (path 3: N = 50 P, fmc_mrfft.h)
  int mr_P = 0;            // 0: not eligible (N not 50 P S, or no window instantiation)
  int mr_S = 1;            // sub-rows per row (mr_split(N))
  void* mr_cw = nullptr;   // S > 1: [S][omS] combination twiddles
  int mr_lo = -1;          // window position the tables below were built for
  void* mr_tw1 = nullptr;
  void* mr_om = nullptr;
  void* tw1g = nullptr;    // N = 2048 only: tables of the single-pass P = 32 kernels (windows > 256 pixels,
  void* omg = nullptr;     //   host coefficients: TEMPORAL layer screens, centred_fft2)
  void* pk_tw1 = nullptr;  // N = 256, 512: tables of the packed rows (fmc_wavefft.h: build_tw1_pk / build_om_pk)
  void* pk_om = nullptr;
  void* pks_tw1 = nullptr; // grids of the packed sub-rows (fmc_core.h: pks_count -- every multiple of 64 from 192 to 4096 but 256, 512): tables of the packed sub-rows (fmc_wavefft.h: build_tw1_pk for M = 256, build_pcw)
  void* pks_cw = nullptr;
  void* pks_cw16 = nullptr; // ... 256 entries per sub-row: all sixteen planes of 256-point sub-rows, centred windows of 129 ... 256 pixels
  void* pks_cw8 = nullptr;  // ... with 128 entries per sub-row: eight planes, centred windows of 97 ... 128 pixels (float64 pipeline)
  double* W = nullptr;
  void* V = nullptr;
  size_t V_cap = 0;        // realisations
  size_t V_bytes = 0;      // size of the slab behind V (may exceed V_cap realisations: taken from the cache)
  double* partial = nullptr;
  size_t partial_cap = 0;
  double* out = nullptr;   // device results of the last run
  size_t out_cap = 0;      // doubles
  int64_t last_n_iter = 0;
  int last_coherent = 0;
  double* logamp = nullptr;
  size_t logamp_cap = 0;
  double* cre = nullptr;
  double* cim = nullptr;
  size_t coef_cap = 0;     // realisations
  double* phs = nullptr;
  size_t phs_cap = 0;
  // sub-harmonics
  double* sh_scale = nullptr;  // [27]
  double* sh_mu = nullptr;     // [27][2]
  double* sh_ex = nullptr;     // [27][Np][2]
  double* sh_ey = nullptr;
  double* sh_coef = nullptr;   // [batch][27][2]
  double* sh_mean = nullptr;
  double* sh_dcol = nullptr;   // [batch][Np][9][2] column-folded coefficients (separable grids)
  bool sh_sep = false;         // fx independent of i and fy independent of j within every level
  double* sh_in_re = nullptr;  // host-mode coefficients [batch][27]
  double* sh_in_im = nullptr;
  size_t sh_cap = 0;
  double* ps_dev = nullptr;    // [3][N][N] powerspec, logamp powerspec, lf_mask left by fastmc_powerspec_set
  double* layers = nullptr;    // [n_layers][N][N] real layer screens (TEMPORAL mode)
  int n_layers = 0;
  unsigned long long* hist = nullptr;
  size_t hist_cap = 0;
  // timing
  std::vector<TimedSpan> spans;
  std::vector<hipEvent_t> pool;
  size_t pool_used = 0;
  double t_ms[4] = {0, 0, 0, 0};
  int64_t t_n[4] = {0, 0, 0, 0};
  // RCCL exchange buffer (the communicator itself belongs to the device, see DeviceComm)
  double* gather_buf = nullptr;
  size_t gather_cap = 0;
  // fastmc_run_async: kernels enqueued, events not read yet (fastmc_wait / the exchange finish the bookkeeping)
  bool light_rows = false; // fastmc_run_npstream on two streams: the MODE 1 rows as four-wave workgroups (WCfg D = 9)
  int rng_f64 = 0;        // fastmc_set_rng_precision: the device generator at float64 precision (fused into the P = 16 rows, else staged in cre / cim)
  const Gen64Entry* g64 = nullptr;   // its log table on this device (gen64_table)
  // ONE caller at a time on a handle's stream, events and result bookkeeping: the entry points that use them take this lock
  // (HandleLock), with a deadline, so that an exchange a deadline thread is still inside and the caller that gave up on it
  // (fastmc_comm_abort takes no lock) never run on the handle together, and neither can wait for ever
  std::timed_mutex use_mu;
  unsigned long long* clk = nullptr;   // device: clock stamps of the last k_rows_wave launch (fastmc_last_clock)
  bool clk_fresh = false;  // the LAST row launch of this handle was a stamping one (k_rows_wave): stamps of an earlier launch are never reported
  int comm_slot = -1;      // slot of the communicator tables; -1: the device's (the only form outside the fake-RCCL tests, cslot())
  char last_rows[96] = "", last_cols[96] = "";   // the row / column kernels of the last launch, as c++filt prints them (fastmc_last_kernels)
  bool pending = false;
  QueueSlot q[2];
  // the landing copies of a queued step run on a COPY stream behind an event of the compute stream, so that the next step's
  // kernels do not queue up behind them; whatever overwrites what a copy may still be reading (out, hist, gather_buf) first
  // waits for `copy_guard`, the completion event of the latest such copies
  hipStream_t cstream = nullptr;
  hipEvent_t copy_fence = nullptr;      // recorded on the compute stream before a slot's copies
  hipEvent_t copy_guard = nullptr;
  struct NpsWork* nps = nullptr;   // workspace of the numpy-stream generator (GPU_RNG 'numpy'; fmc_npstream.h), created on first use
  size_t last_out_doubles = 0;    // size of the last run's result vector in `out`
  hipEvent_t ex_a = nullptr, ex_b = nullptr;   // around the collectives of the last exchange
  bool ex_recorded = false;
  double ex_ms = 0;
};

static void cs_turns(double t, double* c, double* s) {
  // exact at multiples of 1/8 turn, else libm on the reduced angle
  const double a = 2.0 * M_PI * t;
  *c = std::cos(a);
  *s = std::sin(a);
}

template <class T>
static int dev_alloc(T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, n * sizeof(T)));
  return 0;
}
#define TRY(x)            \
  do {                    \
    int r__ = (x);        \
    if (r__ != 0) return r__; \
  } while (0)

struct ScratchBuf {   // device scratch freed on every return path
  double* p = nullptr;
  ~ScratchBuf() { if (p) hipFree(p); }
};

static int grow(double** p, size_t* cap, size_t need) {
  if (*cap >= need) return 0;
  if (*p) HIPCHK(hipFree(*p));
  *p = nullptr;
  *cap = 0;
  HIPCHK(hipMalloc((void**)p, need * sizeof(double)));
  *cap = need;
  return 0;
}

// before anything writes out / hist / gather_buf: the copies of a queued step that may still be reading them (see cstream)
static void guard_outputs(fastmc_ctx* h) {
  if (h->copy_guard) hipStreamWaitEvent(h->stream, h->copy_guard, 0);
}

static hipEvent_t next_event(fastmc_ctx* h) {
  if (h->pool_used == h->pool.size()) {
    hipEvent_t e;
    hipEventCreate(&e);
    h->pool.push_back(e);
  }
  return h->pool[h->pool_used++];
}
struct Span {
  fastmc_ctx* h;
  hipEvent_t a, b;
  int fam;
  Span(fastmc_ctx* h_, int fam_) : h(h_), fam(fam_) {
    a = next_event(h);
    b = next_event(h);
    hipEventRecord(a, h->stream);
  }
  ~Span() {
    hipEventRecord(b, h->stream);
    h->spans.push_back({a, b, fam});
  }
};
static void timing_begin(fastmc_ctx* h) {
  h->spans.clear();
  h->pool_used = 0;
  for (int i = 0; i < 4; ++i) { h->t_ms[i] = 0; h->t_n[i] = 0; }
}
static void timing_end(fastmc_ctx* h) {   // after stream sync
  for (auto& s : h->spans) {
    float ms = 0;
    hipEventElapsedTime(&ms, s.a, s.b);
    h->t_ms[1 + s.family] += ms;
    h->t_n[1 + s.family] += 1;
  }
  if (!h->spans.empty()) {
    float ms = 0;
    hipEventElapsedTime(&ms, h->spans.front().a, h->spans.back().b);
    h->t_ms[0] = ms;
    h->t_n[0] = (int64_t)h->spans.size();
  }
}

// after any synchronisation of the handle's stream: read the events of an asynchronous run / of the last exchange
static void finish_pending(fastmc_ctx* h) {
  if (h->pending) {
    timing_end(h);
    h->pending = false;
  }
  if (h->ex_recorded) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ex_a, h->ex_b) == hipSuccess) h->ex_ms = ms;
    h->ex_recorded = false;
  }
}

// The timing state of a queue slot takes the place of the handle's own for the length of a call (and back): the launch code
// (Span, timing_begin / timing_end, exchange_begin / exchange_end) never knows which step it is timing.
static void swap_slot(fastmc_ctx* h, QueueSlot& q) {
  std::swap(h->spans, q.spans);
  std::swap(h->pool, q.pool);
  std::swap(h->pool_used, q.pool_used);
  std::swap(h->ex_a, q.ex_a);
  std::swap(h->ex_b, q.ex_b);
  std::swap(h->ex_recorded, q.ex_recorded);
  std::swap(h->pending, q.pending);
}
struct SlotScope {
  fastmc_ctx* h;
  QueueSlot& q;
  SlotScope(fastmc_ctx* h_, QueueSlot& q_) : h(h_), q(q_) { swap_slot(h, q); }
  ~SlotScope() { swap_slot(h, q); }
};

// ------------------------------------------------------------------ misc entry points
#if FMC_TU == 0
extern "C" int fastmc_version(void) { return FASTMC_VERSION; }
#endif
#if FMC_TU == 0
extern "C" const char* fastmc_last_error(void) { return g_err.c_str(); }
#endif
#if FMC_TU == 0
extern "C" int fastmc_device_count(int* n) {
  if (!n) return fail(FASTMC_EINVAL, "n is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { *n = 0; return fail(FASTMC_ENODEV, hipGetErrorString(e)); }
  *n = c;
  return 0;
}
#endif

// N = 64 P with P = 2^k times 1, 3, 5, 7 or 9, 2 <= P <= 32
static bool wave_supported(int N) {
  if (N % 64 != 0) return false;
  if (N == 4096) return true;       // 4 sub-rows of 1024
  if (wave_rt_split(N)) return true;   // 64 P S with a run-time sub-row count (k_rows_mr<..., LN = 64>)
  switch (N / 64) {
    case 2: case 3: case 4: case 5: case 6: case 7: case 8: case 9: case 10: case 12: case 14: case 16: case 18: case 20:
    case 24: case 28: case 32: return true;
    default: return false;
  }
}

// Default kernel family: wave where N = 64 P, else chirp-z where it applies and the grid is big enough to pay for two
// transforms per row (below ~96 points the direct kernels win), else direct.
static bool wave_supported(int N);
static int default_path(int N, int blu_P, int mr_P) { return wave_supported(N) ? 1 : (mr_P ? 3 : ((blu_P && N >= 96) ? 2 : 0)); }

// Chirp-z family: smallest M = 64 P (P = 4, 8, 16, 24, 32) with M >= N + Np - 1 and a window instantiation
// (NS = 2: Np <= 128; NS = 4: Np <= 256, P = 8, 16, 24); 0 when there is none.
static int blu_pick_P(int N, int Np, int* SB = nullptr, int* B = nullptr) {
  if (SB) *SB = 1;
  if (B) *B = N;
  // (Every N: the grids of other generator layouts -- 50 P S, the multiples of 64 -- take these rows for host / staged coefficients
  // only: family_streams_ok.  Rounds 2-5 excluded them here.)
  if (N < 2) return 0;
  const int ns = (Np + 63) / 64;
  if (ns > 4) return 0;
  // (12 and 28 -- M = 768, 1792 -- since round 6, windows of up to 128 pixels: +10-13 % over the next of 1024 / 2048; M = 1280 (P = 20)
  // spills 356 B at eight waves and LOSES 40 % to M = 1536: not built -- profiles/r06_ab_chirpz_sizes.txt; FASTMC_BLU_P=0 keeps the
  // five sizes of rounds 1-5 for an A/B)
  static const bool more = !(getenv("FASTMC_BLU_P") && atoi(getenv("FASTMC_BLU_P")) == 0);
  for (int P : {4, 8, 12, 16, 24, 28, 32}) {
    if (64 * P < N + Np - 1) continue;
    if ((P == 12 || P == 28) && (!more || ns > 2)) continue;
    if (ns > 2 && !(P == 8 || P == 16 || P == 24)) continue;
    return P;
  }
  // rows beyond the largest M: input blocks on the M = 1024 pipeline (2200, 2816, ... 4095: up to six blocks; beyond 4096 up to
  // eleven: 8192 / 768 at Np = 256)
  const int nblk = blu_blocks(N, Np);
  if (nblk >= 2 && nblk <= 12) {
    if (SB) *SB = nblk;
    if (B) *B = blu_block_len(Np);
    return 16;
  }
  return 0;
}

// 50-lane family: N = 50 P with a window instantiation (NS = 2: Np <= 128; NS = 4: Np <= 256 for P = 8, 10, 12, 16, 20, 24)
constexpr bool mr_has_ns4(int P) { return P == 8 || P == 10 || P == 12 || P == 16 || P == 20 || P == 24; }
static int mr_pick_P(int N, int Np) {
  if (!mr_supported(N)) return 0;
  const int P = N / MR_LN / mr_split(N), ns = (Np + 63) / 64;
  if (ns <= 2) return P;
  return (ns <= 4 && mr_has_ns4(P)) ? P : 0;
}

// Which window instantiation of the wave family serves this handle: NS = 2 (Np <= 128), NS = 4
// (Np <= 256; P = 8, 16, 32), NS = P (any window, powers of two), or 0 when none fits the LDS.
constexpr size_t LDS_MAX = 160 * 1024;
// P with an NS = 4 instantiation (windows of 129-256 pixels): 512, 576, 640, 768, 1024 (and 2048 / 4096 as sub-rows), 1280, 1536
constexpr bool has_ns4(int P) { return P == 8 || P == 9 || P == 10 || P == 12 || P == 16 || P == 20 || P == 24; }
template <class R, int PP>
static int pick_ns(const fastmc_ctx* h) {
  if (h->NS <= 2) return wave_lds_bytes<R, PP, 2>(h->omS) <= LDS_MAX ? 2 : 0;
  if constexpr (has_ns4(PP)) {
    if (h->NS <= 4 && wave_lds_bytes<R, PP, 4>(h->omS) <= LDS_MAX) return 4;
  }
  if constexpr (PP == 16) {  // windows of 257-512 pixels (also on the split grids 2048 / 4096): eight output slots per lane
    if (h->NS <= 8 && wave_lds_bytes<R, PP, 8>(h->omS) <= LDS_MAX) return 8;
  }
  if (h->S > 1) return 0;    // split rows: windows up to 512 pixels only
  if constexpr (is_pow2(PP) && PP >= 4) {
    if (h->NS <= PP && wave_lds_bytes<R, PP, PP>(h->omS) <= LDS_MAX) return PP;
  }
  return 0;
}

// Launch configuration of the wave family for this handle: instantiation, waves per workgroup.
template <class R>
static void wave_config(const fastmc_ctx* h, int* ns, int* wpb) {
#define FMC_CASE(PP)                                                                                         \
  case PP:                                                                                                   \
    *ns = pick_ns<R, PP>(h);                                                                                 \
    *wpb = WaveCfg<R, PP, 2>::WPB;                                                                           \
    if constexpr (has_ns4(PP)) { if (*ns == 4) *wpb = WaveCfg<R, PP, 4>::WPB; }                              \
    if constexpr (PP == 16) { if (*ns == 8) *wpb = WaveCfg<R, PP, 8>::WPB; }                                   \
    if constexpr (is_pow2(PP) && PP >= 4) { if (*ns == PP && PP > 2) *wpb = WaveCfg<R, PP, PP>::WPB; }       \
    break;
  switch (h->P) {
    FMC_CASE(2) FMC_CASE(3) FMC_CASE(4) FMC_CASE(5) FMC_CASE(6) FMC_CASE(7) FMC_CASE(8) FMC_CASE(9) FMC_CASE(10)
    FMC_CASE(12) FMC_CASE(14) FMC_CASE(16) FMC_CASE(18) FMC_CASE(20) FMC_CASE(24) FMC_CASE(28) FMC_CASE(32)
    default: *ns = 0; *wpb = 1; break;
  }
#undef FMC_CASE
}

// One retired handle per device is kept whole (stream, events, every device buffer) and handed to the next
// fastmc_create of the same (N, Np, precision): a sweep builds one short-lived handle per geometry sample
// (fast/complete_orbit_simulation.py:217-228) and ~15 hipMalloc / hipFree pairs per object cost more than its spectrum.
#if FMC_TU == 0
struct HandleCache {
  std::mutex mu;
  fastmc_ctx* dev[64] = {};
  fastmc_ctx* take(int device, int N, int Np, int precision);
  fastmc_ctx* swap_in(fastmc_ctx* h);     // returns the handle to free (the previous occupant), or nullptr
};
static HandleCache g_handles;

extern "C" int fastmc_create(fastmc_t** out, int device_id, int N, int Np, int precision) {
  if (!out) return fail(FASTMC_EINVAL, "handle pointer is NULL");
  *out = nullptr;
  // Every N <= 4096 has a kernel family.  Beyond it the grids with a run-time sub-row count go on: N = 64 P S (wave_rt_split:
  // 4608, 5120, 6144, 6400, 7168, 7680, 8192 ...) and N = 50 P S (mr_split: 4200, 4500, 4800, 5000, 6000, 7000, 8000 ...), S <= 8
  // sub-rows of P <= 24 values per lane -- nothing in their kernels depends on the size.  The cap of 8192 is memory and test
  // coverage, not the kernels (the spectrum tables alone are 12 B per pixel: 0.8 GB at 8192^2).
  // ... and any other N <= 8192 through the chirp-z kernels with its rows in input blocks (blu_pick_P), for windows of up to 256 pixels.
  if (N < 4 || (N > 4096 && !(N <= 8192 && (wave_rt_split(N) || mr_supported(N) || (Np >= 1 && Np <= N && blu_pick_P(N, Np))))))
    return fail(FASTMC_EINVAL, "N must be in [4, 8192]; beyond 4096 the window must be at most 256 pixels unless N is a grid of S <= 8 sub-rows, "
                               "N = 64 P S / 50 P S with 7 <= P <= 24");
  if (Np < 1 || Np > N) return fail(FASTMC_EINVAL, "Np must be in [1, N]");
  if (precision != FASTMC_F64 && precision != FASTMC_F32) return fail(FASTMC_EINVAL, "precision must be FASTMC_F64 or FASTMC_F32");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(FASTMC_ENODEV, "no HIP device visible: libfastmc has no CPU fallback");
  if (device_id < 0 || device_id >= count) return fail(FASTMC_EINVAL, "device_id out of range");
  HIPCHK(hipSetDevice(device_id));
  if (fastmc_ctx* r = g_handles.take(device_id, N, Np, precision)) {
    *out = r;
    return 0;
  }
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device_id));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(FASTMC_ENODEV, std::string("device is ") + prop.gcnArchName + ", libfastmc is built for gfx950 only");
  fastmc_ctx* h = new fastmc_ctx();
  h->device = device_id;
  h->N = N;
  h->Np = Np;
  h->precision_req = precision;
  // the float32 pipeline exists for the wave family (N = 64 P, 2048, 4096) and its direct cross-check; every other grid
  // (chirp-z, 50-lane, run-time sub-rows, tiny) computes in float64 whatever was asked for
  if (precision == FASTMC_F32 && (!wave_supported(N) || wave_rt_split(N))) precision = FASTMC_F64;
  h->precision = precision;
  // the device generator starts at the precision the handle computes in: a float64 handle draws the reference's 53-bit normals
  // and colours in float64 (fast/funcs.py:352-356, fast/fast.py:594); fastmc_set_rng_precision(FASTMC_F32) opts into the float32 draw
  h->rng_f64 = precision == FASTMC_F64;
  h->rsz = precision == FASTMC_F64 ? 8 : 4;
  h->blu_P = blu_pick_P(N, Np, &h->blu_SB, &h->blu_B);
  if (const char* e = getenv("FASTMC_NO_DENSE16")) h->no_dense = e[0] && e[0] != '0';
  h->mr_P = mr_pick_P(N, Np);
  h->mr_S = h->mr_P ? mr_split(N) : 1;
  h->path = default_path(N, h->blu_P, h->mr_P);
  h->S = h->path == 1 ? spec_split(N) : 1;
  h->P = N / 64 / h->S;
  h->NS = (Np + 63) / 64;
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete h; return fail(FASTMC_EHIP, hipGetErrorString(e)); }
  *out = h;
  return 0;
}
#endif

// One retired V slab per device stays allocated for the next handle: a sweep builds hundreds of
// short-lived handles (fast/complete_orbit_simulation.py:227-236 builds one Fast per pass) and a
// 1.5 GiB hipMalloc / hipFree pair costs 60-150 ms every few objects (tools/alloc_probe.py).
struct SlabCache {
  std::mutex mu;
  struct Entry { void* p = nullptr; size_t bytes = 0; } dev[64];
  void* take(int device, size_t need, size_t* bytes) {
    std::lock_guard<std::mutex> g(mu);
    Entry& e = dev[device & 63];
    if (!e.p || e.bytes < need || e.bytes > 4 * need + ((size_t)64 << 20)) return nullptr;
    void* p = e.p;
    *bytes = e.bytes;
    e = Entry();
    return p;
  }
  void give(int device, void* p, size_t bytes) {     // keeps the larger slab, frees the other
    void* drop = p;
    {
      std::lock_guard<std::mutex> g(mu);
      Entry& e = dev[device & 63];
      if (!e.p || e.bytes < bytes) { drop = e.p; e.p = p; e.bytes = bytes; }
    }
    if (drop) hipFree(drop);
  }
};
static SlabCache g_slabs;

#if FMC_TU == 0
fastmc_ctx* HandleCache::take(int device, int N, int Np, int precision) {
  std::lock_guard<std::mutex> g(mu);
  fastmc_ctx* h = dev[device & 63];
  if (!h || h->device != device || h->N != N || h->Np != Np || h->precision_req != precision) return nullptr;
  dev[device & 63] = nullptr;
  return h;
}
fastmc_ctx* HandleCache::swap_in(fastmc_ctx* h) {
  std::lock_guard<std::mutex> g(mu);
  fastmc_ctx* old = dev[h->device & 63];
  dev[h->device & 63] = h;
  return old;
}
#endif

static void destroy_now(fastmc_ctx* h);

#if FMC_TU == 0
extern "C" void fastmc_destroy(fastmc_t* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  finish_pending(h);
  if (h->cstream) hipStreamSynchronize(h->cstream);
  h->copy_guard = nullptr;
  for (QueueSlot& q : h->q) { q.busy = q.pending = q.ex_recorded = q.stalled = false; q.landed = q.hist_landed = 0; }
  // back to the state fastmc_create leaves: problem unset, results forgotten, options at their defaults; buffers kept
  h->have_spec = h->have_pupil = h->have_sh = h->have_ps = false;
  h->last_n_iter = 0;
  h->last_out_doubles = 0;
  h->last_coherent = 0;
  h->last_rows[0] = h->last_cols[0] = 0;
  h->batch = 0;
  h->rng_f64 = h->precision == FASTMC_F64;      // the state fastmc_create leaves
  if (h->comm_slot >= 0) { fastmc_comm_abort(h); h->comm_slot = -1; }     // (fake-RCCL tests: the slot belonged to this handle, not to a device)
  h->path = default_path(h->N, h->blu_P, h->mr_P);
  h->lo = 0;
  h->df = h->dx = h->wsum = 0;
  if (h->layers) { hipFree(h->layers); h->layers = nullptr; h->n_layers = 0; }
  if (h->cre) { hipFree(h->cre); h->cre = nullptr; }
  if (h->cim) { hipFree(h->cim); h->cim = nullptr; }
  h->coef_cap = 0;
  if (h->phs) { hipFree(h->phs); h->phs = nullptr; h->phs_cap = 0; }
  if (fastmc_ctx* old = g_handles.swap_in(h)) destroy_now(old);
}
#endif

#if FMC_TU == 0
static void nps_free(NpsWork* w);
#endif
static void destroy_now(fastmc_ctx* h) {
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
#if FMC_TU == 0
  if (h->nps) { nps_free(h->nps); h->nps = nullptr; }
#endif
  if (h->V) { g_slabs.give(h->device, h->V, h->V_bytes); h->V = nullptr; }
  void* ptrs[] = {h->mr_tw1, h->mr_om, h->mr_cw, h->blu_tw1, h->blu_om, h->blu_twf, h->blu_pre, h->blu_vhat, h->blu_post, h->pbz_tw, h->pbz_pre, h->pbz_vhat, h->pbz_post, h->bad, h->ampf, h->ampf_s, h->amp, h->amp_s, h->amp_p, h->ampf_p, h->tw, h->tw1, h->om, h->cw, h->tw1g, h->omg, h->pk_tw1, h->pk_om, h->pks_tw1, h->pks_cw, h->pks_cw8, h->pks_cw16, h->W, h->V, h->partial, h->out, h->logamp, h->cre,
                  h->cim, h->phs, h->sh_scale, h->sh_mu, h->sh_ex, h->sh_ey, h->sh_coef, h->sh_mean, h->sh_dcol, h->sh_in_re,
                  h->sh_in_im, h->hist, h->gather_buf, h->layers, h->ps_dev, (void*)h->clk};
  for (void* p : ptrs)
    if (p) hipFree(p);
  for (auto e : h->pool) hipEventDestroy(e);
  if (h->cstream) { hipStreamSynchronize(h->cstream); hipStreamDestroy(h->cstream); }
  if (h->copy_fence) hipEventDestroy(h->copy_fence);
  for (QueueSlot& q : h->q) {
    for (auto e : q.pool) hipEventDestroy(e);
    for (hipEvent_t e : {q.ex_a, q.ex_b, q.done}) if (e) hipEventDestroy(e);
    if (q.pinned) hipHostFree(q.pinned);
    if (q.pinned_hist) hipHostFree(q.pinned_hist);
  }
  if (h->ex_a) hipEventDestroy(h->ex_a);
  if (h->ex_b) hipEventDestroy(h->ex_b);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

// Exclusive use of a handle for the length of an entry point.  A handle that stays busy for FASTMC_HANDLE_BUSY_TIMEOUT
// seconds (default 30) is an error, not a hang: it means another thread -- typically an exchange that missed its deadline --
// is still inside the library on this handle.
static double handle_busy_timeout() {
  const char* e = getenv("FASTMC_HANDLE_BUSY_TIMEOUT");
  const double v = e ? atof(e) : 30.0;
  return v > 0 ? v : 30.0;
}
struct HandleLock {
  std::unique_lock<std::timed_mutex> lk;
  bool ok;
  explicit HandleLock(fastmc_ctx* h) : lk(h->use_mu, std::defer_lock), ok(false) {
    ok = lk.try_lock_for(std::chrono::duration<double>(handle_busy_timeout()));
  }
};
#define FMC_LOCK(h)                                                                                                     \
  HandleLock lock__(h);                                                                                                 \
  if (!lock__.ok) return fail(FASTMC_ESTATE, "handle is in use by another thread (an exchange that missed its deadline has not returned)")

#if FMC_TU == 0
extern "C" int fastmc_kernel_path(fastmc_t* h, int force) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  if (force == 1 && !wave_supported(h->N)) return fail(FASTMC_EINVAL, "wave kernels need N = 64 P with P = 2^k times 1, 3, 5, 7 or 9, 2 <= P <= 32");
  if (force == 2 && !h->blu_P) return fail(FASTMC_EINVAL, "chirp-z kernels need Np <= 256 and a grid that is neither 64 P S nor 50 P S");
  if (force == 3 && !h->mr_P) return fail(FASTMC_EINVAL, "50-lane kernels need N = 50 P S (not 64 P') with P = 2^k times 1, 3, 5, 7 or 9, P <= 24, S <= 5, and Np <= 128 (256 for P = 8, 10, 12, 16, 20, 24)");
  if (force >= 0 && force <= 3) h->path = force;
  return h->path;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_last_kernels(fastmc_t* h, char* rows, char* cols, int cap) {
  if (!h || !rows || !cols || cap < 1) return fail(FASTMC_EINVAL, "null handle / buffers");
  snprintf(rows, (size_t)cap, "%s", h->last_rows);
  snprintf(cols, (size_t)cap, "%s", h->last_cols);
  return 0;
}

// Effective shader clock of the last row-kernel launch that stamped it (k_rows_wave: the wave family): shader-clock ticks per
// constant-rate tick inside one workgroup in the middle of the launch.  *ghz: the clock in GHz; *span_us: how long the stamping
// workgroup ran.  Blocks until the handle's stream is idle.  FASTMC_ESTATE when no launch has stamped yet.
extern "C" int fastmc_last_clock(fastmc_t* h, double* ghz, double* span_us) {
  if (!h || !ghz || !span_us) return fail(FASTMC_EINVAL, "null handle / outputs");
  FMC_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  if (!h->clk || !h->clk_fresh) return fail(FASTMC_ESTATE, "the last row kernel of this handle did not stamp the clock (only k_rows_wave does)");
  unsigned long long c[4] = {0, 0, 0, 0};
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(c, h->clk, sizeof(c), hipMemcpyDeviceToHost));
  if (c[3] <= c[1] || c[2] <= c[0]) return fail(FASTMC_ESTATE, "no row kernel has stamped the clock on this handle yet");
  int wall_khz = 0;
  HIPCHK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, h->device));
  if (wall_khz <= 0) wall_khz = 100000;                       // gfx9: s_memrealtime counts at 100 MHz
  const double wall_s = (double)(c[3] - c[1]) / ((double)wall_khz * 1e3);
  *ghz = (double)(c[2] - c[0]) / wall_s * 1e-9;
  *span_us = wall_s * 1e6;
  return 0;
}

// shape of the result vector resident on the device (what fastmc_wait copies): iterations, coherent flag; 0 iterations = none
extern "C" int fastmc_last_result_shape(fastmc_t* h, int64_t* n_iter, int* coherent) {
  if (!h || !n_iter || !coherent) return fail(FASTMC_EINVAL, "null handle / outputs");
  *n_iter = h->last_out_doubles ? h->last_n_iter : 0;
  *coherent = h->last_coherent;
  return 0;
}

// the precision the handle computes in (fastmc_create promotes FASTMC_F32 to FASTMC_F64 on grids without float32 kernels)
extern "C" int fastmc_precision(fastmc_t* h) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  return h->precision;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_set_batch(fastmc_t* h, int batch) {
  if (!h || batch < 0) return fail(FASTMC_EINVAL, "bad batch");
  h->batch = batch;
  return 0;
}
#endif

static int default_batch(const fastmc_ctx* h);
static int pks_p16_from();
static bool pbz_ok(const fastmc_ctx* h);
#if FMC_TU == 0
extern "C" int fastmc_get_batch(fastmc_t* h, int* batch) {
  if (!h || !batch) return fail(FASTMC_EINVAL, "null argument");
  *batch = default_batch(h);
  return 0;
}
#endif

// The tables of the float64 generator (fmc_gen64.h: 256 (cos, sin) entries, 128 log entries), one copy per device, uploaded on first use and kept.
// (Host helpers of every translation unit -- run_locked below refers to them -- but only unit 0's entry points ever call them.)
static const Gen64Entry* gen64_table(int device) {
  static std::mutex mu;
  static std::map<int, Gen64Entry*> tabs;
  std::lock_guard<std::mutex> lk(mu);
  auto it = tabs.find(device);
  if (it != tabs.end()) return it->second;
  Gen64Entry host[GEN64_LOG_ENTRIES + GEN64_TRIG_ENTRIES];
  gen64_build_table(host);
  Gen64Entry* d = nullptr;
  if (hipMalloc((void**)&d, GEN64_TABLE_BYTES) != hipSuccess) return nullptr;
  if (hipMemcpy(d, host, GEN64_TABLE_BYTES, hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return nullptr; }
  tabs[device] = d;
  return d;
}
static int ensure_gen64_table(fastmc_ctx* h) {       // the float64 generator's tables on this handle's device (uploaded once per device)
  if (h->rng_f64 && !h->g64) {
    HIPCHK(hipSetDevice(h->device));
    h->g64 = gen64_table(h->device);
    if (!h->g64) return fail(FASTMC_EHIP, "could not upload the float64 generator's table");
  }
  return 0;
}
#if FMC_TU == 0
extern "C" int fastmc_set_rng_precision(fastmc_t* h, int precision) {
  if (!h || (precision != FASTMC_F64 && precision != FASTMC_F32)) return fail(FASTMC_EINVAL, "precision must be FASTMC_F64 or FASTMC_F32");
  h->rng_f64 = precision == FASTMC_F64;
  return ensure_gen64_table(h);
}
#endif

// the 16 x 4 dense kernels stage four table rows: windows of up to 128 pixels fit
template <class R>
static bool dense16r_fits(const fastmc_ctx* h) {
  return wave_lds_bytes_d<R, 16, 2, 4>(h->omS) <= 160 * 1024;
}

static int default_batch(const fastmc_ctx* h) {
  if (h->batch > 0) return h->batch;
  // V slab (batch * N * Np complex) of up to 2 GiB: fewer, larger launches (measured at 1024^2 f64:
  // 96 / 216 / 1008 realisations per launch -> 0.97 / 1.00 / 1.05 of the throughput; Infinity-Cache
  // residency of the slab does not matter, the pipeline is VALU-bound, and HBM is 288 GB)
  const double per = (double)h->N * h->Np * 2 * h->rsz;
  int b = (int)(2048.0 * 1024 * 1024 / per);       // round 2: 1568 realisations per launch +0.9 % over 1176 at 1024^2
  b = std::max(1, std::min(b, 4096));
  if (h->path != 0 && (pks_grid(h->N) || (h->path == 1 && h->rsz == 8 && pks_p16(h->N) && h->N >= pks_p16_from())) &&
      ((h->lo >= h->N / 2 - 64 && h->lo + h->Np <= h->N / 2 + 64) || (h->rsz == 8 && pks_L0(h->N) == 1 && h->lo >= h->N / 2 - 128 && h->lo + h->Np <= h->N / 2 + 128)) &&
      !(h->rsz == 4 && (pks_rt(h->N) || h->lo < h->N / 2 - 48 || h->lo + h->Np > h->N / 2 + 48))) {
    // packed sub-rows (pks_variant; fmc_kernels.h: k_rows_pks): a tile is one 128-byte line of V positions x BPG realisations, and the
    // launch's workgroups (one per CU) walk the tiles -- whole groups of BPG realisations, and of the group counts within a quarter of
    // the slab limit the one whose tiles fill the last round over the 256 CUs best (1856: 10 % of the launch was an almost empty round)
    const bool planes8 = h->lo < h->N / 2 - 48 || h->lo + h->Np > h->N / 2 + 48, planes16 = h->lo < h->N / 2 - 64 || h->lo + h->Np > h->N / 2 + 64;
    const int L0 = pks_L0(h->N), WPB = planes16 ? 8 : planes8 ? (L0 == 1 ? 12 : 8) : (L0 == 0 ? (h->rsz == 8 ? FMC_PKS_WPB0 : 12) : FMC_PKS_WPB), G = L0 == 1 ? 4 : 8;
    const int LR = 128 / (2 * h->rsz), BPG = ROWS_PER_WAVE * WPB / (LR / G), nmax = b / BPG;
    if (nmax >= 1) {
      int best = nmax;
      double best_eff = 0.0;
      for (int nbb = nmax; nbb >= std::max(1, nmax - nmax / 4); --nbb) {
        const int64_t t = (int64_t)(h->N / LR) * nbb;
        const double eff = t >= 8 * 256 ? (double)t / (double)(((t + 255) / 256) * 256) : 1.0;      // (fewer tiles: one workgroup each, no walk)
        if (eff > best_eff + 1e-9) { best_eff = eff; best = nbb; }
      }
      return best * BPG;
    }
    return b;
  }
  if (h->path == 2 && pbz_ok(h)) {
    // chirp-z rows on the packed pipeline (k_rows_pbz): tiles of one 128-byte line of V (eight rows) x 32 realisations, walked by one
    // workgroup per CU -- the same rule as for the packed sub-rows
    const int BPG = ROWS_PER_WAVE * PBZ_WPB / 2, nmax = b / BPG;
    if (nmax >= 1) {
      int best = nmax;
      double best_eff = 0.0;
      for (int nbb = nmax; nbb >= std::max(1, nmax - nmax / 4); --nbb) {
        const int64_t t = (int64_t)((h->N + 7) / 8) * nbb;
        const double eff = t >= 8 * 256 ? (double)t / (double)(((t + 255) / 256) * 256) : 1.0;
        if (eff > best_eff + 1e-9) { best_eff = eff; best = nbb; }
      }
      return best * BPG;
    }
    return b;
  }
  if (h->path == 1) {
    // whole number of workgroup rounds over the 256 CUs: the row kernel runs one 12-wave (P=32: 4/6)
    // workgroup per CU and has batch * N/8 wave-items
    int ns = 0, wpb = 1;
    if (h->rsz == 8) wave_config<double>(h, &ns, &wpb); else wave_config<float>(h, &ns, &wpb);
    if (h->P == 16 && ns == 2 && h->S == 1 && !h->no_dense && (h->rsz == 8 ? dense16r_fits<double>(h) : dense16r_fits<float>(h))) wpb = 16;
    int quantum = std::max(1, 256 * wpb * ROWS_PER_WAVE / h->N);
    if (pk_grid(h->N)) quantum = 256 * (h->N == 512 ? PkCfg<double, 2, 0>::WPB : PkCfg<double, 1, 0>::WPB) * ROWS_PER_WAVE / 16;   // (64 * 16 / N rows per unit) / N
    if (b >= quantum) b -= b % quantum;
  } else if (b >= 8) {
    b &= ~7;
  }
  return b;
}

// ------------------------------------------------------------------ set_spectrum / set_pupil
// Colouring tables from a spectrum already on the device (`d_ps`), on `stream`; synchronises it.
template <class R>
static int make_amp(fastmc_ctx* h, const double* d_ps, double df, hipStream_t stream) {
  const int N = h->N;
  const size_t n = (size_t)N * N;
  if (!h->amp) HIPCHK(hipMalloc(&h->amp, sizeof(R) * n));
  if (!h->amp_s) HIPCHK(hipMalloc(&h->amp_s, sizeof(R) * n));
  if (!h->ampf) HIPCHK(hipMalloc((void**)&h->ampf, sizeof(float) * n));
  if (!h->ampf_s) HIPCHK(hipMalloc((void**)&h->ampf_s, sizeof(float) * n));
  if (!h->bad) HIPCHK(hipMalloc((void**)&h->bad, 8));
  if (pks_count(N) && !h->amp_p) { HIPCHK(hipMalloc(&h->amp_p, sizeof(R) * n)); HIPCHK(hipMalloc((void**)&h->ampf_p, sizeof(float) * n)); }
  HIPCHK(hipMemsetAsync(h->bad, 0, 8, stream));
  hipLaunchKernelGGL((k_make_amp<R>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_ps, df, N, (R*)h->amp,
                     (R*)h->amp_s, h->ampf, h->ampf_s, h->bad, (R*)h->amp_p, h->ampf_p, pks_count(N) ? pks_count(N) : 1);
  HIPCHK(hipGetLastError());
  unsigned int bad = 0;
  HIPCHK(hipMemcpyAsync(&bad, h->bad, 4, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  if (bad) return fail(FASTMC_EINVAL, "powerspec must be finite and non-negative");
  // direct-family twiddles w_N^e
  if (!h->tw) {
    std::vector<cpx<R>> tw(N);
    for (int e = 0; e < N; ++e) {
      double cc, ss;
      cs_turns((double)e / N, &cc, &ss);
      tw[e] = mk<R>((R)cc, (R)(-ss));
    }
    HIPCHK(hipMalloc(&h->tw, sizeof(cpx<R>) * N));
    HIPCHK(hipMemcpy(h->tw, tw.data(), sizeof(cpx<R>) * N, hipMemcpyHostToDevice));
  }
  return 0;
}

template <class R>
static int upload_spectrum(fastmc_ctx* h, const double* ps, double df) {
  const size_t n = (size_t)h->N * h->N;
  ScratchBuf d_ps;
  HIPCHK(hipMalloc((void**)&d_ps.p, n * 8));
  HIPCHK(hipMemcpyAsync(d_ps.p, ps, n * 8, hipMemcpyHostToDevice, h->stream));
  return make_amp<R>(h, d_ps.p, df, h->stream);
}

#if FMC_TU == 0
static int make_amp_from_device(fastmc_ctx* h, const double* d_ps, double df, hipStream_t stream) {
  h->df = df;
  h->have_spec = false;
  TRY(h->precision == FASTMC_F64 ? make_amp<double>(h, d_ps, df, stream) : make_amp<float>(h, d_ps, df, stream));
  h->have_spec = true;
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_set_spectrum(fastmc_t* h, const double* powerspec, double df) {
  if (!h || !powerspec) return fail(FASTMC_EINVAL, "null argument");
  HIPCHK(hipSetDevice(h->device));
  h->df = df;
  h->have_spec = false;
  TRY(h->precision == FASTMC_F64 ? upload_spectrum<double>(h, powerspec, df) : upload_spectrum<float>(h, powerspec, df));
  h->have_spec = true;
  return 0;
}
#endif

template <class R>
static int upload_table(void** dst, const std::vector<cpx<R>>& v) {
  if (*dst) HIPCHK(hipFree(*dst));
  *dst = nullptr;
  HIPCHK(hipMalloc(dst, sizeof(cpx<R>) * v.size()));
  HIPCHK(hipMemcpy(*dst, v.data(), sizeof(cpx<R>) * v.size(), hipMemcpyHostToDevice));
  return 0;
}

template <class R>
static int upload_wave_tables(fastmc_ctx* h) {
  const int P = h->P, S = h->S;
  h->omS = (h->Np + 7) & ~7;
  std::vector<cpx<R>> tw1((size_t)P * 64), om((size_t)8 * h->omS);
  build_tw1<R>(tw1.data(), P, cs_turns);
  build_om<R>(om.data(), h->omS, P, h->lo, h->Np, cs_turns);
  TRY(upload_table<R>(&h->tw1, tw1));
  TRY(upload_table<R>(&h->om, om));
  if (S > 1) {
    // X[x] = sum_s w_N^{s x} Y_s[x mod N/S]:  cw[s][oi] = w_N^{s (lo + oi)}
    std::vector<cpx<R>> cw((size_t)S * h->omS);
    for (int sp = 0; sp < S; ++sp)
      for (int oi = 0; oi < h->omS; ++oi) {
        double c, sn;
        cs_turns((double)(((long long)sp * (h->lo + oi)) % h->N) / h->N, &c, &sn);
        cw[(size_t)sp * h->omS + oi] = mk<R>((R)c, (R)(-sn));
      }
    TRY(upload_table<R>(&h->cw, cw));
  }
  if (h->N == 2048) {
    std::vector<cpx<R>> tw1g((size_t)32 * 64), omg((size_t)8 * h->omS);
    build_tw1<R>(tw1g.data(), 32, cs_turns);
    build_om<R>(omg.data(), h->omS, 32, h->lo, h->Np, cs_turns);
    TRY(upload_table<R>(&h->tw1g, tw1g));
    TRY(upload_table<R>(&h->omg, omg));
  }
  if (pk_grid(h->N)) {
    const int L = h->N / 16;
    std::vector<cpx<R>> tw((size_t)16 * L), omp((size_t)2 * h->omS);
    build_tw1_pk<R>(tw.data(), L, cs_turns);
    build_om_pk<R>(omp.data(), h->omS, L, h->lo, h->Np, cs_turns);
    TRY(upload_table<R>(&h->pk_tw1, tw));
    TRY(upload_table<R>(&h->pk_om, omp));
  }
  return 0;
}
// Tables of the packed sub-rows (fmc_core.h: pks_split; they depend on N only).  The grids are those of the wave family and, with a
// run-time sub-row count, of the chirp-z and 50-lane families too (fmc_core.h: pks_rt): fastmc_set_pupil calls this for every family.
template <class R>
static int upload_pks_tables(fastmc_ctx* h) {
  if (!pks_count(h->N) || h->pks_tw1) return 0;
  const int Sp = pks_count(h->N), L = pks_L0(h->N) < 0 ? 4 : pk_lanes(pks_L0(h->N));     // (64-point sub-rows: 8 x 8 = 16 x 4 entries)
  std::vector<cpx<R>> tw((size_t)16 * L), pcw((size_t)Sp * PKS_SPAN);
  if (pks_L0(h->N) < 0) build_tw64<R>(tw.data(), cs_turns);
  else build_tw1_pk<R>(tw.data(), L, cs_turns);
  build_pcw<R>(pcw.data(), h->N, Sp, cs_turns);
  TRY(upload_table<R>(&h->pks_tw1, tw));
  TRY(upload_table<R>(&h->pks_cw, pcw));
  if constexpr (sizeof(R) == 8) {
    std::vector<cpx<R>> pcw8((size_t)Sp * pks_span(8));
    build_pcw<R>(pcw8.data(), h->N, Sp, cs_turns, 8);
    TRY(upload_table<R>(&h->pks_cw8, pcw8));
    if (pks_L0(h->N) == 1) {
      std::vector<cpx<R>> pcw16((size_t)Sp * pks_span(16));
      build_pcw<R>(pcw16.data(), h->N, Sp, cs_turns, 16);
      TRY(upload_table<R>(&h->pks_cw16, pcw16));
    }
  }
  return 0;
}

// Which packed-row variant serves this handle's window (fmc_kernels.h: PkCfg): 0 six centred planes, 1 all planes, -1 none
// (a window of more than 256 pixels at N = 512: host coefficients go to the one-row-per-wave kernels, device draws to the
// direct family -- the P = 4 / 8 rows of fmc_wavefft.h do not know the 16 / 32-stream generator layout of these grids).
template <class R>
static int pk_variant(const fastmc_ctx* h) {
  if (!pk_grid(h->N) || h->path != 1) return -1;
  const int centre = h->N == 128 ? pk_centre_mask<0>() : (h->N == 256 ? pk_centre_mask<1>() : pk_centre_mask<2>());
  if (h->Np <= 96 && (window_planes(h->lo, h->Np, 16, h->N == 128 ? 8 : 16) & ~centre) == 0) return 0;
  if (h->Np <= 256) return 1;     // tables + sixteen exchange buffers fit the LDS for every such window
  return -1;
}

// Do the packed sub-rows serve this handle's window (fmc_kernels.h: k_rows_pks)?  0: yes -- a window of up to 96 pixels inside the
// six centred planes of the sub-transforms; -1: no (host coefficients and the column pass always go to the one-row-per-wave
// kernels; device draws with any other window are staged (float64 generator) or go to the direct family (float32 draw): the
// P = 12 / 20 / 28 rows of fmc_wavefft.h do not know the N / 16-stream generator layout of these grids).
// The smallest grid of the P = 16 rows (1024, 2048, 4096) that the packed sub-rows serve instead (fmc_core.h: pks_p16).
static int pks_p16_from() {
  static const int from = getenv("FASTMC_PKS_P16") ? atoi(getenv("FASTMC_PKS_P16")) : FMC_PKS_P16_FROM;
  return from;
}
template <class R>
static int pks_variant(const fastmc_ctx* h) {
  if (!pks_count(h->N) || h->path == 0) return -1;         // (path 0: the direct family was asked for -- fastmc_set_kernel_path)
  if (pks_p16(h->N) && (h->N < pks_p16_from() || sizeof(R) != 8 || h->path != 1)) return -1;
  if (sizeof(R) != 8 && pks_rt(h->N)) return -1;           // (run-time counts: float64 pipeline only; fastmc_create never makes such a handle)
  static const bool off = getenv("FASTMC_PKS") && atoi(getenv("FASTMC_PKS")) == 0;     // A/B: staged / direct instead
  if (off) return -1;
  // the six planes of a sub-transform hold x = N / 2 - 48 ... N / 2 + 47 (fmc_wavefft.h: pks_accumulate) ...
  if (h->lo >= h->N / 2 - 48 && h->lo + h->Np <= h->N / 2 + 48) return 0;
  // ... eight planes N / 2 - 64 ... N / 2 + 63: 1 (float64 pipeline, run-time sub-row counts for every grid; FASTMC_PKS8=0: off)
  static const bool off8 = getenv("FASTMC_PKS8") && atoi(getenv("FASTMC_PKS8")) == 0;
  if (sizeof(R) == 8 && !off8 && h->lo >= h->N / 2 - 64 && h->lo + h->Np <= h->N / 2 + 64) return 1;
  // ... all sixteen planes of a 256-point sub-transform, N / 2 - 128 ... N / 2 + 127: 2 (grids N = S x 256 only; FASTMC_PKS16=0: off)
  static const bool off16 = getenv("FASTMC_PKS16") && atoi(getenv("FASTMC_PKS16")) == 0;
  if (sizeof(R) == 8 && !off8 && !off16 && pks_L0(h->N) == 1 && h->lo >= h->N / 2 - 128 && h->lo + h->Np <= h->N / 2 + 128) return 2;
  return -1;
}

// The chirp-z rows draw 64 generator streams per row and the 50-lane rows 50 S: a grid whose layout is another one (fmc_core.h:
// stream_lanes -- the packed grids, 2048, 4096 and every grid of the packed sub-rows draw N / 16 or N / 8) reaches those families only
// when a caller forces them (fastmc_kernel_path) or as the host-coefficient rows of a packed sub-row grid; its device draws must not
// go through their rows (the float64 generator is staged, the float32 draw goes to the direct family: run_impl, fused_gen64).
static bool family_streams_ok(const fastmc_ctx* h) {
  if (h->path == 2) return stream_lanes(h->N) == WAVE;
  if (h->path == 3) return stream_lanes(h->N) == MR_LN * h->mr_S;
  return true;
}

// Do the chirp-z ROWS of this handle run on the packed 256-point pipeline (fmc_kernels.h: k_rows_pbz)?  Windows of up to 128 pixels
// (a block of 128 inputs and the window must fit 256 points); FASTMC_PBZ=0: the one-row-per-wave chirp-z rows instead (A/B).
static bool pbz_ok(const fastmc_ctx* h) {
  static const bool off = getenv("FASTMC_PBZ") && atoi(getenv("FASTMC_PBZ")) == 0;
  return !off && h->blu_P && h->Np <= 128 && h->precision == FASTMC_F64;
}

template <class R>
static int upload_blu_tables(fastmc_ctx* h) {
  if (!h->blu_P || h->blu_lo == h->lo) return 0;
  const int P = h->blu_P, M = 64 * P, SB = h->blu_SB, B = h->blu_B;
  h->omS = (h->Np + 7) & ~7;
  const int pre_len = std::max(M, SB * B);
  std::vector<cpx<R>> tw1((size_t)P * 64), om((size_t)8 * h->omS), twf(64), pre(pre_len), vhat((size_t)SB * M), post(h->omS);
  build_tw1<R>(tw1.data(), P, cs_turns);
  build_om<R>(om.data(), h->omS, P, 0, h->Np, cs_turns);
  if (!build_blu_tables<R>(h->N, h->Np, h->lo, P, pre.data(), vhat.data(), post.data(), h->omS, twf.data(), cs_turns, B, SB, pre_len))
    return fail(FASTMC_ESTATE, "chirp-z size does not hold the window");
  TRY(upload_table<R>(&h->blu_tw1, tw1));
  TRY(upload_table<R>(&h->blu_om, om));
  TRY(upload_table<R>(&h->blu_twf, twf));
  TRY(upload_table<R>(&h->blu_pre, pre));
  TRY(upload_table<R>(&h->blu_vhat, vhat));
  TRY(upload_table<R>(&h->blu_post, post));
  if (pbz_ok(h)) {
    const int SBp = (h->N + PBZ_B - 1) / PBZ_B;
    std::vector<cpx<R>> tw(PBZ_M), ppre((size_t)(SBp + 1) * PBZ_B), pvhat((size_t)SBp * PBZ_M), ppost(128);      // (one block of zeros more: the rows fetch a block ahead)
    build_tw1_pk<R>(tw.data(), 16, cs_turns);
    if (!build_pbz_tables<R>(h->N, h->Np, h->lo, ppre.data(), pvhat.data(), ppost.data(), cs_turns))
      return fail(FASTMC_ESTATE, "packed chirp-z blocks do not hold the window");
    TRY(upload_table<R>(&h->pbz_tw, tw));
    TRY(upload_table<R>(&h->pbz_pre, ppre));
    TRY(upload_table<R>(&h->pbz_vhat, pvhat));
    TRY(upload_table<R>(&h->pbz_post, ppost));
  }
  h->blu_lo = h->lo;
  return 0;
}

template <class R>
static int upload_mr_tables(fastmc_ctx* h) {
  if (!h->mr_P || h->mr_lo == h->lo) return 0;
  const int P = h->mr_P;
  h->omS = (h->Np + 7) & ~7;
  std::vector<cpx<R>> tw1((size_t)P * 64), om((size_t)5 * h->omS);
  build_tw1_mr<R>(tw1.data(), P, cs_turns);
  build_om_mr<R>(om.data(), h->omS, P, h->lo, h->Np, cs_turns);
  TRY(upload_table<R>(&h->mr_tw1, tw1));
  TRY(upload_table<R>(&h->mr_om, om));
  if (h->mr_S > 1) {
    // X[x] = sum_s w_N^{s x} Y_s[x mod N/S]:  cw[s][oi] = w_N^{s (lo + oi)}
    std::vector<cpx<R>> cw((size_t)h->mr_S * h->omS);
    for (int sp = 0; sp < h->mr_S; ++sp)
      for (int oi = 0; oi < h->omS; ++oi) {
        double c, sn;
        cs_turns((double)(((long long)sp * (h->lo + oi)) % h->N) / h->N, &c, &sn);
        cw[(size_t)sp * h->omS + oi] = mk<R>((R)c, (R)(-sn));
      }
    TRY(upload_table<R>(&h->mr_cw, cw));
  }
  h->mr_lo = h->lo;
  return 0;
}

#if FMC_TU == 0
extern "C" int fastmc_set_pupil(fastmc_t* h, const double* W, int crop_lo, double dx) {
  if (!h || !W) return fail(FASTMC_EINVAL, "null argument");
  if (crop_lo < 0 || crop_lo + h->Np > h->N) return fail(FASTMC_EINVAL, "window [crop_lo, crop_lo+Np) outside the grid");
  HIPCHK(hipSetDevice(h->device));
  const size_t n = (size_t)h->Np * h->Np;
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) s += W[i];
  h->wsum = s;
  h->lo = crop_lo;
  h->dx = dx;
  if (!h->W) HIPCHK(hipMalloc((void**)&h->W, n * sizeof(double)));
  HIPCHK(hipMemcpy(h->W, W, n * sizeof(double), hipMemcpyHostToDevice));
  if (h->path == 2 && h->blu_P) {
    if (h->blu_lo != crop_lo) h->blu_lo = -1;
    TRY(h->precision == FASTMC_F64 ? upload_blu_tables<double>(h) : upload_blu_tables<float>(h));
  }
  if (h->mr_P) {
    if (h->mr_lo != crop_lo) h->mr_lo = -1;
    TRY(h->precision == FASTMC_F64 ? upload_mr_tables<double>(h) : upload_mr_tables<float>(h));
  }
  if (wave_supported(h->N) && h->tables_lo != crop_lo) {      // the tables depend on (N, Np, window position) only
    h->tables_lo = -1;
    TRY(h->precision == FASTMC_F64 ? upload_wave_tables<double>(h) : upload_wave_tables<float>(h));
    h->tables_lo = crop_lo;
  }
  TRY(h->precision == FASTMC_F64 ? upload_pks_tables<double>(h) : upload_pks_tables<float>(h));
  h->have_pupil = true;
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_set_subharm(fastmc_t* h, const double* ps_sh, const double* fx, const double* fy, const double* df) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  if (!ps_sh) { h->have_sh = false; return 0; }
  if (!fx || !fy || !df) return fail(FASTMC_EINVAL, "fx, fy, df required");
  if (!h->have_pupil) return fail(FASTMC_ESTATE, "call fastmc_set_pupil first (window position and dx)");
  HIPCHK(hipSetDevice(h->device));
  const int N = h->N, Np = h->Np;
  // pixel coordinates of funcs.py:229-231: arange(-D/2, D/2, dx), D = N dx
  const double D = h->dx * N;
  std::vector<double> coords(N);
  for (int j = 0; j < N; ++j) coords[j] = -D / 2 + j * h->dx;
  std::vector<double> scale(27), mu(54), ex((size_t)27 * Np * 2), ey((size_t)27 * Np * 2);
  for (int m = 0; m < 27; ++m) {
    const int lvl = m / 9;
    scale[m] = std::sqrt(ps_sh[m]) * df[lvl];
    double mxr = 0, mxi = 0, myr = 0, myi = 0;
    for (int j = 0; j < N; ++j) {
      mxr += std::cos(coords[j] * fx[m]); mxi += std::sin(coords[j] * fx[m]);
      myr += std::cos(coords[j] * fy[m]); myi += std::sin(coords[j] * fy[m]);
    }
    mxr /= N; mxi /= N; myr /= N; myi /= N;
    mu[2 * m] = mxr * myr - mxi * myi;
    mu[2 * m + 1] = mxr * myi + mxi * myr;
    for (int i = 0; i < Np; ++i) {
      const double c = coords[h->lo + i];
      ex[((size_t)m * Np + i) * 2] = std::cos(c * fx[m]);
      ex[((size_t)m * Np + i) * 2 + 1] = std::sin(c * fx[m]);
      ey[((size_t)m * Np + i) * 2] = std::cos(c * fy[m]);
      ey[((size_t)m * Np + i) * 2 + 1] = std::sin(c * fy[m]);
    }
  }
  if (!h->sh_scale) {
    TRY(dev_alloc(&h->sh_scale, 27));
    TRY(dev_alloc(&h->sh_mu, 54));
    TRY(dev_alloc(&h->sh_ex, (size_t)27 * Np * 2));
    TRY(dev_alloc(&h->sh_ey, (size_t)27 * Np * 2));
  }
  HIPCHK(hipMemcpy(h->sh_scale, scale.data(), 27 * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->sh_mu, mu.data(), 54 * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->sh_ex, ex.data(), ex.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->sh_ey, ey.data(), ey.size() * 8, hipMemcpyHostToDevice));
  h->sh_sep = true;
  for (int m = 0; m < 27; ++m) {
    const int lvl = m / 9, i = (m % 9) / 3, j = m % 3;
    if (fx[m] != fx[9 * lvl + j] || fy[m] != fy[9 * lvl + 3 * i]) h->sh_sep = false;
  }
  h->have_sh = true;
  return 0;
}
#endif

// ------------------------------------------------------------------ launches
// every launch leaves the name of its kernel on the handle (fastmc_last_kernels: bench.py prices the instruction mix of what ran)
template <class R> static const char* rname() { return sizeof(R) == 8 ? "double" : "float"; }
static int device_cus(int device) {
  static std::mutex mu;
  static std::map<int, int> cus;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cus.find(device);
  if (it != cus.end()) return it->second;
  hipDeviceProp_t prop;
  const int n = hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  cus[device] = n;
  return n;
}
// Rows per wave of a row-kernel launch: ROWS_PER_WAVE (8) when the launch is many rounds of workgroups; a SMALL launch (a chunk
// of 50 realisations at 1024^2 is 400 workgroups' worth on 256 CUs, one workgroup per CU: two rounds for 1.56 of work) takes the
// divisor that wastes the least of its last round -- 7 % per halving is what the shorter walk costs (table preloads per wave).
template <class R>
static int pick_rpw(fastmc_ctx* h, int N, int nb, int WPB) {
  constexpr int LR = 128 / (int)sizeof(cpx<R>);
  const int step = LR / std::gcd(WPB, LR);                // rows per wave come in multiples of this (whole 128-byte lines)
  const int cus = device_cus(h->device);
  const int64_t row_blocks = (N + LR - 1) / LR;
  double best = 1e300;
  int pick = ROWS_PER_WAVE;
  for (int rpw = ROWS_PER_WAVE; rpw >= step; rpw /= 2) {
    if (rpw % step) break;
    const int bpg = rpw * WPB / LR;
    const int64_t blocks = row_blocks * ((nb + bpg - 1) / bpg);
    const double cost = (double)((blocks + cus - 1) / cus) * rpw * (1.0 + 0.07 * std::log2((double)ROWS_PER_WAVE / rpw));
    if (cost < best * 0.999) { best = cost; pick = rpw; }
  }
  return pick;
}
// Workgroups of `kernel` (threads, dynamic LDS bytes) the device holds at once: what a launch whose workgroups stay and walk its
// tiles is sized to (launch_rows_wave, launch_cols_wave).  Asked of the runtime once per (kernel, LDS size).
static int resident_workgroups(fastmc_ctx* h, const void* kernel, int threads, size_t lds) {
  static std::mutex mu;
  static std::map<std::pair<const void*, size_t>, int> per_cu_of;
  int per_cu;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = per_cu_of.find({kernel, lds});
    if (it == per_cu_of.end()) {
      int n = 1;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 1;
      }
      it = per_cu_of.emplace(std::make_pair(kernel, lds), n).first;
    }
    per_cu = it->second;
  }
  return device_cus(h->device) * per_cu;
}
#define FMC_NOTE(dst, ...) snprintf(dst, sizeof(dst), __VA_ARGS__)
template <class R, int P, int NS, int MODE, int S = 1, int D = 0>
static void launch_rows_wave(fastmc_ctx* h, const RowArgs<R>& A) {
  const size_t lds = wave_lds_bytes_d<R, P, NS, D>(A.omS) + (MODE == 2 ? GEN64_TABLE_BYTES : 0);   // fits for every P = 16 variant (fused_gen64)
  hipFuncSetAttribute((const void*)k_rows_wave<R, P, NS, MODE, S, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  constexpr int WPB = WCfg<R, P, NS, D>::WPB;
  constexpr int LR = 128 / (int)sizeof(cpx<R>);
  RowArgs<R> B = A;
  B.rpw = pick_rpw<R>(h, A.N, A.nb, WPB);
  const int BPG = B.rpw * WPB / LR;
  int blocks = (A.N / LR) * ((A.nb + BPG - 1) / BPG);
  // A launch of many rounds keeps its workgroups: as many as the device holds at once walk the launch's tiles (k_rows_wave:
  // A.tiles), so the tables are staged once per CU instead of once per tile (round 5, 1024^2 float64 generator: +1.2 %,
  // profiles/r05_ab_generator_tables.txt section 6).  Whatever tile height pick_rpw chose: a chunk of 50 realisations of the
  // same-seed mode (3200 tiles of one row per wave) walks too, +1.6 % end to end there.  A launch of fewer than 8 rounds keeps one tile
  // per workgroup.  FASTMC_ROWS_PERSIST=0 switches the walk off (A/B).
  static const int persist = getenv("FASTMC_ROWS_PERSIST") ? atoi(getenv("FASTMC_ROWS_PERSIST")) : 1;
  if (persist) {
    const int resident = resident_workgroups(h, (const void*)k_rows_wave<R, P, NS, MODE, S, D>, WPB * 64, lds);
    if (blocks >= 8 * resident) { B.tiles = blocks; blocks = resident; }
  }
  hipLaunchKernelGGL((k_rows_wave<R, P, NS, MODE, S, D>), dim3(blocks), dim3(WPB * 64), lds, h->stream, B);
  FMC_NOTE(h->last_rows, "k_rows_wave<%s, %d, %d, %d, %d, %d>", rname<R>(), P, NS, MODE, S, D);
  h->clk_fresh = B.clk != nullptr;
}
template <class R, int P, int NS, int EPI, int S = 1, int D = 0>
static void launch_cols_wave(fastmc_ctx* h, const ColArgs<R>& A) {
  const size_t lds = wave_lds_bytes_cols<R, P, NS, D>(A.omS);
  hipFuncSetAttribute((const void*)k_cols_wave<R, P, NS, EPI, S, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  constexpr int WPB = WCfg<R, P, NS, D>::WPB_COLS;
  const int items = A.nb * A.Np;
  int blocks = (items + WPB - 1) / WPB;
  static const int persist = getenv("FASTMC_COLS_PERSIST") ? atoi(getenv("FASTMC_COLS_PERSIST")) : 1;
  if (persist && cols_walk<P, NS, S>()) {          // as launch_rows_wave: a launch of many rounds keeps its workgroups (each wave walks the columns in steps of the grid)
    const int resident = resident_workgroups(h, (const void*)k_cols_wave<R, P, NS, EPI, S, D>, WPB * 64, lds);
    if (blocks >= 8 * resident) blocks = resident;
  }
  hipLaunchKernelGGL((k_cols_wave<R, P, NS, EPI, S, D>), dim3(blocks), dim3(WPB * 64), lds, h->stream, A);
  FMC_NOTE(h->last_cols, "k_cols_wave<%s, %d, %d, %d, %d, %d>", rname<R>(), P, NS, EPI, S, D);
}

// rows with MODE = mode, columns with EPI = epi of one (P, NS, S, D) variant
template <class R, int P, int NS, int S, int DR, int DC = DR>
static void launch_wave_pair(fastmc_ctx* h, const RowArgs<R>& RA, const ColArgs<R>& CA, int mode, int epi) {
  {
    Span s(h, 0);
    // 256 / 512 draw 16 / 32 streams per row (fmc_core.h: stream_lanes): their device-generator rows are the packed kernels
    // (dispatch_pk) or the direct family, never the one-row-per-wave kernels, so MODE 0 is not instantiated for them; the same
    // for 768 / 1280 / 1792 (N / 16 streams per row: the packed sub-rows, staged draws or the direct family)
    if constexpr (S == 1 && (pk_grid(64 * P) || pks_grid(64 * P))) launch_rows_wave<R, P, NS, 1, S, DR>(h, RA);
    else if (mode == 0) launch_rows_wave<R, P, NS, 0, S, DR>(h, RA);
    else if (mode == 2) {
      // the float64 generator fused into the row (run_impl: fused_gen64): every P = 16 variant (1024, and 2048 / 4096 as
      // sub-rows), and the plain variant (D = 0) of the other one-row-per-wave grids (192 ... 1792)
      if constexpr (sizeof(R) == 8 && (P == 16 || (DR == 0 && S == 1 && (NS == 2 || NS == 4)))) launch_rows_wave<R, P, NS, 2, S, DR>(h, RA);
    }
    else launch_rows_wave<R, P, NS, 1, S, DR>(h, RA);
  }
  {
    Span s(h, 1);
    if (epi == 0) launch_cols_wave<R, P, NS, 0, S, DC>(h, CA);
    else launch_cols_wave<R, P, NS, 1, S, DC>(h, CA);
  }
}

// Which row / column variant serves this window (fmc_kernels.h: WCfg).
template <class R, int P, int NS, int S = 1>
static void dispatch_wave(fastmc_ctx* h, const RowArgs<R>& RA, const ColArgs<R>& CA, int mode, int epi) {
  if constexpr (P == 16 && (NS == 2 || NS == 4 || NS == 8)) {
    // 1024, and 2048 / 4096 as sub-rows: the 16 x 4 lane factorisation
    if constexpr (NS == 2) {
      const int win = window_planes(h->lo, h->Np, 16, 16);
      const bool dense = dense16r_fits<R>(h) && !h->no_dense;
      if ((win & ~D16R_CENTRE_MASK) == 0) {
        // the BASELINE geometry.  Unsplit rows: the dense sixteen-wave kernels, also with host coefficients (MODE 1: the
        // reference's own `_r` for a seed goes through the kernels that are benchmarked) and with the screens written out
        // (EPI 1).  Split rows: sixteen-wave workgroups where the tables fit (A/B at 2048^2: rows 37.4 -> 35.7 ms per 5000
        // realisations; 4096^2: -1 %); the split column kernel loses 6 % there at 4096^2 and stays on twelve waves.
        if constexpr (S == 1) {
          if constexpr (sizeof(R) == 8) {
            if (dense && mode == 1 && epi == 0 && h->light_rows) {      // same-seed mode on two streams (fastmc_run_npstream): WCfg D = 9
              { Span s(h, 0); launch_rows_wave<R, 16, 2, 1, 1, 9>(h, RA); }
              { Span s(h, 1); launch_cols_wave<R, 16, 2, 0, 1, 4>(h, CA); }
              return;
            }
          }
          if (dense) { launch_wave_pair<R, 16, 2, 1, 4>(h, RA, CA, mode, epi); return; }
        } else {
          if (dense && mode != 1) { launch_wave_pair<R, 16, 2, S, 4, 5>(h, RA, CA, mode, epi); return; }
        }
        launch_wave_pair<R, 16, 2, S, 5>(h, RA, CA, mode, epi);
        return;
      }
      if (mode != 1 && epi == 0 && (win & ~D16R_WIDE_MASK) == 0) {     // centred windows of 97-128 pixels: eight of the sixteen planes
        if constexpr (S == 1) {
          if (dense) { launch_wave_pair<R, 16, 2, 1, 8>(h, RA, CA, mode, 0); return; }
        }
        launch_wave_pair<R, 16, 2, S, 6>(h, RA, CA, mode, 0);
        return;
      }
    }
    launch_wave_pair<R, 16, NS, S, 7>(h, RA, CA, mode, epi);      // any other window: all sixteen planes
  } else {
    if constexpr (NS == 2 && P > 16 && prune_pays(P, 8, 0)) {
      if (mode == 0 && epi == 0 && (window_planes(h->lo, h->Np, P, 8) & ~centre_planes(P, 8, 0)) == 0) {
        launch_wave_pair<R, P, 2, S, 3>(h, RA, CA, 0, 0);
        return;
      }
    }
    launch_wave_pair<R, P, NS, S, 0>(h, RA, CA, mode, epi);
  }
}

// Packed rows / columns of the 256 and 512 grids (fmc_kernels.h: k_rows_pk / k_cols_pk), variant by window (pk_variant).
template <class R, int L0, int D>
static void launch_pk_pair(fastmc_ctx* h, const RowArgs<R>& RA, const ColArgs<R>& CA, int mode, int epi) {
  using C = PkCfg<R, L0, D>;
  const size_t lds = pk_lds_bytes<R, L0>(RA.omS, C::WPB) + (mode == 2 ? GEN64_TABLE_BYTES : 0), ldc = pk_lds_bytes<R, L0>(RA.omS, C::WPC);
  constexpr int LR = 128 / (int)sizeof(cpx<R>), LU = LR / C::G, BPG = ROWS_PER_WAVE * C::WPB / LU;
  const int tiles_all = (C::N / LR) * ((RA.nb + BPG - 1) / BPG);
  static const int persist = getenv("FASTMC_ROWS_PERSIST") ? atoi(getenv("FASTMC_ROWS_PERSIST")) : 1;
  auto rows = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    hipFuncSetAttribute((const void*)k_rows_pk<R, L0, MODE, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    RowArgs<R> B = RA;
    int blocks = tiles_all;
    if (persist) {        // as launch_rows_wave: a launch of many rounds keeps its workgroups, which walk its tiles (round 6)
      const int resident = resident_workgroups(h, (const void*)k_rows_pk<R, L0, MODE, D>, C::WPB * 64, lds);
      if (blocks >= 8 * resident) { B.tiles = blocks; blocks = resident; }
    }
    hipLaunchKernelGGL((k_rows_pk<R, L0, MODE, D>), dim3(blocks), dim3(C::WPB * 64), lds, h->stream, B);
    FMC_NOTE(h->last_rows, "k_rows_pk<%s, %d, %d, %d>", rname<R>(), L0, MODE, D);
  };
  {
    Span s(h, 0);
    if (mode == 0) rows(std::integral_constant<int, 0>());
    else if (mode == 2) { if constexpr (sizeof(R) == 8) rows(std::integral_constant<int, 2>()); }
    else rows(std::integral_constant<int, 1>());
  }
  {
    Span s(h, 1);
    const int items = CA.nb * ((CA.Np + C::G - 1) / C::G), per = C::WPC;
    if (epi == 0) {
      hipFuncSetAttribute((const void*)k_cols_pk<R, L0, 0, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldc);
      hipLaunchKernelGGL((k_cols_pk<R, L0, 0, D>), dim3((items + per - 1) / per), dim3(C::WPC * 64), ldc, h->stream, CA);
      FMC_NOTE(h->last_cols, "k_cols_pk<%s, %d, %d, %d>", rname<R>(), L0, 0, D);
    } else {
      hipFuncSetAttribute((const void*)k_cols_pk<R, L0, 1, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldc);
      hipLaunchKernelGGL((k_cols_pk<R, L0, 1, D>), dim3((items + per - 1) / per), dim3(C::WPC * 64), ldc, h->stream, CA);
      FMC_NOTE(h->last_cols, "k_cols_pk<%s, %d, %d, %d>", rname<R>(), L0, 1, D);
    }
  }
}
template <class R>
int dispatch_pk(fastmc_ctx* h, const RowArgs<R>& RA, const ColArgs<R>& CA, int mode, int epi) {
  const int v = pk_variant<R>(h);
  if (v < 0) return fail(FASTMC_ESTATE, "no packed-row kernel for this window");
  if (h->N == 128)      { if (v == 0) launch_pk_pair<R, 0, 0>(h, RA, CA, mode, epi); else launch_pk_pair<R, 0, 1>(h, RA, CA, mode, epi); }
  else if (h->N == 256) { if (v == 0) launch_pk_pair<R, 1, 0>(h, RA, CA, mode, epi); else launch_pk_pair<R, 1, 1>(h, RA, CA, mode, epi); }
  else                  { if (v == 0) launch_pk_pair<R, 2, 0>(h, RA, CA, mode, epi); else launch_pk_pair<R, 2, 1>(h, RA, CA, mode, epi); }
  return 0;
}

// Row pass of the packed sub-rows (fmc_kernels.h: k_rows_pks; S <= 0: the sub-row count at run time).
template <class R, int L0, int S, int MODE, int NPL = 6>
static void launch_pks_rows(fastmc_ctx* h, const RowArgs<R>& RA) {
  using C = PksCfg<R, L0, S, NPL>;
  const int Sr = S > 0 ? S : pks_count(RA.N);        // S <= 0: the sub-row count at run time (fmc_core.h: pks_rt, pks_p16)
  const size_t lds = pks_lds_bytes<R, L0, S, NPL>(Sr) + (MODE == 2 ? GEN64_TABLE_BYTES : 0);
  constexpr int LR = 128 / (int)sizeof(cpx<R>), LU = LR / C::G, BPG = ROWS_PER_WAVE * C::WPB / LU;
  int blocks = (Sr * C::M / LR) * ((RA.nb + BPG - 1) / BPG);
  RowArgs<R> B = RA;
  B.S = Sr;
  hipFuncSetAttribute((const void*)k_rows_pks<R, L0, S, MODE, NPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  static const int persist = getenv("FASTMC_ROWS_PERSIST") ? atoi(getenv("FASTMC_ROWS_PERSIST")) : 1;
  if (persist) {        // as launch_rows_wave: a launch of many rounds keeps its workgroups, which walk its tiles
    const int resident = resident_workgroups(h, (const void*)k_rows_pks<R, L0, S, MODE, NPL>, C::WPB * 64, lds);
    if (blocks >= 8 * resident) { B.tiles = blocks; blocks = resident; }
  }
  hipLaunchKernelGGL((k_rows_pks<R, L0, S, MODE, NPL>), dim3(blocks), dim3(C::WPB * 64), lds, h->stream, B);
  if (NPL == 6) FMC_NOTE(h->last_rows, "k_rows_pks<%s, %d, %d, %d>", rname<R>(), L0, S, MODE);
  else FMC_NOTE(h->last_rows, "k_rows_pks<%s, %d, %d, %d, %d>", rname<R>(), L0, S, MODE, NPL);
}
template <class R> int dispatch_pks_rows(fastmc_ctx* h, const RowArgs<R>& RA_in, int mode);
template <class R, int L0, int S, int EPI, int NPL = 6>
static void launch_pks_cols(fastmc_ctx* h, const ColArgs<R>& CA) {
  using C = PksCfg<R, L0, S, NPL>;
  constexpr int WPC = PksColCfg<R, L0, S, NPL>::WPC;
  const int Sr = S > 0 ? S : pks_count(CA.N);
  const size_t lds = pks_cols_lds_bytes<R, L0, S, NPL>(Sr);
  const int items = CA.nb * ((CA.Np + C::G - 1) / C::G);
  ColArgs<R> B = CA;
  B.S = Sr;
  hipFuncSetAttribute((const void*)k_cols_pks<R, L0, S, EPI, NPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_cols_pks<R, L0, S, EPI, NPL>), dim3((items + WPC - 1) / WPC), dim3(WPC * 64), lds, h->stream, B);
  if (NPL == 6) FMC_NOTE(h->last_cols, "k_cols_pks<%s, %d, %d, %d>", rname<R>(), L0, S, EPI);
  else FMC_NOTE(h->last_cols, "k_cols_pks<%s, %d, %d, %d, %d>", rname<R>(), L0, S, EPI, NPL);
}
// rows (device generator) and columns of the packed sub-rows: the pair keeps V permuted along ky (fmc_kernels.h: k_rows_pks)
template <class R>
int dispatch_pks(fastmc_ctx* h, const RowArgs<R>& RA_in, const ColArgs<R>& CA_in, int mode, int epi) {
  TRY(dispatch_pks_rows<R>(h, RA_in, mode));
  Span s(h, 1);
  ColArgs<R> CA = CA_in;
  CA.tw = (const cpx<R>*)h->pks_tw1; CA.cw = (const cpx<R>*)h->pks_cw;
  // (the grids of pks_rt: the kernels with a run-time count -- S = 0 odd, -2 even; float64 pipeline only, fastmc_create)
  const int S = pks_ct(h->N) ? pks_ct(h->N) : ((pks_count(h->N) & 1) ? 0 : -2), L0 = pks_L0(h->N);
  if constexpr (sizeof(R) == 8) {
    if (pks_variant<R>(h) == 1) {      // eight planes: the run-time kernels, whatever the grid
      CA.cw = (const cpx<R>*)h->pks_cw8;
      const int S8 = (pks_count(h->N) & 1) ? 0 : -2;
#define FMC_PKSC8(LL, SS) if (L0 == LL && S8 == SS) { if (epi == 0) launch_pks_cols<R, LL, SS, 0, 8>(h, CA); else launch_pks_cols<R, LL, SS, 1, 8>(h, CA); return 0; }
      FMC_PKSC8(1, 0) FMC_PKSC8(1, -2) FMC_PKSC8(0, 0) FMC_PKSC8(-1, 0)
#undef FMC_PKSC8
      return fail(FASTMC_ESTATE, "no eight-plane packed sub-row column kernel for this grid");
    }
    if (pks_variant<R>(h) == 2) {      // all sixteen planes (256-point sub-rows)
      CA.cw = (const cpx<R>*)h->pks_cw16;
      if (pks_count(h->N) & 1) { if (epi == 0) launch_pks_cols<R, 1, 0, 0, 16>(h, CA); else launch_pks_cols<R, 1, 0, 1, 16>(h, CA); }
      else { if (epi == 0) launch_pks_cols<R, 1, -2, 0, 16>(h, CA); else launch_pks_cols<R, 1, -2, 1, 16>(h, CA); }
      return 0;
    }
  }
#define FMC_PKSC(LL, SS)                                                                  \
  if (L0 == LL && S == SS) {                                                              \
    if (epi == 0) launch_pks_cols<R, LL, SS, 0>(h, CA); else launch_pks_cols<R, LL, SS, 1>(h, CA); \
    return 0;                                                                             \
  }
  FMC_PKSC(1, 3) FMC_PKSC(1, 5) FMC_PKSC(1, 6) FMC_PKSC(1, 7) FMC_PKSC(0, 3) FMC_PKSC(0, 5) FMC_PKSC(0, 7) FMC_PKSC(0, 9)
  FMC_PKSC(-1, 3) FMC_PKSC(-1, 5) FMC_PKSC(-1, 7) FMC_PKSC(-1, 9)
  if constexpr (sizeof(R) == 8) { FMC_PKSC(1, 0) FMC_PKSC(1, -2) FMC_PKSC(0, 0) FMC_PKSC(-1, 0) }
#undef FMC_PKSC
  return fail(FASTMC_ESTATE, "no packed sub-row column kernel for this grid");
}
template <class R>
int dispatch_pks_rows(fastmc_ctx* h, const RowArgs<R>& RA_in, int mode) {
  Span s(h, 0);
  const int S = pks_ct(h->N) ? pks_ct(h->N) : ((pks_count(h->N) & 1) ? 0 : -2), L0 = pks_L0(h->N);
  RowArgs<R> RA = RA_in;      // the family's own tables; the colouring tables with the input-side fftshift sign folded in
  RA.amp = (const R*)h->amp_p; RA.ampf = h->ampf_p; RA.tw = (const cpx<R>*)h->pks_tw1; RA.cw = (const cpx<R>*)h->pks_cw;
  if constexpr (sizeof(R) == 8) {
    if (pks_variant<R>(h) == 1) {      // eight planes (centred windows of 97 ... 128 pixels): the run-time kernels, whatever the grid
      RA.cw = (const cpx<R>*)h->pks_cw8;
      const int S8 = (pks_count(h->N) & 1) ? 0 : -2;
#define FMC_PKS8(LL, SS) if (L0 == LL && S8 == SS) { if (mode == 0) launch_pks_rows<R, LL, SS, 0, 8>(h, RA); else launch_pks_rows<R, LL, SS, 2, 8>(h, RA); return 0; }
      FMC_PKS8(1, 0) FMC_PKS8(1, -2) FMC_PKS8(0, 0) FMC_PKS8(-1, 0)
#undef FMC_PKS8
      return fail(FASTMC_ESTATE, "no eight-plane packed sub-row kernel for this grid");
    }
    if (pks_variant<R>(h) == 2) {      // all sixteen planes (centred windows of 129 ... 256 pixels, 256-point sub-rows)
      RA.cw = (const cpx<R>*)h->pks_cw16;
      if (pks_count(h->N) & 1) { if (mode == 0) launch_pks_rows<R, 1, 0, 0, 16>(h, RA); else launch_pks_rows<R, 1, 0, 2, 16>(h, RA); }
      else { if (mode == 0) launch_pks_rows<R, 1, -2, 0, 16>(h, RA); else launch_pks_rows<R, 1, -2, 2, 16>(h, RA); }
      return 0;
    }
  }
#define FMC_PKS(LL, SS)                                                                                   \
  if (L0 == LL && S == SS) {                                                                              \
    if (mode == 0) { launch_pks_rows<R, LL, SS, 0>(h, RA); return 0; }                                    \
    if constexpr (sizeof(R) == 8) { if (mode == 2) { launch_pks_rows<R, LL, SS, 2>(h, RA); return 0; } }  \
  }
  FMC_PKS(1, 3) FMC_PKS(1, 5) FMC_PKS(1, 6) FMC_PKS(1, 7) FMC_PKS(0, 3) FMC_PKS(0, 5) FMC_PKS(0, 7) FMC_PKS(0, 9)
  FMC_PKS(-1, 3) FMC_PKS(-1, 5) FMC_PKS(-1, 7) FMC_PKS(-1, 9)
  if constexpr (sizeof(R) == 8) { FMC_PKS(1, 0) FMC_PKS(1, -2) FMC_PKS(0, 0) FMC_PKS(-1, 0) }
#undef FMC_PKS

  return fail(FASTMC_ESTATE, "no packed sub-row kernel for this grid / mode");
}

template <class R, int P, int NS, bool BLK = false>
static void dispatch_blu_pn(fastmc_ctx* h, const RowArgs<R>& RA, const ColArgs<R>& CA, int mode, int epi, bool rows_done = false) {
  constexpr int WPB = BluCfg<R, P, NS>::WPB, WPC = BLK ? BluCfg<R, P, NS>::WPB : BluCfg<R, P, NS>::WPB_COLS;
  const size_t lds = blu_lds_bytes<R, P, NS>(RA.omS, WPB), ldc = blu_lds_bytes<R, P, NS>(RA.omS, WPC);
  constexpr int LR = 128 / (int)sizeof(cpx<R>);
  RowArgs<R> RB = RA;
  RB.rpw = pick_rpw<R>(h, RA.N, RA.nb, WPB);
  const int BPG = RB.rpw * WPB / LR;
  const int blocks = ((RA.N + LR - 1) / LR) * ((RA.nb + BPG - 1) / BPG);
  if (!rows_done) {
    Span s(h, 0);
    if (mode == 0) {
      hipFuncSetAttribute((const void*)k_rows_blu<R, P, NS, 0, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_rows_blu<R, P, NS, 0, BLK>), dim3(blocks), dim3(WPB * 64), lds, h->stream, RB);
      FMC_NOTE(h->last_rows, "k_rows_blu<%s, %d, %d, %d, %s>", rname<R>(), P, NS, 0, BLK ? "true" : "false");
    } else if (mode == 2) {
      // the float64 generator fused into the row (run_impl: fused_gen64 has checked that its tables fit)
      if constexpr (sizeof(R) == 8) {
        const size_t lds2 = lds + GEN64_TABLE_BYTES;
        hipFuncSetAttribute((const void*)k_rows_blu<R, P, NS, 2, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        hipLaunchKernelGGL((k_rows_blu<R, P, NS, 2, BLK>), dim3(blocks), dim3(WPB * 64), lds2, h->stream, RB);
        FMC_NOTE(h->last_rows, "k_rows_blu<%s, %d, %d, %d, %s>", rname<R>(), P, NS, 2, BLK ? "true" : "false");
      }
    } else {
      hipFuncSetAttribute((const void*)k_rows_blu<R, P, NS, 1, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_rows_blu<R, P, NS, 1, BLK>), dim3(blocks), dim3(WPB * 64), lds, h->stream, RB);
      FMC_NOTE(h->last_rows, "k_rows_blu<%s, %d, %d, %d, %s>", rname<R>(), P, NS, 1, BLK ? "true" : "false");
    }
  }
  {
    Span s(h, 1);
    const int items = CA.nb * CA.Np;
    if (epi == 0) {
      hipFuncSetAttribute((const void*)k_cols_blu<R, P, NS, 0, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldc);
      hipLaunchKernelGGL((k_cols_blu<R, P, NS, 0, BLK>), dim3((items + WPC - 1) / WPC), dim3(WPC * 64), ldc, h->stream, CA);
      FMC_NOTE(h->last_cols, "k_cols_blu<%s, %d, %d, %d, %s>", rname<R>(), P, NS, 0, BLK ? "true" : "false");
    } else {
      hipFuncSetAttribute((const void*)k_cols_blu<R, P, NS, 1, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldc);
      hipLaunchKernelGGL((k_cols_blu<R, P, NS, 1, BLK>), dim3((items + WPC - 1) / WPC), dim3(WPC * 64), ldc, h->stream, CA);
      FMC_NOTE(h->last_cols, "k_cols_blu<%s, %d, %d, %d, %s>", rname<R>(), P, NS, 1, BLK ? "true" : "false");
    }
  }
}

// Rows of the chirp-z grids on the packed 256-point pipeline (fmc_kernels.h: k_rows_pbz): four rows per wavefront, blocks of 128 inputs.
template <class R, int NPL, int MODE>
static void launch_pbz_rows(fastmc_ctx* h, const RowArgs<R>& RA) {
  Span s(h, 0);
  const size_t lds = pbz_lds_bytes<R>(PBZ_WPB) + (MODE == 2 ? GEN64_TABLE_BYTES : 0);
  constexpr int LR = 128 / (int)sizeof(cpx<R>), LU = LR / 4, BPG = ROWS_PER_WAVE * PBZ_WPB / LU;
  RowArgs<R> B = RA;
  B.tw = (const cpx<R>*)h->pbz_tw;
  B.blu.pre = (const cpx<R>*)h->pbz_pre; B.blu.vhat = (const cpx<R>*)h->pbz_vhat; B.blu.post = (const cpx<R>*)h->pbz_post;
  B.blu.SB = (RA.N + PBZ_B - 1) / PBZ_B; B.blu.B = PBZ_B;
  int blocks = ((RA.N + LR - 1) / LR) * ((RA.nb + BPG - 1) / BPG);
  hipFuncSetAttribute((const void*)k_rows_pbz<R, NPL, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  static const int persist = getenv("FASTMC_ROWS_PERSIST") ? atoi(getenv("FASTMC_ROWS_PERSIST")) : 1;
  if (persist) {        // as launch_rows_wave: a launch of many rounds keeps its workgroups, which walk its tiles
    const int resident = resident_workgroups(h, (const void*)k_rows_pbz<R, NPL, MODE>, PBZ_WPB * 64, lds);
    if (blocks >= 8 * resident) { B.tiles = blocks; blocks = resident; }
  }
  hipLaunchKernelGGL((k_rows_pbz<R, NPL, MODE>), dim3(blocks), dim3(PBZ_WPB * 64), lds, h->stream, B);
  FMC_NOTE(h->last_rows, "k_rows_pbz<%s, %d, %d>", rname<R>(), NPL, MODE);
}
template <class R, int NPL, int EPI>
static void launch_pbz_cols(fastmc_ctx* h, const ColArgs<R>& CA) {
  Span s(h, 1);
  const size_t lds = pbz_lds_bytes<R>(PBZ_WPC);
  ColArgs<R> B = CA;
  B.tw = (const cpx<R>*)h->pbz_tw;
  B.blu.pre = (const cpx<R>*)h->pbz_pre; B.blu.vhat = (const cpx<R>*)h->pbz_vhat; B.blu.post = (const cpx<R>*)h->pbz_post;
  B.blu.SB = (CA.N + PBZ_B - 1) / PBZ_B; B.blu.B = PBZ_B;
  const int items = CA.nb * ((CA.Np + 3) / 4);
  hipFuncSetAttribute((const void*)k_cols_pbz<R, NPL, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_cols_pbz<R, NPL, EPI>), dim3((items + PBZ_WPC - 1) / PBZ_WPC), dim3(PBZ_WPC * 64), lds, h->stream, B);
  FMC_NOTE(h->last_cols, "k_cols_pbz<%s, %d, %d>", rname<R>(), NPL, EPI);
}
template <class R>
int dispatch_blu(fastmc_ctx* h, const RowArgs<R>& RA, const ColArgs<R>& CA, int mode, int epi) {
  const int ns = h->NS <= 2 ? 2 : 4;
  bool rows_done = false;
  if constexpr (sizeof(R) == 8) {
    if (pbz_ok(h)) {         // windows of up to 128 pixels: rows and columns on the packed pipeline
      static const bool cols_too = !(getenv("FASTMC_PBZ_COLS") && atoi(getenv("FASTMC_PBZ_COLS")) == 0);      // A/B: k_cols_blu instead
      if (h->Np <= 96) { if (mode == 0) launch_pbz_rows<R, 6, 0>(h, RA); else if (mode == 2) launch_pbz_rows<R, 6, 2>(h, RA); else launch_pbz_rows<R, 6, 1>(h, RA); }
      else { if (mode == 0) launch_pbz_rows<R, 8, 0>(h, RA); else if (mode == 2) launch_pbz_rows<R, 8, 2>(h, RA); else launch_pbz_rows<R, 8, 1>(h, RA); }
      rows_done = true;
      if (cols_too) {
        if (h->Np <= 96) { if (epi == 0) launch_pbz_cols<R, 6, 0>(h, CA); else launch_pbz_cols<R, 6, 1>(h, CA); }
        else { if (epi == 0) launch_pbz_cols<R, 8, 0>(h, CA); else launch_pbz_cols<R, 8, 1>(h, CA); }
        return 0;
      }
    }
  }
  if (h->blu_SB > 1) {       // rows in input blocks on the M = 1024 pipeline
    if (ns == 2) dispatch_blu_pn<R, 16, 2, true>(h, RA, CA, mode, epi, rows_done);
    else dispatch_blu_pn<R, 16, 4, true>(h, RA, CA, mode, epi, rows_done);
    return 0;
  }
#define FMC_BLU(PP, NN) if (h->blu_P == PP && ns == NN) { dispatch_blu_pn<R, PP, NN>(h, RA, CA, mode, epi, rows_done); return 0; }
  FMC_BLU(4, 2) FMC_BLU(8, 2) FMC_BLU(8, 4) FMC_BLU(16, 2) FMC_BLU(16, 4) FMC_BLU(24, 2) FMC_BLU(24, 4) FMC_BLU(32, 2)
  FMC_BLU(12, 2) FMC_BLU(28, 2)
#undef FMC_BLU
  return fail(FASTMC_ESTATE, "no chirp-z instantiation for this grid / window");
}

template <class R, int P, int NS, bool SPLIT, int LN, int PR>
static void launch_mr(fastmc_ctx* h, const RowArgs<R>& RA, const ColArgs<R>& CA, int mode, int epi) {
  constexpr int WPB = MrCfg<R, P, NS, LN>::WPB;
  const size_t lds = mr_lds_bytes<R, P, NS, LN>(RA.omS);
  constexpr int LR = 128 / (int)sizeof(cpx<R>);
  RowArgs<R> RB = RA;
  RB.rpw = pick_rpw<R>(h, RA.N, RA.nb, WPB);
  const int BPG = RB.rpw * WPB / LR;
  const int blocks = ((RA.N + LR - 1) / LR) * ((RA.nb + BPG - 1) / BPG);
  {
    Span s(h, 0);
    // (LN = 64: the run-time-split grids are multiples of 64, whose device draws go to the packed sub-rows, staged draws or the direct
    // family -- they draw N / 16 or N / 8 streams per row -- so their rows exist for host coefficients only; run_impl never asks for more)
    if (mode == 0) {
      if constexpr (LN != WAVE) {
      hipFuncSetAttribute((const void*)k_rows_mr<R, P, NS, 0, SPLIT, LN, PR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_rows_mr<R, P, NS, 0, SPLIT, LN, PR>), dim3(blocks), dim3(WPB * 64), lds, h->stream, RB);
      FMC_NOTE(h->last_rows, "k_rows_mr<%s, %d, %d, %d, %s, %d, %d>", rname<R>(), P, NS, 0, SPLIT ? "true" : "false", LN, PR);
      }
    } else if (mode == 2) {
      // the float64 generator fused into the row (run_impl: fused_gen64 has checked that its tables fit)
      if constexpr (PR == 0 && sizeof(R) == 8 && LN != WAVE) {
        const size_t lds2 = lds + GEN64_TABLE_BYTES;
        hipFuncSetAttribute((const void*)k_rows_mr<R, P, NS, 2, SPLIT, LN, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        hipLaunchKernelGGL((k_rows_mr<R, P, NS, 2, SPLIT, LN, 0>), dim3(blocks), dim3(WPB * 64), lds2, h->stream, RB);
        FMC_NOTE(h->last_rows, "k_rows_mr<%s, %d, %d, %d, %s, %d, %d>", rname<R>(), P, NS, 2, SPLIT ? "true" : "false", LN, 0);
      }
    } else if constexpr (PR == 0) {
      hipFuncSetAttribute((const void*)k_rows_mr<R, P, NS, 1, SPLIT, LN, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_rows_mr<R, P, NS, 1, SPLIT, LN, 0>), dim3(blocks), dim3(WPB * 64), lds, h->stream, RB);
      FMC_NOTE(h->last_rows, "k_rows_mr<%s, %d, %d, %d, %s, %d, %d>", rname<R>(), P, NS, 1, SPLIT ? "true" : "false", LN, 0);
    }
  }
  {
    Span s(h, 1);
    const int items = CA.nb * CA.Np;
    if (epi == 0) {
      hipFuncSetAttribute((const void*)k_cols_mr<R, P, NS, 0, SPLIT, LN, PR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_cols_mr<R, P, NS, 0, SPLIT, LN, PR>), dim3((items + WPB - 1) / WPB), dim3(WPB * 64), lds, h->stream, CA);
      FMC_NOTE(h->last_cols, "k_cols_mr<%s, %d, %d, %d, %s, %d, %d>", rname<R>(), P, NS, 0, SPLIT ? "true" : "false", LN, PR);
    } else if constexpr (PR == 0) {
      hipFuncSetAttribute((const void*)k_cols_mr<R, P, NS, 1, SPLIT, LN, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_cols_mr<R, P, NS, 1, SPLIT, LN, 0>), dim3((items + WPB - 1) / WPB), dim3(WPB * 64), lds, h->stream, CA);
      FMC_NOTE(h->last_cols, "k_cols_mr<%s, %d, %d, %d, %s, %d, %d>", rname<R>(), P, NS, 1, SPLIT ? "true" : "false", LN, 0);
    }
  }
}

// Pruned-plane variants (device generator + detector, NS = 2, window inside the centred 96-pixel plane set): centre block of
// residue 0 for 64 P S and for 50 P S with S even, 5 for 50 P S with S odd.
template <class R, int P, int NS, bool SPLIT, int LN = MR_LN>
static void dispatch_mr_pn(fastmc_ctx* h, const RowArgs<R>& RA, const ColArgs<R>& CA, int mode, int epi) {
  constexpr int L1 = LN == MR_LN ? 10 : 8;
  if constexpr (NS == 2 && P >= 16 && LN == MR_LN) {      // (64 lanes: host-coefficient rows only -- launch_mr)
    if (mode == 0 && epi == 0) {
      const int S = RA.N / (LN * P);
      const int win = window_planes(h->lo, h->Np, P, L1);
      if constexpr (LN == MR_LN && prune_pays(P, 10, 5)) {
        if ((S & 1) && (win & ~centre_planes(P, 10, 5)) == 0) { launch_mr<R, P, NS, SPLIT, LN, 1>(h, RA, CA, mode, epi); return; }
      }
      if constexpr (SPLIT && prune_pays(P, L1, 0)) {
        if ((LN != MR_LN || !(S & 1)) && (win & ~centre_planes(P, L1, 0)) == 0) { launch_mr<R, P, NS, SPLIT, LN, 2>(h, RA, CA, mode, epi); return; }
      }
    }
  }
  launch_mr<R, P, NS, SPLIT, LN, 0>(h, RA, CA, mode, epi);
}

// P of the split grids (mr_split: the smallest S that leaves 7 <= P <= 24)
constexpr bool mr_split_P(int P) { return P == 7 || P == 9 || P == 10 || P == 14 || P == 16 || P == 18 || P == 20 || P == 24; }

// PART 0: P <= 9, PART 1: P >= 10 (two translation units)
template <class R, int PART>
int dispatch_mr_part(fastmc_ctx* h, const RowArgs<R>& RA, const ColArgs<R>& CA, int mode, int epi) {
  const int ns = h->NS <= 2 ? 2 : 4;
  const bool split = h->mr_S > 1;
#define FMC_MR(PP)                                                                                                    \
  if (h->mr_P == PP) {                                                                                                \
    if (!split) {                                                                                                     \
      if (ns == 2) { dispatch_mr_pn<R, PP, 2, false>(h, RA, CA, mode, epi); return 0; }                               \
      if constexpr (mr_has_ns4(PP)) { if (ns == 4) { dispatch_mr_pn<R, PP, 4, false>(h, RA, CA, mode, epi); return 0; } } \
    } else if constexpr (mr_split_P(PP)) {                                                                            \
      if (ns == 2) { dispatch_mr_pn<R, PP, 2, true>(h, RA, CA, mode, epi); return 0; }                                \
      if constexpr (mr_has_ns4(PP)) { if (ns == 4) { dispatch_mr_pn<R, PP, 4, true>(h, RA, CA, mode, epi); return 0; } }  \
    }                                                                                                                 \
  }
  if constexpr (PART == 0) {
    FMC_MR(2) FMC_MR(3) FMC_MR(4) FMC_MR(5) FMC_MR(6) FMC_MR(7) FMC_MR(8) FMC_MR(9)
  } else {
    FMC_MR(10) FMC_MR(12) FMC_MR(14) FMC_MR(16) FMC_MR(18) FMC_MR(20) FMC_MR(24)
  }
#undef FMC_MR
  return fail(FASTMC_ESTATE, "no 50-lane instantiation for this grid / window");
}

// Wave-family grids with a run-time sub-row count (fmc_core.h: wave_rt_split): the split kernels of the 50-lane family on
// the 64-lane pipeline; tables as for every wave size (upload_wave_tables with P = h->P, S = h->S).
template <class R>
int dispatch_ws(fastmc_ctx* h, const RowArgs<R>& RA, const ColArgs<R>& CA, int mode, int epi) {
  const int ns = h->NS <= 2 ? 2 : 4;
#define FMC_WS(PP)                                                                                                          \
  if (h->P == PP) {                                                                                                         \
    if (ns == 2) { dispatch_mr_pn<R, PP, 2, true, WAVE>(h, RA, CA, mode, epi); return 0; }                                   \
    if constexpr (has_ns4(PP)) { if (ns == 4) { dispatch_mr_pn<R, PP, 4, true, WAVE>(h, RA, CA, mode, epi); return 0; } }     \
  }
  FMC_WS(7) FMC_WS(9) FMC_WS(10) FMC_WS(14) FMC_WS(16) FMC_WS(18) FMC_WS(20) FMC_WS(24)      // (16: 7168 and 8192 only)
#undef FMC_WS
  return fail(FASTMC_ESTATE, "no run-time-split instantiation for this grid / window");
}

// N = 2048 (S = 2) and 4096 (S = 4): sub-rows of 1024 points through the P = 16 pipeline
template <class R, int S>
static int dispatch_wave_split(fastmc_ctx* h, const RowArgs<R>& RA, const ColArgs<R>& CA, int mode, int epi) {
  const int ns = pick_ns<R, 16>(h);
  if (ns == 2) dispatch_wave<R, 16, 2, S>(h, RA, CA, mode, epi);
  else if (ns == 4) dispatch_wave<R, 16, 4, S>(h, RA, CA, mode, epi);
  else if (ns == 8) dispatch_wave<R, 16, 8, S>(h, RA, CA, mode, epi);
  else return fail(FASTMC_ESTATE, "no split-row instantiation for this window");
  return 0;
}

template <class R, int P>
static int dispatch_wave_ns(fastmc_ctx* h, const RowArgs<R>& RA, const ColArgs<R>& CA, int mode, int epi) {
  // up to three instantiations per (R, P): windows up to 128 pixels, up to 256, and the general case
  const int ns = pick_ns<R, P>(h);
  if (ns == 2) dispatch_wave<R, P, 2>(h, RA, CA, mode, epi);
  else if (ns == 4 && has_ns4(P)) {
    if constexpr (has_ns4(P)) dispatch_wave<R, P, 4>(h, RA, CA, mode, epi);
  } else if (ns == 8 && P == 16) {
    if constexpr (P == 16) dispatch_wave<R, 16, 8>(h, RA, CA, mode, epi);
  } else if (ns == P) {
    if constexpr (is_pow2(P)) dispatch_wave<R, P, P>(h, RA, CA, mode, epi);
  }
  else return fail(FASTMC_ESTATE, "no wave instantiation for this window");
  return 0;
}

// The launches of the wave family in three parts, each an explicit instantiation in its own translation unit:
//   PART 0: P <= 12;   PART 1: P = 16 and the 2048 / 4096 sub-rows;   PART 2: P = 14, 18, 20, 24, 28, 32 (the general 2048 window)
template <class R, int PART>
int dispatch_wave_part(fastmc_ctx* h, RowArgs<R>& RA, ColArgs<R>& CA, int mode, int epi, bool general_2048) {
  if constexpr (PART == 1) {
    if (h->S == 2) return dispatch_wave_split<R, 2>(h, RA, CA, mode, epi);
    if (h->S == 4) return dispatch_wave_split<R, 4>(h, RA, CA, mode, epi);
    return dispatch_wave_ns<R, 16>(h, RA, CA, mode, epi);
  } else if constexpr (PART == 0) {
    switch (h->P) {
      // P = 2, 4 (128, 256) and the windows of up to 256 pixels at P = 8 (512) belong to the packed rows (dispatch_pk): only
      // the whole-grid window of 512 is left to the one-row-per-wave kernels, with host coefficients (run_impl)
      case 3: dispatch_wave<R, 3, 2>(h, RA, CA, mode, epi); break;
      case 5: dispatch_wave<R, 5, 2>(h, RA, CA, mode, epi); break;
      case 6: dispatch_wave<R, 6, 2>(h, RA, CA, mode, epi); break;
      case 7: dispatch_wave<R, 7, 2>(h, RA, CA, mode, epi); break;
      case 8:
        if (pick_ns<R, 8>(h) != 8) return fail(FASTMC_ESTATE, "no wave instantiation for this window");
        dispatch_wave<R, 8, 8>(h, RA, CA, mode, epi);
        break;
      case 9: TRY((dispatch_wave_ns<R, 9>(h, RA, CA, mode, epi))); break;
      case 10: TRY((dispatch_wave_ns<R, 10>(h, RA, CA, mode, epi))); break;
      case 12: TRY((dispatch_wave_ns<R, 12>(h, RA, CA, mode, epi))); break;
      default: return fail(FASTMC_ESTATE, "no wave instantiation for this grid size");
    }
  } else {
    if (general_2048) {
      RA.om = (const cpx<R>*)h->omg;
      CA.om = RA.om;
      dispatch_wave<R, 32, 32>(h, RA, CA, mode, epi);
      return 0;
    }
    switch (h->P) {
      case 14: dispatch_wave<R, 14, 2>(h, RA, CA, mode, epi); break;
      case 18: dispatch_wave<R, 18, 2>(h, RA, CA, mode, epi); break;
      case 20: TRY((dispatch_wave_ns<R, 20>(h, RA, CA, mode, epi))); break;
      case 24: TRY((dispatch_wave_ns<R, 24>(h, RA, CA, mode, epi))); break;
      case 28: dispatch_wave<R, 28, 2>(h, RA, CA, mode, epi); break;
      default: return fail(FASTMC_ESTATE, "no wave instantiation for this grid size");
    }
  }
  return 0;
}

template <class R>
int dispatch_direct(fastmc_ctx* h, const RowArgs<R>& RA_in, const ColArgs<R>& CA_in, int mode, int epi) {
  // LDS: twiddles [N] + row/column [N] + segment partials [S][Np] (S*Np <= max(256, Np))
  size_t lds = ((size_t)2 * h->N + (size_t)std::max(DIRECT_THREADS, h->Np)) * sizeof(cpx<R>);
  RowArgs<R> RA = RA_in;
  ColArgs<R> CA = CA_in;
  RA.tw_global = CA.tw_global = 0;
  if (lds > 160 * 1024 - 4096) {
    // keep the N twiddles in global memory (L2-resident) and only the row / column in the LDS
    lds -= (size_t)h->N * sizeof(cpx<R>);
    RA.tw_global = CA.tw_global = 1;
  }
  if (lds > 160 * 1024 - 4096) return fail(FASTMC_EINVAL, "N too large for the direct kernels at this precision");
  {
    Span s(h, 0);
    if (mode == 0) {
      hipFuncSetAttribute((const void*)k_rows_direct<R, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_rows_direct<R, 0>), dim3(RA.nb * h->N), dim3(DIRECT_THREADS), lds, h->stream, RA);
      FMC_NOTE(h->last_rows, "k_rows_direct<%s, %d>", rname<R>(), 0);
    } else {
      hipFuncSetAttribute((const void*)k_rows_direct<R, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_rows_direct<R, 1>), dim3(RA.nb * h->N), dim3(DIRECT_THREADS), lds, h->stream, RA);
      FMC_NOTE(h->last_rows, "k_rows_direct<%s, %d>", rname<R>(), 1);
    }
  }
  {
    Span s(h, 1);
    if (epi == 0) {
      hipFuncSetAttribute((const void*)k_cols_direct<R, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_cols_direct<R, 0>), dim3(CA.nb * h->Np), dim3(DIRECT_THREADS), lds, h->stream, CA);
      FMC_NOTE(h->last_cols, "k_cols_direct<%s, %d>", rname<R>(), 0);
    } else {
      hipFuncSetAttribute((const void*)k_cols_direct<R, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_cols_direct<R, 1>), dim3(CA.nb * h->Np), dim3(DIRECT_THREADS), lds, h->stream, CA);
      FMC_NOTE(h->last_cols, "k_cols_direct<%s, %d>", rname<R>(), 1);
    }
  }
  return 0;
}

// ------------------------------------------------------------------ translation units (see the top of the file)
#define FMC_FAMILY_SIG(R) fastmc_ctx*, const RowArgs<R>&, const ColArgs<R>&, int, int
#define FMC_PKS_SIG(R) fastmc_ctx*, const RowArgs<R>&, int
#define FMC_WAVE_SIG(R) fastmc_ctx*, RowArgs<R>&, ColArgs<R>&, int, int, bool
// unit -> what it instantiates.  The float32 pipeline exists for the wave and direct families; chirp-z, 50-lane and
// run-time-split grids run the float64 kernels whatever precision was asked for (fastmc_create).
//   1 / 2 / 3: wave family float64, parts 0 / 1 / 2        4 / 5 / 6: wave family float32, parts 0 / 1 / 2
//   7: chirp-z float64, direct float64 and float32          8: 50-lane P <= 9 and run-time-split, float64     9: 50-lane P >= 10
//   10: packed rows of the 256 / 512 grids, both precisions
#if defined(FMC_SPLIT_BUILD) || FMC_TU != 0
// declared `extern` in every unit but the one that defines it
#if FMC_TU != 1
extern template int dispatch_wave_part<double, 0>(FMC_WAVE_SIG(double));
#endif
#if FMC_TU != 2
extern template int dispatch_wave_part<double, 1>(FMC_WAVE_SIG(double));
#endif
#if FMC_TU != 3
extern template int dispatch_wave_part<double, 2>(FMC_WAVE_SIG(double));
#endif
#if FMC_TU != 4
extern template int dispatch_wave_part<float, 0>(FMC_WAVE_SIG(float));
#endif
#if FMC_TU != 5
extern template int dispatch_wave_part<float, 1>(FMC_WAVE_SIG(float));
#endif
#if FMC_TU != 6
extern template int dispatch_wave_part<float, 2>(FMC_WAVE_SIG(float));
#endif
#if FMC_TU != 7
extern template int dispatch_blu<double>(FMC_FAMILY_SIG(double));
extern template int dispatch_direct<double>(FMC_FAMILY_SIG(double));
extern template int dispatch_direct<float>(FMC_FAMILY_SIG(float));
#endif
#if FMC_TU != 8
extern template int dispatch_mr_part<double, 0>(FMC_FAMILY_SIG(double));
extern template int dispatch_ws<double>(FMC_FAMILY_SIG(double));
#endif
#if FMC_TU != 9
extern template int dispatch_mr_part<double, 1>(FMC_FAMILY_SIG(double));
#endif
#if FMC_TU != 10
extern template int dispatch_pk<double>(FMC_FAMILY_SIG(double));
extern template int dispatch_pk<float>(FMC_FAMILY_SIG(float));
extern template int dispatch_pks<double>(FMC_FAMILY_SIG(double));
extern template int dispatch_pks<float>(FMC_FAMILY_SIG(float));
#endif
#endif
#if FMC_TU == 1
template int dispatch_wave_part<double, 0>(FMC_WAVE_SIG(double));
#elif FMC_TU == 2
template int dispatch_wave_part<double, 1>(FMC_WAVE_SIG(double));
#elif FMC_TU == 3
template int dispatch_wave_part<double, 2>(FMC_WAVE_SIG(double));
#elif FMC_TU == 7
template int dispatch_blu<double>(FMC_FAMILY_SIG(double));
template int dispatch_direct<double>(FMC_FAMILY_SIG(double));
#ifndef FMC_ONLY_F64
template int dispatch_direct<float>(FMC_FAMILY_SIG(float));
#endif
#elif FMC_TU == 8
template int dispatch_mr_part<double, 0>(FMC_FAMILY_SIG(double));
template int dispatch_ws<double>(FMC_FAMILY_SIG(double));
#elif FMC_TU == 9
template int dispatch_mr_part<double, 1>(FMC_FAMILY_SIG(double));
#elif FMC_TU == 10
template int dispatch_pk<double>(FMC_FAMILY_SIG(double));
template int dispatch_pks<double>(FMC_FAMILY_SIG(double));
#ifndef FMC_ONLY_F64
template int dispatch_pk<float>(FMC_FAMILY_SIG(float));
template int dispatch_pks<float>(FMC_FAMILY_SIG(float));
#endif
#elif !defined(FMC_ONLY_F64)
#if FMC_TU == 4
template int dispatch_wave_part<float, 0>(FMC_WAVE_SIG(float));
#elif FMC_TU == 5
template int dispatch_wave_part<float, 1>(FMC_WAVE_SIG(float));
#elif FMC_TU == 6
template int dispatch_wave_part<float, 2>(FMC_WAVE_SIG(float));
#endif
#endif

#if FMC_TU == 0   // everything below: the run loop, the remaining entry points and the small kernels' launches
struct RunSpec {
  int mode;                 // 0 device RNG, 1 host coefficients
  int epi;                  // 0 powers, 1 screens
  uint64_t seed;
  int64_t real0, n_real;
  const double* coeff_re;   // host
  const double* coeff_im;
  const double* sh_re;
  const double* sh_im;
  const double* logamp;     // host, 2*n_real
  double logamp_var;
  int coherent;
  double* out;              // host
  double* phs;              // host
  bool async = false;       // fastmc_run_async: no host copy, no wait
  // coefficients drawn on the device (numpy stream): called per batch to fill cre / cim (and the sub-harmonic inputs) for
  // realisations [bs, bs + nb) of this run instead of the uploads of mode 1
  std::function<int(int64_t, int)> fill;
  const double* logamp_dev = nullptr;   // log-amplitudes already on the device, in output order, scaled
  // ... or the whole run's coefficients already on the device (numpy stream, one-pass generator): [n_real][N^2] each, and the
  // sub-harmonic inputs [n_real][27] each; batches read them in place
  const double* coef_dev_re = nullptr;
  const double* coef_dev_im = nullptr;
  const double* sh_dev_re = nullptr;
  const double* sh_dev_im = nullptr;
  double* out_dev = nullptr;            // write the results here instead of h->out (the chunks of fastmc_run_npstream: one copy back per call, not per chunk)
};

// Does this handle's row kernel draw the float64 generator itself (MODE 2)?  The one-row-per-wave grids of the wave family
// (192 ... 1792: the plain variant, windows up to 256 pixels), the packed rows, and the P = 16 rows: 1024, and
// 2048 / 4096 as sub-rows, float64 pipeline, any window the family serves (every variant's tables + the 4 KB of generator tables fit the
// LDS: 152 KB + 64 omS <= 160 KB for the sixteen-wave variants, whose omS <= 128; 128 KB + 64 omS for the twelve-wave ones, omS <= 512).
template <class R>
static bool fused_gen64(fastmc_ctx* h) {
  if constexpr (sizeof(R) != 8) return false;
  if (getenv("FASTMC_GEN64_STAGED")) return false;          // A/B: the round-3 form (k_gen_coeffs_f64 -> cre / cim -> MODE 1 rows)
  if (h->path == 1 && pk_grid(h->N) && pk_variant<R>(h) >= 0) return true;      // 128 / 256 / 512: the packed rows draw it themselves too
  if (h->path != 0 && pks_grid(h->N)) return pks_variant<R>(h) >= 0;            // 192 ... 3968 (pks_split): the packed sub-rows, else staged
  if (pks_p16(h->N) && pks_variant<R>(h) >= 0) return true;                     // 1024 / 2048 / 4096 where the packed sub-rows serve them
  if constexpr (sizeof(R) == 8) {
    if (!family_streams_ok(h)) return false;         // (a forced family whose rows draw another stream layout: staged)
    if (h->path == 2 && pbz_ok(h)) return true;      // chirp-z rows on the packed pipeline draw it themselves
    if (h->path == 2) {      // chirp-z family: its rows draw it too where the tables fit
      const int ns2 = h->NS <= 2 ? 2 : 4;
      const int omS = h->omS;
#define FMC_G64_BLU(PP, NN) if (h->blu_P == PP && ns2 == NN) return blu_lds_bytes<R, PP, NN>(omS, BluCfg<R, PP, NN>::WPB) + GEN64_TABLE_BYTES <= LDS_MAX;
      if (h->blu_SB > 1) { FMC_G64_BLU(16, 2) FMC_G64_BLU(16, 4) return false; }
      FMC_G64_BLU(4, 2) FMC_G64_BLU(8, 2) FMC_G64_BLU(8, 4) FMC_G64_BLU(16, 2) FMC_G64_BLU(16, 4) FMC_G64_BLU(24, 2) FMC_G64_BLU(24, 4) FMC_G64_BLU(32, 2)
      FMC_G64_BLU(12, 2) FMC_G64_BLU(28, 2)
#undef FMC_G64_BLU
      return false;
    }
    // 50-lane family (path 3) and the run-time-split wave grids: the kernels of fmc_mrfft.h, where their tables + 4 KB fit
    const bool ws = h->path == 1 && wave_rt_split(h->N);
    if (h->path == 3 || ws) {
      const int ns2 = h->NS <= 2 ? 2 : 4, PP_ = ws ? h->P : h->mr_P;
      const int omS = h->omS;
#define FMC_G64_MR(PP, LNN)                                                                                     \
      if (PP_ == PP) {                                                                                          \
        if (ns2 == 2) return mr_lds_bytes<R, PP, 2, LNN>(omS) + GEN64_TABLE_BYTES <= LDS_MAX;                   \
        if constexpr (LNN == MR_LN ? mr_has_ns4(PP) : has_ns4(PP)) return mr_lds_bytes<R, PP, 4, LNN>(omS) + GEN64_TABLE_BYTES <= LDS_MAX; \
        return false;                                                                                           \
      }
      if (ws) { FMC_G64_MR(7, WAVE) FMC_G64_MR(9, WAVE) FMC_G64_MR(10, WAVE) FMC_G64_MR(14, WAVE) FMC_G64_MR(16, WAVE) FMC_G64_MR(18, WAVE) FMC_G64_MR(20, WAVE) FMC_G64_MR(24, WAVE) }
      else { FMC_G64_MR(2, MR_LN) FMC_G64_MR(3, MR_LN) FMC_G64_MR(4, MR_LN) FMC_G64_MR(5, MR_LN) FMC_G64_MR(6, MR_LN) FMC_G64_MR(7, MR_LN) FMC_G64_MR(8, MR_LN)
             FMC_G64_MR(9, MR_LN) FMC_G64_MR(10, MR_LN) FMC_G64_MR(12, MR_LN) FMC_G64_MR(14, MR_LN) FMC_G64_MR(16, MR_LN) FMC_G64_MR(18, MR_LN) FMC_G64_MR(20, MR_LN) FMC_G64_MR(24, MR_LN) }
#undef FMC_G64_MR
      return false;
    }
  }
  if (h->path != 1 || wave_rt_split(h->N) || pk_grid(h->N)) return false;
  int ns = 0, wpb = 0;
  wave_config<R>(h, &ns, &wpb);
  if (h->P == 16) {      // not the whole-grid window (NS = P: its tables leave no room, and nothing draws into it)
    // every variant dispatch_wave can pick for this window must hold the generator's tables beside its own (they all do for the
    // windows the family serves -- sixteen waves: 153 KB + 48 omS, omS <= 128; twelve waves: 133 KB + 48 omS, omS <= 512 -- but
    // the launch must never be the place that finds out)
    if constexpr (sizeof(R) == 8) {
      const size_t g = GEN64_TABLE_BYTES;
      if (ns == 2) return wave_lds_bytes_d<R, 16, 2, 7>(h->omS) + g <= LDS_MAX && (!dense16r_fits<R>(h) || wave_lds_bytes_d<R, 16, 2, 8>(h->omS) + g <= LDS_MAX);
      if (ns == 4) return wave_lds_bytes_d<R, 16, 4, 7>(h->omS) + g <= LDS_MAX;
      if (ns == 8) return wave_lds_bytes_d<R, 16, 8, 7>(h->omS) + g <= LDS_MAX;
    }
    return false;
  }
  // the other one-row-per-wave grids (192 ... 1792, S = 1): windows of up to 256 pixels, where the generator's 4 KB of tables fit the LDS
  if (h->S != 1 || (ns != 2 && ns != 4)) return false;
  if constexpr (sizeof(R) == 8) {
    auto fits = [&](auto tag) {
      constexpr int PP = decltype(tag)::value;
      if (ns == 2) return wave_lds_bytes_d<R, PP, 2, 0>(h->omS) + GEN64_TABLE_BYTES <= LDS_MAX;
      if constexpr (has_ns4(PP)) return wave_lds_bytes_d<R, PP, 4, 0>(h->omS) + GEN64_TABLE_BYTES <= LDS_MAX;
      return false;
    };
    switch (h->P) {
      case 3: return fits(std::integral_constant<int, 3>());
      case 5: return fits(std::integral_constant<int, 5>());
      case 6: return fits(std::integral_constant<int, 6>());
      case 7: return fits(std::integral_constant<int, 7>());
      case 9: return fits(std::integral_constant<int, 9>());
      case 10: return fits(std::integral_constant<int, 10>());
      case 12: return fits(std::integral_constant<int, 12>());
      case 14: return fits(std::integral_constant<int, 14>());
      case 18: return fits(std::integral_constant<int, 18>());
      case 20: return fits(std::integral_constant<int, 20>());
      case 24: return fits(std::integral_constant<int, 24>());
      case 28: return fits(std::integral_constant<int, 28>());
      default: return false;
    }
  }
  return false;
}

template <class R>
static int run_impl(fastmc_ctx* h, const RunSpec& S) {
  const int N = h->N, Np = h->Np;
  const size_t N2 = (size_t)N * N;
  int B = default_batch(h);
  // device generator at float64 precision: the draws of a batch are staged like uploaded coefficients (16 B each), <= 2 GiB
  // ... fused into the row kernels where they have the form (MODE 2: no coefficient passes through HBM), staged otherwise
  const bool fused64 = S.mode == 0 && h->rng_f64 && fused_gen64<R>(h);
  const bool gen64 = S.mode == 0 && h->rng_f64 && !fused64;
  const int kmode = fused64 ? 2 : ((S.mode == 1 || gen64) ? 1 : 0);           // MODE of the row kernels
  if (!h->clk) {       // four words for the row kernel's clock stamps (fastmc_last_clock)
    HIPCHK(hipMalloc((void**)&h->clk, 4 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(h->clk, 0, 4 * sizeof(unsigned long long), h->stream));
  }
  // host coefficients: 256 MB per upload; coefficients drawn on the device (S.fill): 2 GiB per array -- fewer, fuller launches
  const bool devcoef = S.mode == 1 && S.coef_dev_re != nullptr;
  if (S.mode == 1) B = std::max(1, std::min<int>(B, (int)(((S.fill || devcoef) ? 2048.0 : 256.0) * 1024 * 1024 / (N2 * 8.0))));
  if (gen64) B = std::max(1, std::min<int>(B, (int)(2048.0 * 1024 * 1024 / (N2 * 16.0))));
  if (S.epi == 1) B = std::max(1, std::min<int>(B, (int)(256.0 * 1024 * 1024 / (2.0 * Np * Np * 8.0))));
  B = (int)std::min<int64_t>(B, S.n_real);

  // buffers
  if (h->V_cap < (size_t)B) {
    if (h->V) g_slabs.give(h->device, h->V, h->V_bytes);
    h->V = nullptr; h->V_cap = 0;
    const size_t need = (size_t)B * N * Np * sizeof(cpx<R>);
    h->V = g_slabs.take(h->device, need, &h->V_bytes);
    if (!h->V) {
      HIPCHK(hipMalloc(&h->V, need));
      h->V_bytes = need;
    }
    h->V_cap = B;
  }
  // detector partials are kept for FIN_SPAN realisations so that one finalize launch serves many batches
  const int64_t FIN_SPAN = std::max<int64_t>(B, std::min<int64_t>(S.n_real, 32768));
  TRY(grow(&h->partial, &h->partial_cap, (size_t)FIN_SPAN * Np * 4));
  const size_t out_need = (size_t)S.n_real * 2 * (S.coherent ? 2 : 1);
  if (S.epi == 0) TRY(grow(&h->out, &h->out_cap, out_need));
  if (S.epi == 0 && S.logamp) {
    TRY(grow(&h->logamp, &h->logamp_cap, (size_t)S.n_real * 2));
    HIPCHK(hipMemcpyAsync(h->logamp, S.logamp, (size_t)S.n_real * 16, hipMemcpyHostToDevice, h->stream));
  }
  if (kmode == 1 && !devcoef && h->coef_cap < (size_t)B) {
    if (h->cre) HIPCHK(hipFree(h->cre));
    if (h->cim) HIPCHK(hipFree(h->cim));
    h->cre = h->cim = nullptr; h->coef_cap = 0;
    HIPCHK(hipMalloc((void**)&h->cre, (size_t)B * N2 * 8));
    HIPCHK(hipMalloc((void**)&h->cim, (size_t)B * N2 * 8));
    h->coef_cap = B;
  }
  if (S.epi == 1) TRY(grow(&h->phs, &h->phs_cap, (size_t)2 * B * Np * Np));
  const bool sh = h->have_sh;
  if (sh && h->sh_cap < (size_t)B) {
    for (double** p : {&h->sh_coef, &h->sh_mean, &h->sh_dcol, &h->sh_in_re, &h->sh_in_im})
      if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
    TRY(dev_alloc(&h->sh_coef, (size_t)B * 54));
    TRY(dev_alloc(&h->sh_dcol, (size_t)B * Np * 18));
    TRY(dev_alloc(&h->sh_mean, (size_t)B * 2));
    TRY(dev_alloc(&h->sh_in_re, (size_t)B * 27));
    TRY(dev_alloc(&h->sh_in_im, (size_t)B * 27));
    h->sh_cap = B;
  }
  if (sh && devcoef && (!S.sh_dev_re || !S.sh_dev_im)) return fail(FASTMC_EINVAL, "sub-harmonics are set: sh_dev_re / sh_dev_im required");
  if (sh && S.mode == 1 && !S.fill && !devcoef && (!S.sh_re || !S.sh_im)) return fail(FASTMC_EINVAL, "sub-harmonics are set: sh_re / sh_im required");

  RngKey key{(uint32_t)S.seed, (uint32_t)(S.seed >> 32)};
  timing_begin(h);
  int64_t fin_start = 0;
  for (int64_t bs = 0; bs < S.n_real; bs += B) {
    const int nb = (int)std::min<int64_t>(B, S.n_real - bs);
    if (devcoef) {
      // in place
    } else if (S.mode == 1 && S.fill) {
      TRY(S.fill(bs, nb));
    } else if (S.mode == 1) {
      HIPCHK(hipMemcpyAsync(h->cre, S.coeff_re + (size_t)bs * N2, (size_t)nb * N2 * 8, hipMemcpyHostToDevice, h->stream));
      HIPCHK(hipMemcpyAsync(h->cim, S.coeff_im + (size_t)bs * N2, (size_t)nb * N2 * 8, hipMemcpyHostToDevice, h->stream));
    } else if (gen64) {
      Span sg(h, 0);     // counted with the row pass: in float32 mode the generator is part of the row kernel
      const int64_t threads = (int64_t)nb * N * stream_lanes(N);
      hipLaunchKernelGGL(k_gen_coeffs_f64, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, h->stream, key, (uint64_t)(S.real0 + bs), nb, N,
                         h->g64, h->cre, h->cim);
    }
    if (sh) {
      ShCoefArgs SA;
      SA.nb = nb; SA.key = key; SA.g0 = (uint64_t)(S.real0 + bs);
      SA.sh_re = SA.sh_im = nullptr;
      if (devcoef) {
        SA.sh_re = S.sh_dev_re + (size_t)bs * 27; SA.sh_im = S.sh_dev_im + (size_t)bs * 27;
      } else if (S.mode == 1 && S.fill) {
        SA.sh_re = h->sh_in_re; SA.sh_im = h->sh_in_im;          // filled on the device by S.fill
      } else if (S.mode == 1) {
        HIPCHK(hipMemcpyAsync(h->sh_in_re, S.sh_re + (size_t)bs * 27, (size_t)nb * 27 * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->sh_in_im, S.sh_im + (size_t)bs * 27, (size_t)nb * 27 * 8, hipMemcpyHostToDevice, h->stream));
        SA.sh_re = h->sh_in_re; SA.sh_im = h->sh_in_im;
      }
      SA.scale = h->sh_scale; SA.mu = h->sh_mu; SA.coef = h->sh_coef; SA.mean = h->sh_mean; SA.rng_f64 = h->rng_f64;
      hipLaunchKernelGGL(k_subharm_coeffs, dim3((nb + 63) / 64), dim3(64), 0, h->stream, SA);
      if (h->sh_sep)
        hipLaunchKernelGGL(k_subharm_cols, dim3((nb * Np + 255) / 256), dim3(256), 0, h->stream, (const double*)h->sh_coef,
                           (const double*)h->sh_ex, nb, Np, h->sh_dcol);
    }
    RowArgs<R> RA;
    RA.N = N; RA.Np = Np; RA.lo = h->lo; RA.nb = nb;
    RA.om = (const cpx<R>*)h->om; RA.omS = h->omS;
    RA.V = (cpx<R>*)h->V; RA.key = key; RA.g0 = (uint64_t)(S.real0 + bs);
    RA.cre = devcoef ? S.coef_dev_re + (size_t)bs * N2 : h->cre;
    RA.cim = devcoef ? S.coef_dev_im + (size_t)bs * N2 : h->cim;
    RA.g64 = h->g64;
    RA.clk = h->clk;
    h->clk_fresh = false;        // set again by the launch that stamps (launch_rows_wave): another family's row never reports old stamps
    ColArgs<R> CA;
    CA.N = N; CA.Np = Np; CA.lo = h->lo; CA.nb = nb;
    CA.V = (const cpx<R>*)h->V; CA.om = RA.om; CA.omS = h->omS;
    CA.W = h->W;
    CA.sh.enabled = sh ? (h->sh_sep ? 2 : 1) : 0; CA.sh.dcol = h->sh_dcol; CA.sh.coef = h->sh_coef; CA.sh.mean = h->sh_mean; CA.sh.ex = h->sh_ex; CA.sh.ey = h->sh_ey;
    CA.partial = h->partial + (size_t)((bs - fin_start)) * Np * 4; CA.phs = h->phs;
#ifdef FMC_ISA_SUBSET   // tools/isa_stats.py: only the kernels whose instruction mix bench.py prices (same code, a tenth of the compile time)
    RA.amp = (const R*)h->amp_s; RA.ampf = h->ampf_s; RA.tw = (const cpx<R>*)h->tw1; CA.tw = RA.tw;
    RA.cw = (const cpx<R>*)h->cw; CA.cw = RA.cw; RA.tw_global = CA.tw_global = 0;
    if constexpr (sizeof(R) == 8) {
      if (h->N == 2048 && kmode != 1) {        // (as the full build: 2048 goes to the packed sub-rows with a run-time count)
        if (kmode == 2) launch_pks_rows<R, 1, -2, 2>(h, RA); else launch_pks_rows<R, 1, -2, 0>(h, RA);
        launch_pks_cols<R, 1, -2, 0>(h, CA);
        return 0;
      }
    }
    if (h->S == 2) dispatch_wave<R, 16, 2, 2>(h, RA, CA, kmode, S.epi);
    else dispatch_wave<R, 16, 2>(h, RA, CA, kmode, S.epi);
#else
    // packed sub-rows (fmc_core.h: pks_split): rows with the device generator and their own column pass, for the centred windows, on
    // grids of the wave, chirp-z and 50-lane families alike.  Any other window: the float64 generator is staged onto the family's
    // host-coefficient rows (fused_gen64 is false there), the float32 draw goes to the direct family -- no other row kernel knows the
    // N / 16 (N / 8) streams per row of these grids.
    const bool pks = h->path != 0 && kmode != 1 && pks_variant<R>(h) >= 0;
    const bool pks_to_direct = !pks && kmode == 0 && !family_streams_ok(h);
    // A window the grid's own one-row-per-wave rows have no form for (wider than 128 pixels where the radix has no 256-pixel variant:
    // 448, 896, 1152, 1344, 1792 ...) went to the direct family -- O(N Np) per row, 9 k it/s at 1344^2 / Np = 129 where Np = 128 runs 284 k.
    // Host or staged coefficients take the chirp-z rows instead (any N, windows of up to 256 pixels; float64 pipeline).
    bool blu_fallback = false;
    if constexpr (sizeof(R) == 8) {
      if (h->path == 1 && !pks && kmode == 1 && h->blu_P && h->Np > 128) {
        int ns = 0, wpb_unused = 0;
        wave_config<R>(h, &ns, &wpb_unused);
        blu_fallback = ns == 0 && !(h->N == 2048 && wave_lds_bytes<R, 32, 32>(h->omS) <= LDS_MAX);
        if (blu_fallback && h->blu_lo != h->lo) h->blu_lo = -1;
      }
    }
    if ((h->path == 2 || blu_fallback) && !pks && !pks_to_direct) {
      TRY(upload_blu_tables<R>(h));
      RA.amp = (const R*)h->amp; RA.ampf = h->ampf; RA.tw = (const cpx<R>*)h->blu_tw1; RA.om = (const cpx<R>*)h->blu_om;
      RA.cw = nullptr; RA.tw_global = 0;
      RA.blu.twf = (const cpx<R>*)h->blu_twf; RA.blu.pre = (const cpx<R>*)h->blu_pre;
      RA.blu.vhat = (const cpx<R>*)h->blu_vhat; RA.blu.post = (const cpx<R>*)h->blu_post;
      RA.blu.SB = h->blu_SB; RA.blu.B = h->blu_B;
      CA.tw = RA.tw; CA.om = RA.om; CA.cw = nullptr; CA.tw_global = 0; CA.blu = RA.blu;
      if constexpr (sizeof(R) == 8) { TRY(dispatch_blu<R>(h, RA, CA, kmode, S.epi)); }
      else return fail(FASTMC_ESTATE, "chirp-z grids run the float64 kernels (fastmc_create)");
    } else if (h->path == 3 && !pks && !pks_to_direct) {
      TRY(upload_mr_tables<R>(h));
      RA.amp = (const R*)h->amp_s; RA.ampf = h->ampf_s; RA.tw = (const cpx<R>*)h->mr_tw1; RA.om = (const cpx<R>*)h->mr_om;
      RA.cw = (const cpx<R>*)h->mr_cw; RA.tw_global = 0;
      CA.tw = RA.tw; CA.om = RA.om; CA.cw = RA.cw; CA.tw_global = 0;
      if constexpr (sizeof(R) == 8) {
        if (h->mr_P <= 9) { TRY((dispatch_mr_part<R, 0>(h, RA, CA, kmode, S.epi))); }
        else { TRY((dispatch_mr_part<R, 1>(h, RA, CA, kmode, S.epi))); }
      } else return fail(FASTMC_ESTATE, "50-lane grids run the float64 kernels (fastmc_create)");
    } else if (!pks && !pks_to_direct && pk_variant<R>(h) >= 0) {
      RA.amp = (const R*)h->amp_s; RA.ampf = h->ampf_s; RA.tw = (const cpx<R>*)h->pk_tw1; RA.om = (const cpx<R>*)h->pk_om;
      RA.cw = nullptr; RA.tw_global = 0;
      CA.tw = RA.tw; CA.om = RA.om; CA.cw = nullptr; CA.tw_global = 0;
      TRY(dispatch_pk<R>(h, RA, CA, kmode, S.epi));
    } else {
    bool wave_ok = h->path == 1 && !(pk_grid(h->N) && kmode == 0) && !(pks_grid(h->N) && kmode == 0 && !pks);    // packed grids beyond the packed windows: see pk_variant, pks_variant
    const int wmode = kmode;
    // the one-row-per-wave kernels of the N / 16-stream grids exist for host coefficients only (launch_wave_pair): a device draw must
    // never reach them (it would transform whatever `cre` / `cim` hold)
    if (!pks && wave_ok && (pk_grid(h->N) || pks_grid(h->N)) && wmode != 1)
      return fail(FASTMC_ESTATE, "internal: device draws routed to a host-coefficient row kernel");
    bool general_2048 = false;   // N = 2048, window > 256 pixels, host coefficients: single-pass P = 32 kernels
    if (wave_ok) {
      int ns, wpb_unused;
      wave_config<R>(h, &ns, &wpb_unused);
      if (ns == 0) {
        // window tables exceed the LDS / no instantiation: direct family (still on the GPU)
        if (h->N == 2048 && kmode == 1 && wave_lds_bytes<R, 32, 32>(h->omS) <= LDS_MAX) general_2048 = true;
        else wave_ok = false;
      }
    }
    RA.amp = (const R*)(wave_ok ? h->amp_s : h->amp);
    RA.ampf = wave_ok ? h->ampf_s : h->ampf;
    RA.tw = (const cpx<R>*)(wave_ok ? (general_2048 ? h->tw1g : h->tw1) : h->tw);
    CA.tw = RA.tw;
    RA.cw = (const cpx<R>*)h->cw;
    RA.tw_global = 0;
    CA.tw_global = 0;
    CA.cw = RA.cw;
    if (pks) {
      // rows with the device generator and the column pass, V permuted along ky between them (fmc_kernels.h: k_rows_pks / k_cols_pks)
      TRY(dispatch_pks<R>(h, RA, CA, kmode, S.epi));
    } else if (general_2048 || wave_ok) {
      if (!general_2048 && wave_rt_split(h->N)) {
        if constexpr (sizeof(R) == 8) { TRY(dispatch_ws<R>(h, RA, CA, kmode, S.epi)); }
        else return fail(FASTMC_ESTATE, "run-time-split grids run the float64 kernels (fastmc_create)");
      } else if (general_2048 || (h->S == 1 && h->P >= 14 && h->P != 16)) {
        TRY((dispatch_wave_part<R, 2>(h, RA, CA, wmode, S.epi, general_2048)));
      } else if (h->P == 16) {
        TRY((dispatch_wave_part<R, 1>(h, RA, CA, kmode, S.epi, false)));
      } else {
        TRY((dispatch_wave_part<R, 0>(h, RA, CA, wmode, S.epi, false)));
      }
    } else {
      TRY(dispatch_direct<R>(h, RA, CA, kmode, S.epi));
    }
    }
#endif
    if (S.epi == 0) {
      const int64_t done = bs + nb;                 // realisations with partials ready: [fin_start, done)
      if (done == S.n_real || done - fin_start + B > FIN_SPAN) {
        Span sp(h, 2);
        FinArgs FA;
        FA.nb = (int)(done - fin_start); FA.Np = Np; FA.coherent = S.coherent; FA.n_real = S.n_real; FA.j0 = fin_start;
        FA.partial = h->partial; FA.logamp = S.logamp_dev ? S.logamp_dev : (S.logamp ? h->logamp : nullptr);
        FA.logamp_sigma = std::sqrt(S.logamp_var); FA.rng_f64 = h->rng_f64; FA.key = key; FA.g0 = (uint64_t)(S.real0 + fin_start);
        FA.dx2 = h->dx * h->dx; FA.norm = h->wsum * (h->dx * h->dx); FA.out = S.out_dev ? S.out_dev : h->out;
        guard_outputs(h);
        hipLaunchKernelGGL(k_finalize, dim3((FA.nb + 3) / 4), dim3(256), 0, h->stream, FA);
        fin_start = done;
      }
    } else {
      // screens of this batch -> host: Re planes of [bs, bs+nb), Im planes offset by n_real
      const size_t plane = (size_t)Np * Np;
      HIPCHK(hipMemcpyAsync(S.phs + (size_t)bs * plane, h->phs, (size_t)nb * plane * 8, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipMemcpyAsync(S.phs + (size_t)(S.n_real + bs) * plane, h->phs + (size_t)nb * plane, (size_t)nb * plane * 8,
                            hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
    }
    HIPCHK(hipGetLastError());
  }
  if (S.epi == 0) { h->last_n_iter = 2 * S.n_real; h->last_coherent = S.coherent; h->last_out_doubles = out_need; }
  if (S.async) {          // results stay on the device; fastmc_wait or the exchange synchronises
    h->pending = true;
    return 0;
  }
  if (S.epi == 0)
    HIPCHK(hipMemcpyAsync(S.out, h->out, out_need * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  timing_end(h);
  return 0;
}

static int run_locked(fastmc_ctx* h, const RunSpec& S);
static int histogram_device(fastmc_ctx* h, double lo, double hi, int nbins);
static int stall_until_abort(const std::vector<int>& devices);
static std::atomic<int> g_abort_gen[64];      // bumped by fastmc_comm_abort: wakes a stalled exchange of that device
static int cslot(const fastmc_ctx* h);        // slot of a handle in the communicator tables (its device's, see the definition)
static int run_checked(fastmc_ctx* h, const RunSpec& S) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  FMC_LOCK(h);
  return run_locked(h, S);
}
static int run_locked(fastmc_ctx* h, const RunSpec& S) {      // the caller holds the handle's lock
  if (h->pending) {       // an asynchronous run nobody waited for: its events are read before they are reused
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    finish_pending(h);
  }
  if (!h->have_spec || !h->have_pupil) return fail(FASTMC_ESTATE, "set_spectrum and set_pupil must be called first");
  if (S.n_real <= 0) return fail(FASTMC_EINVAL, "n_real must be positive");
  if (S.real0 < 0) return fail(FASTMC_EINVAL, "real0 must be non-negative");
  HIPCHK(hipSetDevice(h->device));
  TRY(ensure_gen64_table(h));         // (a float64 handle draws at float64 precision from fastmc_create on)
#ifdef FMC_ONLY_F64   // experiment builds (make variant ... DEFS="-DFMC_ONLY_F64 ..."): half the compile time
  if (h->precision != FASTMC_F64) return fail(FASTMC_ESTATE, "this build has no float32 kernels (FMC_ONLY_F64)");
  return run_impl<double>(h, S);
#else
  return h->precision == FASTMC_F64 ? run_impl<double>(h, S) : run_impl<float>(h, S);
#endif
}

#if FMC_TU == 0
extern "C" int fastmc_run(fastmc_t* h, uint64_t seed, int64_t real0, int64_t n_real, const double* logamp,
                          double logamp_var, int coherent, double* out) {
  if (!out) return fail(FASTMC_EINVAL, "out is NULL");
  if (!(logamp_var >= 0.0)) return fail(FASTMC_EINVAL, "logamp_var must be >= 0");
  RunSpec S{0, 0, seed, real0, n_real, nullptr, nullptr, nullptr, nullptr, logamp, logamp_var, coherent, out, nullptr};
  return run_checked(h, S);
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_run_async(fastmc_t* h, uint64_t seed, int64_t real0, int64_t n_real, double logamp_var, int coherent) {
  if (!(logamp_var >= 0.0)) return fail(FASTMC_EINVAL, "logamp_var must be >= 0");
  RunSpec S{0, 0, seed, real0, n_real, nullptr, nullptr, nullptr, nullptr, nullptr, logamp_var, coherent, nullptr, nullptr};
  S.async = true;
  return run_checked(h, S);
}

extern "C" int fastmc_wait(fastmc_t* h, double* out) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  FMC_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  if (out) {
    if (h->last_n_iter <= 0 || !h->last_out_doubles) return fail(FASTMC_ESTATE, "no run results on the device");
    HIPCHK(hipMemcpyAsync(out, h->out, h->last_out_doubles * 8, hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  finish_pending(h);
  return 0;
}
#endif

#if FMC_TU == 0
// ---- two steps in flight (QueueSlot)
static int slot_copies_begin(fastmc_ctx* h) {      // the copies that follow wait for everything enqueued on the compute stream so far
  if (!h->cstream) {
    HIPCHK(hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->copy_fence, hipEventDisableTiming));
  }
  HIPCHK(hipEventRecord(h->copy_fence, h->stream));
  HIPCHK(hipStreamWaitEvent(h->cstream, h->copy_fence, 0));
  return 0;
}
static int slot_land(fastmc_ctx* h, QueueSlot& q, const double* dev, size_t n) {     // D2H into the slot's pinned buffer, on the copy stream
  TRY(slot_copies_begin(h));
  if (q.pinned_cap < n) {
    if (q.pinned) HIPCHK(hipHostFree(q.pinned));
    q.pinned = nullptr; q.pinned_cap = 0;
    HIPCHK(hipHostMalloc((void**)&q.pinned, n * 8, hipHostMallocDefault));
    q.pinned_cap = n;
  }
  HIPCHK(hipMemcpyAsync(q.pinned, dev, n * 8, hipMemcpyDeviceToHost, h->cstream));
  q.landed = n;
  return 0;
}
static int slot_land_hist(fastmc_ctx* h, QueueSlot& q, int nbins) {
  TRY(slot_copies_begin(h));
  const size_t n = (size_t)nbins + 2;
  if (q.hist_cap < n) {
    if (q.pinned_hist) HIPCHK(hipHostFree(q.pinned_hist));
    q.pinned_hist = nullptr; q.hist_cap = 0;
    HIPCHK(hipHostMalloc((void**)&q.pinned_hist, n * 8, hipHostMallocDefault));
    q.hist_cap = n;
  }
  HIPCHK(hipMemcpyAsync(q.pinned_hist, h->hist, n * 8, hipMemcpyDeviceToHost, h->cstream));
  q.hist_landed = n;
  return 0;
}
static int slot_mark_done(fastmc_ctx* h, QueueSlot& q) {
  if (!q.done) HIPCHK(hipEventCreateWithFlags(&q.done, hipEventDisableTiming));
  TRY(slot_copies_begin(h));                             // the completion follows everything on the compute stream so far, copies or not
  HIPCHK(hipEventRecord(q.done, h->cstream));
  h->copy_guard = q.done;
  q.busy = true;
  return 0;
}

extern "C" int fastmc_run_queued(fastmc_t* h, uint64_t seed, int64_t real0, int64_t n_real, double logamp_var, int coherent, int slot,
                                 int fetch) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  if (slot < 0 || slot > 1) return fail(FASTMC_EINVAL, "slot must be 0 or 1");
  if (!(logamp_var >= 0.0)) return fail(FASTMC_EINVAL, "logamp_var must be >= 0");
  FMC_LOCK(h);
  QueueSlot& q = h->q[slot];
  if (q.busy) return fail(FASTMC_ESTATE, "this slot is still in flight (fastmc_queue_wait first)");
  {
    SlotScope sc(h, q);
    RunSpec S{0, 0, seed, real0, n_real, nullptr, nullptr, nullptr, nullptr, nullptr, logamp_var, coherent, nullptr, nullptr};
    S.async = true;
    TRY(run_locked(h, S));
  }
  q.landed = q.hist_landed = 0;
  if (fetch) TRY(slot_land(h, q, h->out, h->last_out_doubles));
  return slot_mark_done(h, q);
}

// the dB histogram of a queued step's own results (no exchange), landed on the slot
extern "C" int fastmc_histogram_queued(fastmc_t* h, double lo_db, double hi_db, int nbins, int slot) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  if (slot < 0 || slot > 1) return fail(FASTMC_EINVAL, "slot must be 0 or 1");
  FMC_LOCK(h);
  QueueSlot& q = h->q[slot];
  if (!q.busy) return fail(FASTMC_ESTATE, "fastmc_run_queued on this slot first");
  HIPCHK(hipSetDevice(h->device));
  TRY(histogram_device(h, lo_db, hi_db, nbins));
  TRY(slot_land_hist(h, q, nbins));
  return slot_mark_done(h, q);
}

extern "C" int fastmc_queue_wait(fastmc_t* h, int slot, double* out, int64_t out_cap, int64_t* hist, int hist_cap) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  if (slot < 0 || slot > 1) return fail(FASTMC_EINVAL, "slot must be 0 or 1");
  FMC_LOCK(h);
  QueueSlot& q = h->q[slot];
  if (!q.busy) return fail(FASTMC_ESTATE, "nothing is queued on this slot");
  HIPCHK(hipSetDevice(h->device));
  if (q.stalled) {               // the injected fault: an exchange that never completes, until the caller aborts
    while (g_abort_gen[cslot(h)].load() == q.stall_gen) usleep(500);      // an abort since the exchange was queued ends the stall
    const int rc = fail(FASTMC_ECOMM, "exchange aborted (FASTMC_TEST_STALL_GATHER)");
    hipEventSynchronize(q.done);
    { SlotScope sc(h, q); finish_pending(h); }
    q.stalled = q.busy = false;
    q.landed = q.hist_landed = 0;
    return rc;
  }
  HIPCHK(hipEventSynchronize(q.done));
  {
    SlotScope sc(h, q);
    finish_pending(h);          // this step's events -> fastmc_last_timing / fastmc_last_exchange_ms
  }
  q.busy = false;
  if (out) {
    if ((size_t)out_cap < q.landed) return fail(FASTMC_EINVAL, "out is smaller than what the step landed");
    if (!q.landed) return fail(FASTMC_ESTATE, "the step on this slot landed no results on this handle (fetch = 0 and no gather)");
    memcpy(out, q.pinned, q.landed * 8);
  }
  if (hist) {
    if ((size_t)hist_cap < q.hist_landed || !q.hist_landed) return fail(FASTMC_EINVAL, "no histogram landed on this slot, or hist is too small");
    memcpy(hist, q.pinned_hist, q.hist_landed * 8);
  }
  return (int)q.landed;
}
#endif

#if FMC_TU == 0
// ------------------------------------------------------------------ numpy's normal stream on the device (fmc_npstream.h)
struct NpsSegBuf {           // what classify / scan leave behind for emit, one per segment kept alive
  NpsEvent* events = nullptr;
  uint32_t* evcount = nullptr;
  uint32_t* maps = nullptr;
  uint8_t* tile_e = nullptr;
  uint64_t* tile_base = nullptr;
  u128* tile_state = nullptr;
  int64_t cap_tiles = 0;
};
struct NpsOneBuf {           // the one-pass generator's buffers for one segment: its normals and the look-back words
  double* out = nullptr;
  size_t cap_out = 0;
  uint32_t* xexit = nullptr;
  unsigned long long* agg = nullptr;
  u128* tile_state = nullptr;
  uint32_t* ticket = nullptr;
  int64_t cap_tiles = 0;
};
struct NpsWork {
  NpsOneBuf one[4];          // like seg[]
  NpsSegBuf seg[4];          // [2 * set + k]: two sets (chunks alternate), k = 0 the coefficients of a chunk, 1 its sub-harmonic draws
  hipStream_t gstream = nullptr;       // the generator chain (tile states, classify, scan) of chunk c + 1 runs here beside the
  hipEvent_t ev_gen[2] = {nullptr, nullptr};   //   emit + Monte-Carlo kernels of chunk c on the handle's stream
  hipEvent_t ev_used[2] = {nullptr, nullptr};
  u128* states = nullptr;    // [cap_states] chain of generator states: states[k] = before segment k of the call
  uint64_t* consumed = nullptr;    // [cap_states]
  uint32_t* overflow = nullptr;    // [cap_states] per segment
  size_t cap_states = 0;
  double* la = nullptr;      // log-amplitude normals of the run (device)
  size_t la_cap = 0;
  double* res = nullptr;     // results of all chunks of one fastmc_run_npstream call (device): copied back once
  size_t res_cap = 0;
  // the device's ziggurat tables and jump table, looked up ONCE per call under g_nps_mu (nps_prepare): the launches below never
  // touch the map, which fastmc_npstream_set_tables of another thread / device may be inserting into
  NpsTables* tab = nullptr;
  NpsJump* jump = nullptr;
};
static void nps_free(NpsWork* w) {
  if (w->gstream) hipStreamSynchronize(w->gstream);
  for (NpsSegBuf& b : w->seg)
    for (void* p : {(void*)b.events, (void*)b.evcount, (void*)b.maps, (void*)b.tile_e, (void*)b.tile_base, (void*)b.tile_state}) if (p) hipFree(p);
  for (NpsOneBuf& b : w->one)
    for (void* p : {(void*)b.out, (void*)b.xexit, (void*)b.agg, (void*)b.tile_state, (void*)b.ticket}) if (p) hipFree(p);
  for (void* p : {(void*)w->states, (void*)w->consumed, (void*)w->overflow, (void*)w->la, (void*)w->res}) if (p) hipFree(p);
  for (int i = 0; i < 2; ++i) { if (w->ev_gen[i]) hipEventDestroy(w->ev_gen[i]); if (w->ev_used[i]) hipEventDestroy(w->ev_used[i]); }
  if (w->gstream) hipStreamDestroy(w->gstream);
  delete w;
}
static std::mutex g_nps_mu;
static std::map<int, std::pair<NpsTables*, NpsJump*>> g_nps_dev;     // per device: ziggurat tables (from numpy), jump table

static void nps_jump_table(NpsJump* J) {
  const u128 mult = ((((u128)0x2360ED051FC65DA4ull) << 64) | 0x4385DF649FCCF645ull);
  u128 a = mult, g = 1;        // a^(2^k) and 1 + a + ... + a^(2^k - 1)
  for (int k = 0; k < 64; ++k) {
    J->a[k] = a;
    J->c[k] = g;
    g = g * (a + 1);
    a = a * a;
  }
}

extern "C" int fastmc_npstream_set_tables(int device_id, const double* wi, const uint64_t* ki, const double* fi) {
  if (!wi || !ki || !fi) return fail(FASTMC_EINVAL, "null table");
  HIPCHK(hipSetDevice(device_id));
  std::lock_guard<std::mutex> lk(g_nps_mu);
  auto& e = g_nps_dev[device_id];
  if (!e.first) {
    HIPCHK(hipMalloc((void**)&e.first, sizeof(NpsTables)));
    HIPCHK(hipMalloc((void**)&e.second, sizeof(NpsJump)));
    NpsJump* J = new NpsJump;
    nps_jump_table(J);
    hipError_t r = hipMemcpy(e.second, J, sizeof(NpsJump), hipMemcpyHostToDevice);
    delete J;
    HIPCHK(r);
  }
  NpsTables T;
  memcpy(T.wi, wi, sizeof(T.wi)); memcpy(T.ki, ki, sizeof(T.ki)); memcpy(T.fi, fi, sizeof(T.fi));
  HIPCHK(hipMemcpy(e.first, &T, sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

static int64_t nps_tiles_for(uint64_t n, int tile_words = NPS_T) {       // upper bound of the words n normals consume (mean 1.0222 per normal), in tiles
  const uint64_t words = n + n / 32 + 2 * (uint64_t)NPS_T;
  return (int64_t)((words + tile_words - 1) / tile_words);
}
#ifndef FMC_NPS1_NSUB
#define FMC_NPS1_NSUB 2
#endif
constexpr int NPS1_NSUB = FMC_NPS1_NSUB;          // the one-pass generator's tile: NSUB * 2048 words
static int nps_seg_reserve(NpsSegBuf& b, int64_t tiles) {
  if (b.cap_tiles >= tiles) return 0;
  for (void* p : {(void*)b.events, (void*)b.evcount, (void*)b.maps, (void*)b.tile_e, (void*)b.tile_base, (void*)b.tile_state}) if (p) hipFree(p);
  b = NpsSegBuf();
  HIPCHK(hipMalloc((void**)&b.events, (size_t)tiles * NPS_EVCAP * sizeof(NpsEvent)));
  HIPCHK(hipMalloc((void**)&b.evcount, (size_t)tiles * 4));
  HIPCHK(hipMalloc((void**)&b.maps, (size_t)tiles * NPS_K * 4));
  HIPCHK(hipMalloc((void**)&b.tile_e, (size_t)tiles));
  HIPCHK(hipMalloc((void**)&b.tile_base, (size_t)tiles * 8));
  HIPCHK(hipMalloc((void**)&b.tile_state, (size_t)tiles * sizeof(u128)));
  b.cap_tiles = tiles;
  return 0;
}
static int nps_reserve_states(NpsWork* w, size_t n) {
  if (w->cap_states >= n) return 0;
  for (void* p : {(void*)w->states, (void*)w->consumed, (void*)w->overflow}) if (p) hipFree(p);
  w->states = nullptr; w->consumed = nullptr; w->overflow = nullptr; w->cap_states = 0;
  HIPCHK(hipMalloc((void**)&w->states, n * sizeof(u128)));
  HIPCHK(hipMalloc((void**)&w->consumed, n * 8));
  HIPCHK(hipMalloc((void**)&w->overflow, n * 4));
  w->cap_states = n;
  return 0;
}
// classify + scan of segment k of the call: n normals from states[k]; leaves states[k + 1]
static int nps_segment(fastmc_ctx* h, NpsSegArgs& A, NpsSegBuf& b, size_t k, u128 inc, uint64_t n, hipStream_t stream = nullptr) {
  if (!stream) stream = h->stream;
  NpsWork* w = h->nps;
  const int64_t tiles = nps_tiles_for(n);
  TRY(nps_seg_reserve(b, tiles));
  A.state = w->states + k; A.inc = inc; A.n = n; A.ntiles = tiles; A.tab = w->tab; A.jump = w->jump;
  A.events = b.events; A.evcount = b.evcount; A.maps = b.maps; A.tile_e = b.tile_e; A.tile_base = b.tile_base; A.tile_state = b.tile_state;
  A.state_out = w->states + k + 1; A.consumed = w->consumed + k; A.overflow = w->overflow + k;
  static const bool general_scan = [] { const char* e = getenv("FASTMC_NPS_GENERAL_SCAN"); return e && *e == '1'; }();      // (tests)
  A.flags = general_scan ? NPS_SCAN_GENERAL : 0u;
  hipLaunchKernelGGL(k_nps_tilestates, dim3((unsigned)((tiles + NPS_THREADS - 1) / NPS_THREADS)), dim3(NPS_THREADS), 0, stream, A);
  hipLaunchKernelGGL(k_nps_classify, dim3((unsigned)tiles), dim3(NPS_THREADS), 0, stream, A);
  hipLaunchKernelGGL(k_nps_scan, dim3(1), dim3(NPS_SCAN_THREADS), 0, stream, A);
  return 0;
}
// The one-pass form (fmc_npstream.h: k_nps_onepass) when the segment's normals fit a device buffer of their own: every word is
// generated once.  FASTMC_NPS_ONEPASS_MAX_GB (default 16) bounds that buffer; FASTMC_NPS_THREEPASS=1 forces the three-pass form (tests).
static bool nps_use_onepass(uint64_t n) {
  static const bool three = [] { const char* e = getenv("FASTMC_NPS_THREEPASS"); return e && *e == '1'; }();
  static const double max_gb = [] { const char* e = getenv("FASTMC_NPS_ONEPASS_MAX_GB"); return e ? atof(e) : 16.0; }();
  return !three && (double)n * 8.0 <= max_gb * 1073741824.0;
}
// segment k of the call, n normals from states[k], written to b.out[0 ... n); leaves states[k + 1]
static int nps_onepass(fastmc_ctx* h, NpsOneBuf& b, size_t k, u128 inc, uint64_t n, hipStream_t stream = nullptr) {
  if (!stream) stream = h->stream;
  NpsWork* w = h->nps;
  const std::pair<NpsTables*, NpsJump*> dev(w->tab, w->jump);
  const int64_t tiles = nps_tiles_for(n, NPS1_NSUB * NPS_SUB);
  if (b.cap_tiles < tiles) {
    for (void* p : {(void*)b.xexit, (void*)b.agg, (void*)b.tile_state}) if (p) hipFree(p);
    b.xexit = nullptr; b.agg = nullptr; b.tile_state = nullptr; b.cap_tiles = 0;
    HIPCHK(hipMalloc((void**)&b.xexit, (size_t)tiles * 4));
    HIPCHK(hipMalloc((void**)&b.agg, (size_t)tiles * 8));
    HIPCHK(hipMalloc((void**)&b.tile_state, (size_t)tiles * sizeof(u128)));
    if (!b.ticket) HIPCHK(hipMalloc((void**)&b.ticket, 4));
    b.cap_tiles = tiles;
  }
  if (b.cap_out < n) {
    if (b.out) HIPCHK(hipFree(b.out));
    b.out = nullptr; b.cap_out = 0;
    HIPCHK(hipMalloc((void**)&b.out, (size_t)n * 8));
    b.cap_out = n;
  }
  NpsSegArgs A{};
  A.state = w->states + k; A.inc = inc; A.n = n; A.ntiles = tiles; A.tab = dev.first; A.jump = dev.second;
  A.tile_state = b.tile_state;
  A.state_out = w->states + k + 1; A.consumed = w->consumed + k; A.overflow = w->overflow + k;
  NpsOneArgs O;
  O.out = b.out; O.xexit = b.xexit; O.agg = b.agg; O.ticket = b.ticket;
  hipLaunchKernelGGL(k_nps_tilestates1<NPS1_NSUB>, dim3((unsigned)((tiles + NPS_THREADS - 1) / NPS_THREADS)), dim3(NPS_THREADS), 0, stream, A, O);
  hipLaunchKernelGGL(k_nps_onepass<NPS1_NSUB>, dim3((unsigned)tiles), dim3(NPS_THREADS), 0, stream, A, O);
  return 0;
}
// normals [lo, hi) of a classified segment -> out[0 ... hi - lo); optionally a second range of the same segment in the same launch
static NpsEmitRange nps_range(const NpsSegArgs& A, uint64_t lo, uint64_t hi, double* out) {
  // the tiles that can hold them: normal index i starts no earlier than word i and no later than word 1.0625 i + 3 T
  NpsEmitRange R;
  R.tile0 = (int64_t)(lo / NPS_T);
  int64_t t1 = (int64_t)((hi + hi / 16) / NPS_T) + 3;
  if (t1 > A.ntiles) t1 = A.ntiles;
  R.ntiles = t1 > R.tile0 ? t1 - R.tile0 : 0;
  R.lo = lo; R.hi = hi; R.out = out;
  return R;
}
static void nps_emit2(fastmc_ctx* h, const NpsSegArgs& A, const NpsEmitRange& R0, const NpsEmitRange& R1) {
  if (R0.ntiles + R1.ntiles > 0)
    hipLaunchKernelGGL(k_nps_emit, dim3((unsigned)(R0.ntiles + R1.ntiles)), dim3(NPS_THREADS), 0, h->stream, A, R0, R1);
}
static void nps_emit(fastmc_ctx* h, const NpsSegArgs& A, uint64_t lo, uint64_t hi, double* out) {
  NpsEmitRange none;
  none.tile0 = 0; none.ntiles = 0; none.lo = none.hi = 0; none.out = nullptr;
  nps_emit2(h, A, nps_range(A, lo, hi, out), none);
}
static int nps_prepare(fastmc_ctx* h, const uint64_t state_inc[4], size_t n_states, u128* inc) {
  NpsTables* tab = nullptr;
  NpsJump* jump = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_nps_mu);
    auto it = g_nps_dev.find(h->device);
    if (it == g_nps_dev.end() || !it->second.first)
      return fail(FASTMC_ESTATE, "fastmc_npstream_set_tables has not been called for this device");
    tab = it->second.first; jump = it->second.second;
  }
  if (!h->nps) h->nps = new NpsWork;
  h->nps->tab = tab; h->nps->jump = jump;
  TRY(nps_reserve_states(h->nps, n_states));
  const u128 st = ((u128)state_inc[1] << 64) | state_inc[0];
  *inc = ((u128)state_inc[3] << 64) | state_inc[2];
  HIPCHK(hipMemsetAsync(h->nps->overflow, 0, n_states * 4, h->stream));
  HIPCHK(hipMemcpyAsync(h->nps->states, &st, sizeof(u128), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));          // (st is a stack variable)
  return 0;
}

// One array of n normals, exactly `numpy.random.Generator(PCG64 at this state).normal(size=n)`: out (host, n doubles, may be
// NULL), the generator state after it, the words it consumed, the overflow flags (0: the device's answer stands).
extern "C" int fastmc_npstream_normals(fastmc_t* h, const uint64_t state_inc[4], int64_t n, double* out, uint64_t state_after[2],
                                       uint64_t* consumed, uint32_t* overflow) {
  if (!h || !state_inc || n < 1) return fail(FASTMC_EINVAL, "bad argument");
  FMC_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  u128 inc;
  TRY(nps_prepare(h, state_inc, 2, &inc));
  ScratchBuf d;
  if (nps_use_onepass((uint64_t)n)) {
    NpsOneBuf& b = h->nps->one[0];
    TRY(nps_onepass(h, b, 0, inc, (uint64_t)n));
    if (out) HIPCHK(hipMemcpyAsync(out, b.out, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
  } else {
    NpsSegArgs A;
    TRY(nps_segment(h, A, h->nps->seg[0], 0, inc, (uint64_t)n));
    if (out) {
      HIPCHK(hipMalloc((void**)&d.p, (size_t)n * 8));
      nps_emit(h, A, 0, (uint64_t)n, d.p);
      HIPCHK(hipMemcpyAsync(out, d.p, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    }
  }
  u128 st;
  uint64_t cons = 0;
  uint32_t ovf = 0;
  HIPCHK(hipMemcpyAsync(&st, h->nps->states + 1, sizeof(u128), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(&cons, h->nps->consumed, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(&ovf, h->nps->overflow, 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
#ifdef NPS1_EXP_TIMES
  if (const char* path = getenv("FASTMC_NPS_DUMP")) {
    const int64_t tiles = nps_tiles_for((uint64_t)n, NPS1_NSUB * NPS_SUB);
    std::vector<u128> rec((size_t)tiles);
    hipMemcpy(rec.data(), h->nps->one[0].tile_state, rec.size() * sizeof(u128), hipMemcpyDeviceToHost);
    if (FILE* f = fopen(path, "wb")) { fwrite(rec.data(), sizeof(u128), rec.size(), f); fclose(f); }
  }
#endif
  if (state_after) { state_after[0] = (uint64_t)st; state_after[1] = (uint64_t)(st >> 64); }
  if (consumed) *consumed = cons;
  if (overflow) *overflow = ovf;
  return 0;
}
#endif

#if FMC_TU == 0
__global__ void k_nps_scale(double* x, int64_t n, double s) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= s;
}

// The log-amplitude draws of a run (fast/fast.py:123, 639-645; funcs.py:358-365): normal(n_iter) + 1j normal(n_iter), real part
// kept and scaled -- two consecutive arrays of one stream = one array of 2 n_iter.
extern "C" int fastmc_npstream_logamp(fastmc_t* h, const uint64_t state_inc[4], int64_t n_iter, double logamp_var, double* logamp,
                                      uint64_t state_after[2], uint32_t* overflow) {
  if (!h || !state_inc || n_iter < 1 || !(logamp_var >= 0.0)) return fail(FASTMC_EINVAL, "bad argument");
  FMC_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  u128 inc;
  TRY(nps_prepare(h, state_inc, 2, &inc));
  NpsWork* w = h->nps;
  TRY(grow(&w->la, &w->la_cap, (size_t)n_iter));
  if (nps_use_onepass((uint64_t)(2 * n_iter))) {
    TRY(nps_onepass(h, w->one[0], 0, inc, (uint64_t)(2 * n_iter)));
    HIPCHK(hipMemcpyAsync(w->la, w->one[0].out, (size_t)n_iter * 8, hipMemcpyDeviceToDevice, h->stream));
  } else {
    NpsSegArgs A;
    TRY(nps_segment(h, A, w->seg[0], 0, inc, (uint64_t)(2 * n_iter)));
    nps_emit(h, A, 0, (uint64_t)n_iter, w->la);
  }
  hipLaunchKernelGGL(k_nps_scale, dim3((unsigned)((n_iter + 255) / 256)), dim3(256), 0, h->stream, w->la, n_iter, std::sqrt(logamp_var));
  u128 st;
  uint32_t ovf = 0;
  if (logamp) HIPCHK(hipMemcpyAsync(logamp, w->la, (size_t)n_iter * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(&st, w->states + 1, sizeof(u128), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(&ovf, w->overflow, 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  if (state_after) { state_after[0] = (uint64_t)st; state_after[1] = (uint64_t)(st >> 64); }
  if (overflow) *overflow = ovf;
  return 0;
}

// Chunks of the Monte-Carlo loop with the coefficients drawn from numpy's stream on the device.  Everything is enqueued
// without a host round trip: the state chain, the draws, the kernels of every chunk; ONE synchronisation at the end.
extern "C" int fastmc_run_npstream(fastmc_t* h, const uint64_t state_inc[4], int64_t n_chunks, int64_t chunk_real, int64_t logamp_offset,
                                   int coherent, double* out, uint64_t state_after[2], int64_t* bad_chunk) {
  if (!h || !state_inc || !out || n_chunks < 1 || chunk_real < 1 || logamp_offset < 0) return fail(FASTMC_EINVAL, "bad argument");
  FMC_LOCK(h);
  if (!h->have_spec || !h->have_pupil) return fail(FASTMC_ESTATE, "set_spectrum and set_pupil must be called first");
  if (h->precision != FASTMC_F64) return fail(FASTMC_ESTATE, "the numpy-stream generator feeds the float64 pipeline");
  HIPCHK(hipSetDevice(h->device));
  const bool sh = h->have_sh;
  // per chunk ONE array of the stream holds the real parts then the imaginary parts (two consecutive normal() calls are one
  // longer one), a second one the sub-harmonic draws: one classify + scan each
  const int segs = sh ? 2 : 1;
  u128 inc;
  TRY(nps_prepare(h, state_inc, (size_t)n_chunks * segs + 1, &inc));
  NpsWork* w = h->nps;
  if (!w->la || w->la_cap < (size_t)(logamp_offset + n_chunks * 2 * chunk_real))
    return fail(FASTMC_ESTATE, "fastmc_npstream_logamp first (the run's log-amplitudes are drawn before its chunks)");
  const size_t N2 = (size_t)h->N * h->N;
  const size_t per_chunk = (size_t)2 * chunk_real * (coherent ? 2 : 1);
  // results of all chunks land in one pinned buffer; copied out after the one synchronisation
  double* pinned = nullptr;
  HIPCHK(hipHostMalloc((void**)&pinned, per_chunk * n_chunks * 8, hipHostMallocDefault));
  struct Free { double* p; ~Free() { if (p) hipHostFree(p); } } guard{pinned};
  if (w->res_cap < per_chunk * (size_t)n_chunks) {
    if (w->res) HIPCHK(hipFree(w->res));
    w->res = nullptr; w->res_cap = 0;
    HIPCHK(hipMalloc((void**)&w->res, per_chunk * n_chunks * 8));
    w->res_cap = per_chunk * (size_t)n_chunks;
  }
  // Two streams: the generator chain of chunk c + 1 (tile states, classify, scan: sequential in the stream's state, latency-bound)
  // runs beside the emit + Monte-Carlo kernels of chunk c (HBM-bound); the segment buffers alternate between two sets.
  // (FASTMC_NPS_TWO_STREAMS=1: the generator of chunk c + 1 on a stream of its own beside the Monte-Carlo kernels of chunk c.
  //  Nothing overlaps -- round 6 ran the two streams FREE of each other (no events: timing only) and a chunk still took the SUM of
  //  the two, also with the rows as light four-wave workgroups that would fit a CU beside three of the generator's: the generator's
  //  26 000 small workgroups refill every slot the moment it frees, so a row workgroup finds room only when the generator's queue
  //  is empty.  The cross-stream events only add launch gaps: one stream.  profiles/r06_ab_same_seed_mode.txt)
  static const bool two = [] { const char* e = getenv("FASTMC_NPS_TWO_STREAMS"); return e && *e == '1'; }();
  // The chunks' MODE 1 rows (50 realisations each: HBM-bound coefficient reads) as FOUR-wave workgroups (WCfg D = 9): 197-202 us per
  // chunk against 247 for the sixteen-wave form, same-seed mode +3.3 % (profiles/r06_ab_same_seed_mode.txt); FASTMC_NPS_LIGHT_ROWS=0: A/B
  static const bool light = [] { const char* e = getenv("FASTMC_NPS_LIGHT_ROWS"); return !(e && *e == '0'); }();
  h->light_rows = light;
  struct Unlight { fastmc_ctx* h; ~Unlight() { h->light_rows = false; } } unlight{h};
  if (two && !w->gstream) {
    HIPCHK(hipStreamCreateWithFlags(&w->gstream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      HIPCHK(hipEventCreateWithFlags(&w->ev_gen[i], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&w->ev_used[i], hipEventDisableTiming));
    }
  }
  hipStream_t gs = two ? w->gstream : h->stream;
  if (two) {
    hipEvent_t ev0;      // the generator stream starts after what nps_prepare put on the handle's stream
    HIPCHK(hipEventCreateWithFlags(&ev0, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev0, h->stream));
    HIPCHK(hipStreamWaitEvent(w->gstream, ev0, 0));
    hipEventDestroy(ev0);
  }
  const uint64_t nc = (uint64_t)chunk_real * N2, ns = (uint64_t)chunk_real * 27;
  std::vector<NpsSegArgs> AA((size_t)n_chunks * 2);
  const bool onepass = nps_use_onepass(2 * nc);
  auto gen_chunk = [&](int64_t c) -> int {
    const int set = (int)(c & 1);
    if (two && c >= 2) HIPCHK(hipStreamWaitEvent(gs, w->ev_used[set], 0));      // chunk c - 2 has read this set
    if (onepass) {
      TRY(nps_onepass(h, w->one[2 * set], (size_t)c * segs, inc, 2 * nc, gs));
      if (sh) TRY(nps_onepass(h, w->one[2 * set + 1], (size_t)c * segs + 1, inc, 2 * ns, gs));
      if (two) HIPCHK(hipEventRecord(w->ev_gen[set], gs));
      return 0;
    }
    TRY(nps_segment(h, AA[2 * c], w->seg[2 * set], (size_t)c * segs, inc, 2 * nc, gs));      // real parts of the chunk, then its imaginary parts
    if (sh) TRY(nps_segment(h, AA[2 * c + 1], w->seg[2 * set + 1], (size_t)c * segs + 1, inc, 2 * ns, gs));
    if (two) HIPCHK(hipEventRecord(w->ev_gen[set], gs));
    return 0;
  };
  static const bool times = getenv("FASTMC_NPS_TIMES") != nullptr;     // stderr: how long the host took to ENQUEUE the chunks, and the whole call
  const auto t_enq0 = std::chrono::steady_clock::now();
  TRY(gen_chunk(0));
  for (int64_t c = 0; c < n_chunks; ++c) {
    const int set = (int)(c & 1);
    NpsSegArgs* A = &AA[2 * c];
    if (two) HIPCHK(hipStreamWaitEvent(h->stream, w->ev_gen[set], 0));
    if (c + 1 < n_chunks) TRY(gen_chunk(c + 1));
    RunSpec S{1, 0, 0, 0, chunk_real, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, coherent, nullptr, nullptr};
    S.async = true;
    S.logamp_dev = w->la + logamp_offset + (size_t)c * 2 * chunk_real;
    if (onepass) {
      S.coef_dev_re = w->one[2 * set].out; S.coef_dev_im = w->one[2 * set].out + nc;
      if (sh) { S.sh_dev_re = w->one[2 * set + 1].out; S.sh_dev_im = w->one[2 * set + 1].out + ns; }
    } else S.fill = [&, A, h, sh, N2, nc, ns](int64_t bs, int nb) -> int {
      nps_emit2(h, A[0], nps_range(A[0], (uint64_t)bs * N2, (uint64_t)(bs + nb) * N2, h->cre),
                nps_range(A[0], nc + (uint64_t)bs * N2, nc + (uint64_t)(bs + nb) * N2, h->cim));
      if (sh)
        nps_emit2(h, A[1], nps_range(A[1], (uint64_t)bs * 27, (uint64_t)(bs + nb) * 27, h->sh_in_re),
                  nps_range(A[1], ns + (uint64_t)bs * 27, ns + (uint64_t)(bs + nb) * 27, h->sh_in_im));
      return 0;
    };
    h->pending = false;          // no wait between chunks: only the last chunk's kernel times are read (after the one sync below)
    S.out_dev = w->res + per_chunk * c;      // every chunk's results stay on the device until the one copy below (round 6: one launch gap less per chunk)
    TRY(run_impl<double>(h, S));
    if (two) HIPCHK(hipEventRecord(w->ev_used[set], h->stream));
  }
  HIPCHK(hipMemcpyAsync(pinned, w->res, per_chunk * n_chunks * 8, hipMemcpyDeviceToHost, h->stream));
  const auto t_enq1 = std::chrono::steady_clock::now();
  if (two) HIPCHK(hipStreamSynchronize(w->gstream));
  std::vector<uint32_t> ovf((size_t)n_chunks * segs);
  std::vector<u128> states((size_t)n_chunks * segs + 1);
  HIPCHK(hipMemcpyAsync(ovf.data(), w->overflow, ovf.size() * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(states.data(), w->states, states.size() * sizeof(u128), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (times) {
    const auto t_end = std::chrono::steady_clock::now();
    fprintf(stderr, "fastmc_run_npstream: %lld chunks enqueued in %.3f ms (%.1f us per chunk), call %.3f ms (%.1f us per chunk)\n", (long long)n_chunks,
            std::chrono::duration<double, std::milli>(t_enq1 - t_enq0).count(), std::chrono::duration<double, std::micro>(t_enq1 - t_enq0).count() / n_chunks,
            std::chrono::duration<double, std::milli>(t_end - t_enq0).count(), std::chrono::duration<double, std::micro>(t_end - t_enq0).count() / n_chunks);
  }
  finish_pending(h);
  HIPCHK(hipGetLastError());
  {   // FASTMC_NPS_TEST_OVERFLOW=k (tests): the first call of the process with more than k chunks reports chunk k as given up
    static std::atomic<int> armed{[] { const char* e = getenv("FASTMC_NPS_TEST_OVERFLOW"); return e ? atoi(e) : -1; }()};
    int k = armed.load();
    if (k >= 0 && n_chunks > k) {
      int expect = k;
      if (armed.compare_exchange_strong(expect, -1)) ovf[(size_t)k * segs] |= 0x80u;
    }
  }
  int64_t bad = -1;
  for (int64_t c = 0; c < n_chunks && bad < 0; ++c)
    for (int k = 0; k < segs; ++k)
      if (ovf[(size_t)c * segs + k]) { bad = c; break; }
  const int64_t good = bad < 0 ? n_chunks : bad;
  memcpy(out, pinned, per_chunk * good * 8);
  const u128 st = states[(size_t)good * segs];
  if (state_after) { state_after[0] = (uint64_t)st; state_after[1] = (uint64_t)(st >> 64); }
  if (bad_chunk) *bad_chunk = bad;
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_run_coeffs(fastmc_t* h, const double* coeff_re, const double* coeff_im, int64_t n_real,
                                 const double* sh_re, const double* sh_im, const double* logamp, int coherent,
                                 double* out) {
  if (!coeff_re || !coeff_im || !logamp || !out) return fail(FASTMC_EINVAL, "null argument");
  RunSpec S{1, 0, 0, 0, n_real, coeff_re, coeff_im, sh_re, sh_im, logamp, 0.0, coherent, out, nullptr};
  return run_checked(h, S);
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_screens_coeffs(fastmc_t* h, const double* coeff_re, const double* coeff_im, int64_t n_real,
                                     const double* sh_re, const double* sh_im, double* phs) {
  if (!coeff_re || !coeff_im || !phs) return fail(FASTMC_EINVAL, "null argument");
  RunSpec S{1, 1, 0, 0, n_real, coeff_re, coeff_im, sh_re, sh_im, nullptr, 0.0, 0, nullptr, phs};
  return run_checked(h, S);
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_screens(fastmc_t* h, uint64_t seed, int64_t real0, int64_t n_real, double* phs) {
  if (!phs) return fail(FASTMC_EINVAL, "null argument");
  RunSpec S{0, 1, seed, real0, n_real, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, 0, nullptr, phs};
  return run_checked(h, S);
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_rng_coeffs(fastmc_t* h, uint64_t seed, int64_t real, double* out) {
  if (!h || !out) return fail(FASTMC_EINVAL, "null argument");
  HIPCHK(hipSetDevice(h->device));
  if (real < 0) return fail(FASTMC_EINVAL, "realisation index must be non-negative");
  const int N = h->N;
  TRY(ensure_gen64_table(h));
  ScratchBuf d;
  HIPCHK(hipMalloc((void**)&d.p, (size_t)N * N * 16));
  RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32)};
  hipLaunchKernelGGL(k_rng_coeffs, dim3((N * stream_lanes(N) + 255) / 256), dim3(256), 0, h->stream, key, (uint64_t)real, N, h->rng_f64, h->g64, d.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, d.p, (size_t)N * N * 16, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_rng_logamp(fastmc_t* h, uint64_t seed, int64_t iter0, int64_t n_iter, double* out) {
  if (!h || !out || n_iter <= 0) return fail(FASTMC_EINVAL, "bad argument");
  HIPCHK(hipSetDevice(h->device));
  if (iter0 < 0) return fail(FASTMC_EINVAL, "iteration index must be non-negative");
  ScratchBuf d;
  HIPCHK(hipMalloc((void**)&d.p, (size_t)n_iter * 8));
  RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32)};
  hipLaunchKernelGGL(k_rng_logamp, dim3((unsigned)((n_iter + 255) / 256)), dim3(256), 0, h->stream, key, (uint64_t)iter0,
                     n_iter, h->rng_f64, d.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, d.p, (size_t)n_iter * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_set_layer_screens(fastmc_t* h, const double* screens, int n_layers) {
  if (!h || !screens || n_layers < 1 || n_layers > 1024) return fail(FASTMC_EINVAL, "bad argument");
  HIPCHK(hipSetDevice(h->device));
  const size_t n = (size_t)n_layers * h->N * h->N;
  if (h->layers) { HIPCHK(hipFree(h->layers)); h->layers = nullptr; h->n_layers = 0; }
  HIPCHK(hipMalloc((void**)&h->layers, n * 8));
  HIPCHK(hipMemcpy(h->layers, screens, n * 8, hipMemcpyHostToDevice));
  h->n_layers = n_layers;
  return 0;
}
#endif

#if FMC_TU == 0
// one chunk of the frozen-flow series: detector results (out != NULL) and / or the phases themselves (phs != NULL)
static int temporal_impl(fastmc_ctx* h, const double* xs, const double* ys, const int32_t* roll, int M, const double* logamp,
                         int coherent, double* out, double* phs) {
  if (!h || !xs || !ys || !roll || M < 1 || (!out && !phs) || (out && !logamp)) return fail(FASTMC_EINVAL, "bad argument");
  if (!h->have_pupil) return fail(FASTMC_ESTATE, "set_pupil must be called first");
  if (!h->layers) return fail(FASTMC_ESTATE, "set_layer_screens must be called first");
  if (h->N < 2) return fail(FASTMC_EINVAL, "N must be at least 2");
  HIPCHK(hipSetDevice(h->device));
  const int L = h->n_layers, Np = h->Np;
  const size_t nc = (size_t)L * M * Np, np2 = (size_t)M * Np * Np;
  // one scratch slab per call: [xs nc][ys nc][logamp M][out 2 M][roll (L 2 M ints, padded)][phs M Np Np]
  const size_t o_ys = nc, o_la = 2 * nc, o_out = o_la + M, o_roll = o_out + 2 * (size_t)M, o_phs = o_roll + ((size_t)L * 2 * M + 1) / 2 + 1;
  ScratchBuf d;
  HIPCHK(hipMalloc((void**)&d.p, (o_phs + (phs ? np2 : 0)) * 8));
  HIPCHK(hipMemcpyAsync(d.p, xs, nc * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d.p + o_ys, ys, nc * 8, hipMemcpyHostToDevice, h->stream));
  if (out) HIPCHK(hipMemcpyAsync(d.p + o_la, logamp, (size_t)M * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d.p + o_roll, roll, (size_t)L * 2 * M * 4, hipMemcpyHostToDevice, h->stream));
  TemporalArgs A;
  A.N = h->N; A.Np = Np; A.L = L; A.M = M; A.coherent = coherent;
  A.screens = h->layers; A.xs = d.p; A.ys = d.p + o_ys; A.roll = (const int*)(d.p + o_roll); A.W = h->W; A.logamp = d.p + o_la;
  A.dx2 = h->dx * h->dx; A.norm = h->wsum * (h->dx * h->dx); A.out = out ? d.p + o_out : nullptr; A.phs = phs ? d.p + o_phs : nullptr;
  guard_outputs(h);
  hipLaunchKernelGGL(k_temporal_detect, dim3(M), dim3(256), 0, h->stream, A);
  HIPCHK(hipGetLastError());
  if (out) HIPCHK(hipMemcpyAsync(out, d.p + o_out, (size_t)M * 8 * (coherent ? 2 : 1), hipMemcpyDeviceToHost, h->stream));
  if (phs) HIPCHK(hipMemcpyAsync(phs, d.p + o_phs, np2 * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int fastmc_temporal_chunk(fastmc_t* h, const double* xs, const double* ys, const int32_t* roll, int M,
                                     const double* logamp, int coherent, double* out) {
  if (!out || !logamp) return fail(FASTMC_EINVAL, "bad argument");
  return temporal_impl(h, xs, ys, roll, M, logamp, coherent, out, nullptr);
}

extern "C" int fastmc_temporal_phases(fastmc_t* h, const double* xs, const double* ys, const int32_t* roll, int M, double* phs) {
  if (!phs) return fail(FASTMC_EINVAL, "bad argument");
  return temporal_impl(h, xs, ys, roll, M, nullptr, 0, nullptr, phs);
}
#endif

static int histogram_device(fastmc_ctx* h, double lo, double hi, int nbins) {
  if (h->last_n_iter <= 0) return fail(FASTMC_ESTATE, "no run results on the device");
  if (nbins < 1 || !(hi > lo)) return fail(FASTMC_EINVAL, "bad histogram range");
  if (h->hist_cap < (size_t)nbins + 2) {
    if (h->hist) HIPCHK(hipFree(h->hist));
    HIPCHK(hipMalloc((void**)&h->hist, ((size_t)nbins + 2) * 8));
    h->hist_cap = (size_t)nbins + 2;
  }
  guard_outputs(h);
  HIPCHK(hipMemsetAsync(h->hist, 0, ((size_t)nbins + 2) * 8, h->stream));
  const int64_t n = h->last_n_iter;
  hipLaunchKernelGGL(k_histogram, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->out, n, h->last_coherent, lo,
                     hi, nbins, h->hist);
  return 0;
}

#if FMC_TU == 0
extern "C" int fastmc_set_results(fastmc_t* h, const double* values, int64_t n_iter, int coherent) {
  if (!h || !values || n_iter < 1) return fail(FASTMC_EINVAL, "bad argument");
  FMC_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  const size_t need = (size_t)n_iter * (coherent ? 2 : 1);
  TRY(grow(&h->out, &h->out_cap, need));
  guard_outputs(h);
  HIPCHK(hipMemcpyAsync(h->out, values, need * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->last_n_iter = n_iter;
  h->last_coherent = coherent ? 1 : 0;
  h->last_out_doubles = need;
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_histogram(fastmc_t* h, double lo_db, double hi_db, int nbins, int64_t* bins) {
  if (!h || !bins) return fail(FASTMC_EINVAL, "null argument");
  FMC_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  TRY(histogram_device(h, lo_db, hi_db, nbins));
  HIPCHK(hipMemcpyAsync(bins, h->hist, ((size_t)nbins + 2) * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_result_stats(fastmc_t* h, const double* thresholds, int n_thr, double* stats) {
  if (!h || !stats || n_thr < 0 || n_thr > STATS_MAX_THR || (n_thr > 0 && !thresholds)) return fail(FASTMC_EINVAL, "bad argument");
  if (h->last_n_iter <= 0) return fail(FASTMC_ESTATE, "no run results on the device");
  FMC_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  const int nblocks = 256, stride = STATS_NQ + STATS_MAX_THR, nq = STATS_NQ + n_thr;
  ScratchBuf part, thr, res;
  HIPCHK(hipMalloc((void**)&part.p, (size_t)nblocks * stride * 8));
  HIPCHK(hipMalloc((void**)&thr.p, (size_t)STATS_MAX_THR * 8));
  HIPCHK(hipMalloc((void**)&res.p, (size_t)stride * 8));
  if (n_thr) HIPCHK(hipMemcpyAsync(thr.p, thresholds, (size_t)n_thr * 8, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_stats_partial, dim3(nblocks), dim3(256), 0, h->stream, h->out, h->last_n_iter, h->last_coherent,
                     thr.p, n_thr, part.p);
  hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(64), 0, h->stream, part.p, nblocks, nq, res.p);
  HIPCHK(hipGetLastError());
  stats[0] = (double)h->last_n_iter;
  HIPCHK(hipMemcpyAsync(stats + 1, res.p, (size_t)nq * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_link_metrics(fastmc_t* h, int device_id, const double* samples, int64_t n, const fastmc_link_query* queries,
                                   int n_queries, double* out) {
  if (!queries || !out || n_queries < 1) return fail(FASTMC_EINVAL, "bad argument");
  for (int k = 0; k < n_queries; ++k) {
    const fastmc_link_query& q = queries[k];
    if (q.kind < LM_FADE || q.kind > LM_SEP_QAM) return fail(FASTMC_EINVAL, "unknown link query kind");
    if (q.kind == LM_SEP_QAM && !(q.p0 > 1.0)) return fail(FASTMC_EINVAL, "QAM order must exceed 1");
  }
  hipStream_t stream = nullptr;
  const double* x = nullptr;
  int coherent = 0;
  ScratchBuf up;
  if (samples) {
    if (n < 1) return fail(FASTMC_EINVAL, "empty sample vector");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(FASTMC_ENODEV, "no HIP device");
    if (device_id < 0 || device_id >= ndev) return fail(FASTMC_EINVAL, "device_id out of range");
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipMalloc((void**)&up.p, (size_t)n * 8));
    HIPCHK(hipMemcpy(up.p, samples, (size_t)n * 8, hipMemcpyHostToDevice));
    x = (const double*)up.p;
  } else {
    if (!h) return fail(FASTMC_EINVAL, "neither samples nor a handle");
    if (h->last_n_iter <= 0) return fail(FASTMC_ESTATE, "no run results on the device");
    HIPCHK(hipSetDevice(h->device));
    stream = h->stream;
    x = h->out;
    n = h->last_n_iter;
    coherent = h->last_coherent;
  }
  const int nblocks = (int)std::min<int64_t>(256, (n + 255) / 256), stride = STATS_NQ + STATS_MAX_THR;
  ScratchBuf part, sums, res;
  HIPCHK(hipMalloc((void**)&part.p, (size_t)256 * stride * 8));
  HIPCHK(hipMalloc((void**)&sums.p, (size_t)stride * 8));
  HIPCHK(hipMalloc((void**)&res.p, (size_t)n_queries * 4 * 8));
  // sum of the vector (for the mean the BER / SEP integrals normalise by): the statistics pass
  hipLaunchKernelGGL(k_stats_partial, dim3(256), dim3(256), 0, stream, x, n, coherent, (const double*)nullptr, 0, (double*)part.p);
  hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(64), 0, stream, (const double*)part.p, 256, STATS_NQ, (double*)sums.p);
  for (int k = 0; k < n_queries; ++k) {
    hipLaunchKernelGGL(k_link_query, dim3(nblocks), dim3(256), 0, stream, x, n, coherent, (const double*)sums.p, (int)queries[k].kind,
                       queries[k].p0, queries[k].p1, (double*)part.p);
    hipLaunchKernelGGL(k_link_final, dim3(1), dim3(64), 0, stream, (const double*)part.p, nblocks, n, (const double*)sums.p,
                       (int)queries[k].kind, (double*)res.p + 4 * k);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, res.p, (size_t)n_queries * 4 * 8, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_last_timing(fastmc_t* h, double* times_ms, int64_t* launches) {
  if (!h || !times_ms || !launches) return fail(FASTMC_EINVAL, "null argument");
  for (int i = 0; i < 4; ++i) { times_ms[i] = h->t_ms[i]; launches[i] = h->t_n[i]; }
  return 0;
}
#endif

// ------------------------------------------------------------------ power spectrum
static void noll_to_nm(int j, int* n_out, int* m_out) {   // aotools zernIndex (third party; see hostmath.noll_to_nm)
  const int n = (int)((-1.0 + std::sqrt(8.0 * (j - 1) + 1.0)) / 2.0);
  const double pp = j - (n * (n + 1)) / 2.0;
  const int k = n % 2;
  int m = (int)((pp + k) / 2.0) * 2 - k;
  if (m != 0 && (j % 2) != 0) m = -m;
  *n_out = n;
  *m_out = m;
}

// Device scratch of the power-spectrum evaluation, one per device for the life of the process: sweeps build one
// Fast object per geometry sample (fast/complete_orbit_simulation.py:217-228) and ten hipMalloc / hipFree pairs plus
// null-stream synchronisations per object cost more than the kernel (0.2 ms).  Buffers only grow; its own stream.
struct PsScratch {
  std::mutex mu;
  hipStream_t stream = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct Buf { double* p = nullptr; size_t cap = 0; };
  Buf small, ps, la, rows, sc, mask, mo, pf, z, pl, turb, g, al, no;
  int64_t pf_token = 0;      // identity of the pupil filter resident in `pf` (fastmc_ps_params.pupil_filter_token)
  int pf_N = 0;
  int ensure(Buf& b, size_t n) {
    if (b.cap >= n) return 0;
    if (b.p) hipFree(b.p);
    b.p = nullptr; b.cap = 0;
    HIPCHK(hipMalloc((void**)&b.p, n * 8));
    b.cap = n;
    return 0;
  }
};
static PsScratch g_ps_scratch[64];

// Evaluate the spectrum on `device_id` (stream of the device's scratch).  Host outputs (any may be null) are copied
// back; with `into` the spectrum also becomes that handle's colouring tables and stays resident on it
// (fastmc_powerspec_set), with no host round trip of the N x N grids.
static int make_amp_from_device(fastmc_ctx* h, const double* d_ps, double df, hipStream_t stream);

static int powerspec_impl(int device_id, const fastmc_ps_params* p, double* powerspec, double* per_layer, double* logamp_ps,
                          double* lf_mask_out, double* scalars, double* kernel_ms, double* turb, double* g_ao, double* alias_ps,
                          double* noise_ps, fastmc_ctx* into = nullptr, double df = 0.0) {
  if (!p) return fail(FASTMC_EINVAL, "params is NULL");
  const int N = p->N, L = p->n_layers;
  if (N < 2 || N > 16384) return fail(FASTMC_EINVAL, "bad N");
  if (L < 1 || L > PS_MAX_LAYERS) return fail(FASTMC_EINVAL, "n_layers must be in [1, 64]");
  if (!p->cn2 || !p->h || !p->wind || !p->simpson_w) return fail(FASTMC_EINVAL, "null array in params");
  if (p->mask_mode < 0 || p->mask_mode > 3) return fail(FASTMC_EINVAL, "bad mask_mode");
  if (p->mask_mode == 0 && !p->lf_mask) return fail(FASTMC_EINVAL, "mask_mode 0 needs lf_mask");
  if (p->mask_mode == 3 && (p->zmax < 1 || p->zmax > 1024)) return fail(FASTMC_EINVAL, "zmax must be in [1, 1024]");
  if (p->ao_mode < 0 || p->ao_mode > 3) return fail(FASTMC_EINVAL, "bad ao_mode");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(FASTMC_ENODEV, "no HIP device visible: libfastmc has no CPU fallback");
  if (device_id < 0 || device_id >= count) return fail(FASTMC_EINVAL, "device_id out of range");
  HIPCHK(hipSetDevice(device_id));
  PsScratch& S = g_ps_scratch[device_id & 63];
  std::lock_guard<std::mutex> guard(S.mu);
  if (!S.stream) {
    HIPCHK(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&S.e0));
    HIPCHK(hipEventCreate(&S.e1));
  }
  hipStream_t st = S.stream;
  const size_t N2 = (size_t)N * N;
  const int nq = PS_NQ + L;
  const int nmodes = std::max(p->mask_mode == 3 ? p->zmax : 0, 4);
  const bool want_mask = lf_mask_out || into;
  // small inputs packed into one upload: [cn2 L][h L][wind 2L][w N][noll 2*nmodes ints, padded to doubles]
  const size_t o_cn2 = 0, o_h = L, o_wind = 2 * (size_t)L, o_w = 4 * (size_t)L, o_noll = 4 * (size_t)L + N;
  const size_t n_small = o_noll + nmodes + 1;
  TRY(S.ensure(S.small, n_small));
  TRY(S.ensure(S.ps, N2));
  TRY(S.ensure(S.la, N2));
  TRY(S.ensure(S.rows, (size_t)N * nq));
  TRY(S.ensure(S.sc, nq));
  if (p->mask_mode == 0) TRY(S.ensure(S.mask, N2));
  if (want_mask) TRY(S.ensure(S.mo, N2));
  if (p->pupil_filter) TRY(S.ensure(S.pf, N2));
  if (p->lgs_z) TRY(S.ensure(S.z, N2));
  if (per_layer) TRY(S.ensure(S.pl, N2 * L));
  if (turb) TRY(S.ensure(S.turb, N2 * L));
  if (g_ao) TRY(S.ensure(S.g, N2 * L));
  if (alias_ps) TRY(S.ensure(S.al, N2 * L));
  if (noise_ps) TRY(S.ensure(S.no, N2));
  std::vector<double> small(n_small, 0.0);
  memcpy(&small[o_cn2], p->cn2, L * 8);
  memcpy(&small[o_h], p->h, L * 8);
  memcpy(&small[o_wind], p->wind, 2 * (size_t)L * 8);
  memcpy(&small[o_w], p->simpson_w, (size_t)N * 8);
  {
    int* nm = reinterpret_cast<int*>(&small[o_noll]);
    for (int j = 1; j <= nmodes; ++j) noll_to_nm(j, &nm[j - 1], &nm[nmodes + j - 1]);
  }
  HIPCHK(hipMemcpyAsync(S.small.p, small.data(), n_small * 8, hipMemcpyHostToDevice, st));
  if (p->mask_mode == 0) HIPCHK(hipMemcpyAsync(S.mask.p, p->lf_mask, N2 * 8, hipMemcpyHostToDevice, st));
  if (p->pupil_filter) {
    // the filter depends on the aperture only: a caller that re-uses one array names it with a token and it is uploaded once
    const bool resident = p->pupil_filter_token != 0 && p->pupil_filter_token == S.pf_token && S.pf_N == N;
    if (!resident) {
      HIPCHK(hipMemcpyAsync(S.pf.p, p->pupil_filter, N2 * 8, hipMemcpyHostToDevice, st));
      S.pf_token = p->pupil_filter_token;
      S.pf_N = N;
    }
  }
  if (p->lgs_z) HIPCHK(hipMemcpyAsync(S.z.p, p->lgs_z, N2 * 8, hipMemcpyHostToDevice, st));
  PsArgs K;
  K.N = N; K.L = L; K.ao_mode = p->ao_mode; K.alias = p->alias;
  K.dx = p->dx; K.wvl = p->wvl; K.L0 = p->L0; K.l0 = p->l0; K.noise = p->noise; K.d_wfs = p->d_wfs;
  K.t_loop = p->t_loop; K.t_exp = p->t_exp; K.dth_x = p->dtheta[0]; K.dth_y = p->dtheta[1];
  K.cn2 = S.small.p + o_cn2; K.h = S.small.p + o_h; K.wind = S.small.p + o_wind; K.w = S.small.p + o_w;
  K.mask = p->mask_mode == 0 ? S.mask.p : nullptr;
  K.pfilter = p->pupil_filter ? S.pf.p : nullptr;
  K.lgs_z = p->lgs_z ? S.z.p : nullptr;
  K.mask_mode = p->mask_mode; K.zmax = p->zmax; K.modal_mult = p->modal_mult; K.D_zern = p->D_ground;
  K.noll_n = reinterpret_cast<const int*>(S.small.p + o_noll); K.noll_m = K.noll_n + nmodes;
  K.mask_out = want_mask ? S.mo.p : nullptr;
  K.powerspec = S.ps.p; K.per_layer = per_layer ? S.pl.p : nullptr; K.logamp_ps = S.la.p; K.rowsums = S.rows.p;
  K.turb = turb ? S.turb.p : nullptr; K.g_ao = g_ao ? S.g.p : nullptr; K.alias_out = alias_ps ? S.al.p : nullptr;
  K.noise_out = noise_ps ? S.no.p : nullptr;
  HIPCHK(hipEventRecord(S.e0, st));
  hipLaunchKernelGGL(k_powerspec, dim3(N), dim3(PS_THREADS), 0, st, K);
  hipLaunchKernelGGL(k_ps_scalars, dim3(nq), dim3(256), 0, st, S.rows.p, K.w, N, nq, S.sc.p);
  HIPCHK(hipEventRecord(S.e1, st));
  HIPCHK(hipGetLastError());
  std::vector<double> sc(nq);
  if (scalars) HIPCHK(hipMemcpyAsync(sc.data(), S.sc.p, nq * 8, hipMemcpyDeviceToHost, st));
  if (powerspec) HIPCHK(hipMemcpyAsync(powerspec, S.ps.p, N2 * 8, hipMemcpyDeviceToHost, st));
  if (per_layer) HIPCHK(hipMemcpyAsync(per_layer, S.pl.p, N2 * L * 8, hipMemcpyDeviceToHost, st));
  if (logamp_ps) HIPCHK(hipMemcpyAsync(logamp_ps, S.la.p, N2 * 8, hipMemcpyDeviceToHost, st));
  if (lf_mask_out) HIPCHK(hipMemcpyAsync(lf_mask_out, S.mo.p, N2 * 8, hipMemcpyDeviceToHost, st));
  if (turb) HIPCHK(hipMemcpyAsync(turb, S.turb.p, N2 * L * 8, hipMemcpyDeviceToHost, st));
  if (g_ao) HIPCHK(hipMemcpyAsync(g_ao, S.g.p, N2 * L * 8, hipMemcpyDeviceToHost, st));
  if (alias_ps) HIPCHK(hipMemcpyAsync(alias_ps, S.al.p, N2 * L * 8, hipMemcpyDeviceToHost, st));
  if (noise_ps) HIPCHK(hipMemcpyAsync(noise_ps, S.no.p, N2 * 8, hipMemcpyDeviceToHost, st));
  if (into) {
    // keep the three grids a Fast object exposes (powerspec, logamp_powerspec, lf_mask) on the handle, and colour from them
    if (!into->ps_dev) HIPCHK(hipMalloc((void**)&into->ps_dev, 3 * N2 * 8));
    HIPCHK(hipMemcpyAsync(into->ps_dev, S.ps.p, N2 * 8, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(into->ps_dev + N2, S.la.p, N2 * 8, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(into->ps_dev + 2 * N2, S.mo.p, N2 * 8, hipMemcpyDeviceToDevice, st));
    TRY(make_amp_from_device(into, into->ps_dev, df, st));      // synchronises `st`
    into->have_ps = true;
  } else {
    HIPCHK(hipStreamSynchronize(st));
  }
  if (kernel_ms) {
    float ms = 0;
    hipEventElapsedTime(&ms, S.e0, S.e1);
    *kernel_ms = ms;
  }
  if (scalars) {
    for (int i = 0; i < PS_NQ; ++i) scalars[i] = sc[i];
    for (int l = 0; l < L; ++l) scalars[PS_NQ + l] = sc[PS_NQ + l] / sc[4];   // phs_var_weights
  }
  return 0;
}

#if FMC_TU == 0
extern "C" int fastmc_powerspec(int device_id, const fastmc_ps_params* p, double* powerspec, double* per_layer,
                                double* logamp_ps, double* lf_mask_out, double* scalars, double* kernel_ms) {
  return powerspec_impl(device_id, p, powerspec, per_layer, logamp_ps, lf_mask_out, scalars, kernel_ms, nullptr, nullptr, nullptr,
                        nullptr);
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_powerspec_terms(int device_id, const fastmc_ps_params* p, double* turb, double* g_ao, double* alias_ps,
                                      double* noise_ps) {
  if (!turb && !g_ao && !alias_ps && !noise_ps) return fail(FASTMC_EINVAL, "no output requested");
  return powerspec_impl(device_id, p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, turb, g_ao, alias_ps, noise_ps);
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_powerspec_set(fastmc_t* h, const fastmc_ps_params* p, double df, double* scalars, double* kernel_ms) {
  if (!h || !p) return fail(FASTMC_EINVAL, "null argument");
  if (p->N != h->N) return fail(FASTMC_EINVAL, "params.N differs from the handle's grid");
  return powerspec_impl(h->device, p, nullptr, nullptr, nullptr, nullptr, scalars, kernel_ms, nullptr, nullptr, nullptr, nullptr, h, df);
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_powerspec_get(fastmc_t* h, int which, double* out) {
  if (!h || !out || which < 0 || which > 2) return fail(FASTMC_EINVAL, "bad argument");
  if (!h->ps_dev || !h->have_ps) return fail(FASTMC_ESTATE, "fastmc_powerspec_set has not been called on this handle");
  HIPCHK(hipSetDevice(h->device));
  const size_t N2 = (size_t)h->N * h->N;
  HIPCHK(hipMemcpyAsync(out, h->ps_dev + (size_t)which * N2, N2 * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
#endif

// ------------------------------------------------------------------ RCCL (loaded on demand)
// One communicator per DEVICE of this process, shared by every handle on that device: sweeps build many
// short-lived handles, and a communicator costs a rendezvous.  Two ways to create them:
//   fastmc_comm_init      one process per GPU (a launcher distributes the unique id), ncclCommInitRank;
//   fastmc_comm_init_all  one process drives n devices, ncclCommInitAll.
struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId;
  decltype(&ncclCommInitRank) CommInitRank;
  decltype(&ncclCommInitAll) CommInitAll;
  decltype(&ncclCommDestroy) CommDestroy;
  decltype(&ncclCommAbort) CommAbort;
  decltype(&ncclAllGather) AllGather;
  decltype(&ncclAllReduce) AllReduce;
  decltype(&ncclGroupStart) GroupStart;
  decltype(&ncclGroupEnd) GroupEnd;
  decltype(&ncclGetErrorString) GetErrorString;
  void* lib = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;
static int load_rccl() {
  std::lock_guard<std::mutex> g(g_rccl_mu);
  if (g_rccl.lib) return 0;
  if (const char* off = getenv("FASTMC_DISABLE_RCCL"))
    if (off[0] && off[0] != '0') return fail(FASTMC_ECOMM, "RCCL disabled by FASTMC_DISABLE_RCCL");
  // FASTMC_RCCL_LIB: the library to load instead (a path; the tests' stand-in, or an RCCL build that is not on the loader's path)
  void* lib = nullptr;
  if (const char* path = getenv("FASTMC_RCCL_LIB")) {
    if (path[0]) {
      lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
      if (!lib) return fail(FASTMC_ECOMM, std::string("cannot load FASTMC_RCCL_LIB: ") + dlerror());
    }
  }
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return fail(FASTMC_ECOMM, std::string("cannot load librccl: ") + dlerror());
#define SYM(n)                                                        \
  g_rccl.n = (decltype(g_rccl.n))dlsym(lib, "nccl" #n);                \
  if (!g_rccl.n) return fail(FASTMC_ECOMM, "librccl lacks nccl" #n);
  SYM(GetUniqueId) SYM(CommInitRank) SYM(CommInitAll) SYM(CommDestroy) SYM(CommAbort) SYM(AllGather) SYM(AllReduce)
  SYM(GroupStart) SYM(GroupEnd) SYM(GetErrorString)
#undef SYM
  g_rccl.lib = lib;
  return 0;
}
#define NCCLCHK(expr)                                                                          \
  do {                                                                                         \
    ncclResult_t r__ = (expr);                                                                 \
    if (r__ != ncclSuccess) return fail(FASTMC_ECOMM, std::string(#expr) + ": " + g_rccl.GetErrorString(r__)); \
  } while (0)

struct DeviceComm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
};
static DeviceComm g_comm[64];
static std::mutex g_comm_mu;
// A device's communicator is handed to RCCL only under this lock (read g_comm -> enqueue the collectives), and
// fastmc_comm_abort takes it (with a deadline of two seconds: enqueueing does not wait for peers) before ncclCommAbort frees
// the communicator: no thread is ever between "copied the communicator" and "passed it to ncclAllGather" when it is freed.
// What blocks when a peer is gone is the hipStreamSynchronize AFTER the enqueue, outside this lock.
static std::timed_mutex g_enq_mu[64];
// A communicator belongs to a DEVICE: slot device & 63 of the tables above (and of g_abort_gen).  The fake-RCCL tests
// (FASTMC_RCCL_LIB + FASTMC_TEST_VIRTUAL_RANKS=1: tests/stubs/fake_rccl.cpp) put several ranks on the one device a test box has;
// their handles carry a slot of their own (32 + rank) so that the group / stream / ordering code below runs with a world of up
// to 8 where RCCL itself cannot.  Never set with the real library.
static int cslot(const fastmc_ctx* h) { return h->comm_slot >= 0 ? h->comm_slot : (h->device & 63); }
// ... and only when the library that was loaded IS the tests' stand-in: it exports `fake_rccl_marker`; a real RCCL named by
// FASTMC_RCCL_LIB (a build off the loader's path) never takes several ranks on one device whatever the environment says
static bool virtual_ranks_allowed() {
  const char* a = getenv("FASTMC_RCCL_LIB");
  const char* b = getenv("FASTMC_TEST_VIRTUAL_RANKS");
  if (!(a && a[0] && b && b[0] && b[0] != '0')) return false;
  if (load_rccl() != 0) return false;
  return dlsym(g_rccl.lib, "fake_rccl_marker") != nullptr;
}

// FASTMC_TEST_STALL_GATHER=1: the exchange entry points block, without touching RCCL, until fastmc_comm_abort is called for
// one of the devices, and then fail -- the fault that the deadline / fall-back tests inject.  =2: the same AFTER the
// collectives have been enqueued on a real communicator (the caller's abort then races a thread that holds the handle and
// has RCCL work on its stream: what a lost peer looks like from inside).
static int stall_mode() {
  const char* e = getenv("FASTMC_TEST_STALL_GATHER");
  return (e && e[0] && e[0] != '0') ? (e[0] == '2' ? 2 : 1) : 0;
}
static bool stall_requested() { return stall_mode() == 1; }
static int stall_until_abort(const std::vector<int>& devices) {
  std::vector<int> gen(devices.size());
  for (size_t i = 0; i < devices.size(); ++i) gen[i] = g_abort_gen[devices[i] & 63].load();
  for (;;) {
    for (size_t i = 0; i < devices.size(); ++i)
      if (g_abort_gen[devices[i] & 63].load() != gen[i]) return fail(FASTMC_ECOMM, "exchange aborted (FASTMC_TEST_STALL_GATHER)");
    usleep(500);
  }
}
static int exchange_begin(fastmc_ctx* h) {
  if (!h->ex_a) {
    HIPCHK(hipEventCreate(&h->ex_a));
    HIPCHK(hipEventCreate(&h->ex_b));
  }
  HIPCHK(hipEventRecord(h->ex_a, h->stream));
  return 0;
}
static int exchange_end(fastmc_ctx* h) {
  HIPCHK(hipEventRecord(h->ex_b, h->stream));
  h->ex_recorded = true;
  return 0;
}
static DeviceComm device_comm(int slot) {
  std::lock_guard<std::mutex> g(g_comm_mu);
  return g_comm[slot & 63];
}

#if FMC_TU == 0
extern "C" int fastmc_comm_unique_id(uint8_t id128[128]) {
  if (!id128) return fail(FASTMC_EINVAL, "null id");
  TRY(load_rccl());
  ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == 128, "unique id size");
  memcpy(id128, &id, 128);
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_comm_init(fastmc_t* h, const uint8_t id128[128], int world_size, int rank) {
  if (!h || !id128 || world_size < 1 || rank < 0 || rank >= world_size) return fail(FASTMC_EINVAL, "bad argument");
  TRY(load_rccl());
  if (device_comm(cslot(h)).comm) return fail(FASTMC_ESTATE, "this device already has a communicator (fastmc_comm_destroy first)");
  HIPCHK(hipSetDevice(h->device));
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclComm_t c = nullptr;
  const int gen = g_abort_gen[cslot(h)].load();
  NCCLCHK(g_rccl.CommInitRank(&c, world_size, id, rank));
  if (g_abort_gen[cslot(h)].load() != gen) {
    // fastmc_comm_abort was called while the clique was being built (the caller's deadline passed and it took the host
    // path): the communicator must not appear now
    g_rccl.CommAbort(c);
    return fail(FASTMC_ECOMM, "communicator aborted while it was being initialised");
  }
  std::lock_guard<std::mutex> g(g_comm_mu);
  g_comm[cslot(h)] = DeviceComm{c, world_size, rank};
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_comm_init_all(fastmc_t* const* handles, int n) {
  if (!handles || n < 1 || n > 64) return fail(FASTMC_EINVAL, "bad argument");
  std::vector<int> devs(n);
  bool shared = false;
  for (int i = 0; i < n; ++i) {
    if (!handles[i]) return fail(FASTMC_EINVAL, "null handle");
    devs[i] = handles[i]->device;
    for (int j = 0; j < i; ++j) shared = shared || devs[j] == devs[i];
  }
  if (shared) {
    // RCCL needs one device per rank.  Only the stand-in library of the tests takes several ranks on one device: their handles
    // then get slots of their own in the communicator tables (see cslot)
    if (!virtual_ranks_allowed() || n > 32)
      return fail(FASTMC_ECOMM, "two handles on one device: RCCL needs one device per rank (use the host exchange)");
  }
  // The private slots of the virtual ranks hold for this call and are KEPT only when the clique is up: on every failure below the
  // handles go back to their device's slot (a handle left on slot 32 + i addressed the wrong communicator afterwards)
  std::vector<int> old_slot(n);
  for (int i = 0; i < n; ++i) old_slot[i] = handles[i]->comm_slot;
  struct Rollback {
    fastmc_t* const* hs; const std::vector<int>& old; int n; bool keep = false;
    ~Rollback() { if (!keep) for (int i = 0; i < n; ++i) hs[i]->comm_slot = old[i]; }
  } rollback{handles, old_slot, n};
  if (shared) for (int i = 0; i < n; ++i) handles[i]->comm_slot = 32 + i;
  for (int i = 0; i < n; ++i)
    if (device_comm(cslot(handles[i])).comm) return fail(FASTMC_ESTATE, "a device already has a communicator (fastmc_comm_destroy first)");
  TRY(load_rccl());
  std::vector<ncclComm_t> comms(n, nullptr);
  std::vector<int> gen(n);
  for (int i = 0; i < n; ++i) gen[i] = g_abort_gen[cslot(handles[i])].load();
  NCCLCHK(g_rccl.CommInitAll(comms.data(), n, devs.data()));
  for (int i = 0; i < n; ++i)
    if (g_abort_gen[cslot(handles[i])].load() != gen[i]) {       // aborted meanwhile (see fastmc_comm_init)
      for (int j = 0; j < n; ++j) g_rccl.CommAbort(comms[j]);
      return fail(FASTMC_ECOMM, "communicators aborted while they were being initialised");
    }
  std::lock_guard<std::mutex> g(g_comm_mu);
  for (int i = 0; i < n; ++i) g_comm[cslot(handles[i])] = DeviceComm{comms[i], n, i};
  rollback.keep = true;
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_comm_world(fastmc_t* h, int* world_size, int* rank) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  const DeviceComm dc = device_comm(cslot(h));
  if (world_size) *world_size = dc.comm ? dc.world : 0;
  if (rank) *rank = dc.comm ? dc.rank : -1;
  return 0;
}
#endif

// Enqueue this handle's part of the exchange on its stream (inside an RCCL group when several handles of one
// process take part); the host copies follow on the same stream.
static int comm_enqueue_gather(fastmc_ctx* h, const DeviceComm& dc, int64_t n_local, bool powers) {
  HIPCHK(hipSetDevice(h->device));
  guard_outputs(h);
  if (powers) NCCLCHK(g_rccl.AllGather(h->out, h->gather_buf, (size_t)n_local, ncclDouble, dc.comm, h->stream));
  return 0;
}
static int comm_enqueue_hist(fastmc_ctx* h, const DeviceComm& dc, int nbins) {
  HIPCHK(hipSetDevice(h->device));
  guard_outputs(h);
  NCCLCHK(g_rccl.AllReduce(h->hist, h->hist, (size_t)nbins + 2, ncclUint64, ncclSum, dc.comm, h->stream));
  return 0;
}

#if FMC_TU == 0
extern "C" int fastmc_comm_gather(fastmc_t* h, int64_t n_local, double* all_powers, int64_t* hist, double lo_db,
                                  double hi_db, int nbins) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  if (stall_requested()) return stall_until_abort({cslot(h)});
  FMC_LOCK(h);
  int world = 0;
  const int gen = g_abort_gen[cslot(h)].load();
  {
    std::lock_guard<std::timed_mutex> enq(g_enq_mu[cslot(h)]);
    const DeviceComm dc = device_comm(cslot(h));
    if (!dc.comm) return fail(FASTMC_ESTATE, "fastmc_comm_init not called for this device (or its communicator was aborted)");
    if (n_local <= 0 || n_local > h->last_n_iter * (h->last_coherent ? 2 : 1)) return fail(FASTMC_EINVAL, "n_local exceeds the last run");
    world = dc.world;
    HIPCHK(hipSetDevice(h->device));
    if (all_powers) TRY(grow(&h->gather_buf, &h->gather_cap, (size_t)n_local * dc.world));
    if (hist) TRY(histogram_device(h, lo_db, hi_db, nbins));       // local histogram kernel: before the timed exchange
    TRY(exchange_begin(h));
    if (all_powers) TRY(comm_enqueue_gather(h, dc, n_local, true));
    if (hist) TRY(comm_enqueue_hist(h, dc, nbins));
    TRY(exchange_end(h));
  }
  if (stall_mode() == 2) {        // fault injection: the collectives are on the stream, the exchange "never completes"
    while (g_abort_gen[cslot(h)].load() == gen) usleep(500);
    hipStreamSynchronize(h->stream);
    finish_pending(h);
    return fail(FASTMC_ECOMM, "exchange aborted after its collectives were enqueued (FASTMC_TEST_STALL_GATHER=2)");
  }
  if (all_powers) HIPCHK(hipMemcpyAsync(all_powers, h->gather_buf, (size_t)n_local * world * 8, hipMemcpyDeviceToHost, h->stream));
  if (hist) HIPCHK(hipMemcpyAsync(hist, h->hist, ((size_t)nbins + 2) * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  finish_pending(h);
  if (g_abort_gen[cslot(h)].load() != gen) return fail(FASTMC_ECOMM, "the communicator was aborted during the exchange");
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_comm_gather_all(fastmc_t* const* handles, int n, int64_t n_local, double* all_powers, int64_t* hist,
                                      double lo_db, double hi_db, int nbins) {
  if (!handles || n < 1 || n > 64) return fail(FASTMC_EINVAL, "bad argument");
  if (stall_requested()) {
    std::vector<int> devs;
    for (int i = 0; i < n; ++i) if (handles[i]) devs.push_back(cslot(handles[i]));
    return stall_until_abort(devs);
  }
  for (int i = 0; i < n; ++i) if (!handles[i]) return fail(FASTMC_EINVAL, "null handle");
  // every handle for the length of the call (in the caller's order: the group always passes the same order), every
  // device's communicator for the length of the enqueue
  std::vector<std::unique_ptr<HandleLock>> hl;
  for (int i = 0; i < n; ++i) {
    hl.emplace_back(new HandleLock(handles[i]));
    if (!hl.back()->ok) return fail(FASTMC_ESTATE, "a handle is in use by another thread (an exchange that missed its deadline has not returned)");
  }
  std::vector<int> gens(n);
  for (int i = 0; i < n; ++i) gens[i] = g_abort_gen[cslot(handles[i])].load();
  std::vector<std::unique_lock<std::timed_mutex>> enq;
  for (int i = 0; i < n; ++i) enq.emplace_back(g_enq_mu[cslot(handles[i])]);
  std::vector<DeviceComm> dcs(n);
  for (int i = 0; i < n; ++i) {
    fastmc_ctx* h = handles[i];
    dcs[i] = device_comm(cslot(h));
    if (!dcs[i].comm || dcs[i].world != n || dcs[i].rank != i)
      return fail(FASTMC_ESTATE, "handles do not match the communicators of fastmc_comm_init_all (same handles, same order)");
    if (n_local <= 0 || n_local > h->last_n_iter * (h->last_coherent ? 2 : 1)) return fail(FASTMC_EINVAL, "n_local exceeds a handle's last run");
  }
  for (int i = 0; i < n; ++i) {
    HIPCHK(hipSetDevice(handles[i]->device));
    if (all_powers) TRY(grow(&handles[i]->gather_buf, &handles[i]->gather_cap, (size_t)n_local * n));
    if (hist) TRY(histogram_device(handles[i], lo_db, hi_db, nbins));     // local histogram kernels: before the timed exchange
    TRY(exchange_begin(handles[i]));
  }
  if (all_powers) {
    NCCLCHK(g_rccl.GroupStart());
    int rc = 0;
    for (int i = 0; i < n && rc == 0; ++i) rc = comm_enqueue_gather(handles[i], dcs[i], n_local, true);
    ncclResult_t ge = g_rccl.GroupEnd();
    if (rc) return rc;
    if (ge != ncclSuccess) return fail(FASTMC_ECOMM, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge));
  }
  if (hist) {
    NCCLCHK(g_rccl.GroupStart());
    int rc = 0;
    for (int i = 0; i < n && rc == 0; ++i) rc = comm_enqueue_hist(handles[i], dcs[i], nbins);
    ncclResult_t ge = g_rccl.GroupEnd();
    if (rc) return rc;
    if (ge != ncclSuccess) return fail(FASTMC_ECOMM, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge));
  }
  for (int i = 0; i < n; ++i) {
    HIPCHK(hipSetDevice(handles[i]->device));
    TRY(exchange_end(handles[i]));
  }
  enq.clear();          // the communicators are RCCL's from here on: an abort may free them while we wait below
  if (stall_mode() == 2) {
    for (;;) {
      bool aborted = false;
      for (int i = 0; i < n; ++i) aborted = aborted || g_abort_gen[cslot(handles[i])].load() != gens[i];
      if (aborted) break;
      usleep(500);
    }
    for (int i = 0; i < n; ++i) { hipSetDevice(handles[i]->device); hipStreamSynchronize(handles[i]->stream); finish_pending(handles[i]); }
    return fail(FASTMC_ECOMM, "exchange aborted after its collectives were enqueued (FASTMC_TEST_STALL_GATHER=2)");
  }
  // every rank holds the same gathered data: the host copy comes from rank 0, the others are only waited for
  fastmc_ctx* h0 = handles[0];
  HIPCHK(hipSetDevice(h0->device));
  if (all_powers) HIPCHK(hipMemcpyAsync(all_powers, h0->gather_buf, (size_t)n_local * n * 8, hipMemcpyDeviceToHost, h0->stream));
  if (hist) HIPCHK(hipMemcpyAsync(hist, h0->hist, ((size_t)nbins + 2) * 8, hipMemcpyDeviceToHost, h0->stream));
  for (int i = 0; i < n; ++i) {
    HIPCHK(hipSetDevice(handles[i]->device));
    HIPCHK(hipStreamSynchronize(handles[i]->stream));
    finish_pending(handles[i]);
  }
  for (int i = 0; i < n; ++i)
    if (g_abort_gen[cslot(handles[i])].load() != gens[i]) return fail(FASTMC_ECOMM, "a communicator was aborted during the exchange");
  return 0;
}
#endif

#if FMC_TU == 0
// The exchange of a QUEUED step (fastmc_run_queued on the same slot first): the collectives and the copies of their results
// into the slot's pinned buffers are enqueued behind the step's kernels and the call returns; fastmc_queue_wait collects.
extern "C" int fastmc_comm_gather_queued(fastmc_t* h, int64_t n_local, int want_powers, double lo_db, double hi_db, int nbins, int slot) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  if (slot < 0 || slot > 1) return fail(FASTMC_EINVAL, "slot must be 0 or 1");
  FMC_LOCK(h);
  QueueSlot& q = h->q[slot];
  if (!q.busy) return fail(FASTMC_ESTATE, "fastmc_run_queued on this slot first");
  if (stall_requested()) { q.stalled = true; q.stall_gen = g_abort_gen[cslot(h)].load(); return 0; }   // fault injection: enqueueing never blocks; the wait will
  std::lock_guard<std::timed_mutex> enq(g_enq_mu[cslot(h)]);
  const DeviceComm dc = device_comm(cslot(h));
  if (!dc.comm) return fail(FASTMC_ESTATE, "fastmc_comm_init not called for this device (or its communicator was aborted)");
  if (n_local <= 0 || n_local > h->last_n_iter * (h->last_coherent ? 2 : 1)) return fail(FASTMC_EINVAL, "n_local exceeds the last run");
  HIPCHK(hipSetDevice(h->device));
  if (want_powers) TRY(grow(&h->gather_buf, &h->gather_cap, (size_t)n_local * dc.world));
  {
    SlotScope sc(h, q);
    if (nbins > 0) TRY(histogram_device(h, lo_db, hi_db, nbins));
    TRY(exchange_begin(h));
    if (want_powers) TRY(comm_enqueue_gather(h, dc, n_local, true));
    if (nbins > 0) TRY(comm_enqueue_hist(h, dc, nbins));
    TRY(exchange_end(h));
  }
  if (want_powers) TRY(slot_land(h, q, h->gather_buf, (size_t)n_local * dc.world));
  if (nbins > 0) TRY(slot_land_hist(h, q, nbins));
  return slot_mark_done(h, q);
}

extern "C" int fastmc_comm_gather_all_queued(fastmc_t* const* handles, int n, int64_t n_local, int want_powers, double lo_db, double hi_db,
                                             int nbins, int slot) {
  if (!handles || n < 1 || n > 64) return fail(FASTMC_EINVAL, "bad argument");
  if (slot < 0 || slot > 1) return fail(FASTMC_EINVAL, "slot must be 0 or 1");
  for (int i = 0; i < n; ++i) if (!handles[i]) return fail(FASTMC_EINVAL, "null handle");
  std::vector<std::unique_ptr<HandleLock>> hl;
  for (int i = 0; i < n; ++i) {
    hl.emplace_back(new HandleLock(handles[i]));
    if (!hl.back()->ok) return fail(FASTMC_ESTATE, "a handle is in use by another thread (an exchange that missed its deadline has not returned)");
    if (!handles[i]->q[slot].busy) return fail(FASTMC_ESTATE, "fastmc_run_queued on this slot of every handle first");
  }
  if (stall_requested()) {       // fault injection: enqueueing never blocks; the waits will
    for (int i = 0; i < n; ++i) { handles[i]->q[slot].stalled = true; handles[i]->q[slot].stall_gen = g_abort_gen[cslot(handles[i])].load(); }
    return 0;
  }
  std::vector<std::unique_lock<std::timed_mutex>> enq;
  for (int i = 0; i < n; ++i) enq.emplace_back(g_enq_mu[cslot(handles[i])]);
  std::vector<DeviceComm> dcs(n);
  for (int i = 0; i < n; ++i) {
    fastmc_ctx* h = handles[i];
    dcs[i] = device_comm(cslot(h));
    if (!dcs[i].comm || dcs[i].world != n || dcs[i].rank != i)
      return fail(FASTMC_ESTATE, "handles do not match the communicators of fastmc_comm_init_all (same handles, same order)");
    if (n_local <= 0 || n_local > h->last_n_iter * (h->last_coherent ? 2 : 1)) return fail(FASTMC_EINVAL, "n_local exceeds a handle's last run");
  }
  std::vector<std::unique_ptr<SlotScope>> scopes;
  for (int i = 0; i < n; ++i) scopes.emplace_back(new SlotScope(handles[i], handles[i]->q[slot]));
  for (int i = 0; i < n; ++i) {
    HIPCHK(hipSetDevice(handles[i]->device));
    if (want_powers) TRY(grow(&handles[i]->gather_buf, &handles[i]->gather_cap, (size_t)n_local * n));
    if (nbins > 0) TRY(histogram_device(handles[i], lo_db, hi_db, nbins));
    TRY(exchange_begin(handles[i]));
  }
  if (want_powers) {
    NCCLCHK(g_rccl.GroupStart());
    int rc = 0;
    for (int i = 0; i < n && rc == 0; ++i) rc = comm_enqueue_gather(handles[i], dcs[i], n_local, true);
    ncclResult_t ge = g_rccl.GroupEnd();
    if (rc) return rc;
    if (ge != ncclSuccess) return fail(FASTMC_ECOMM, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge));
  }
  if (nbins > 0) {
    NCCLCHK(g_rccl.GroupStart());
    int rc = 0;
    for (int i = 0; i < n && rc == 0; ++i) rc = comm_enqueue_hist(handles[i], dcs[i], nbins);
    ncclResult_t ge = g_rccl.GroupEnd();
    if (rc) return rc;
    if (ge != ncclSuccess) return fail(FASTMC_ECOMM, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge));
  }
  for (int i = 0; i < n; ++i) {
    HIPCHK(hipSetDevice(handles[i]->device));
    TRY(exchange_end(handles[i]));
  }
  scopes.clear();
  // every rank holds the same gathered data: it lands on the host from rank 0; the others only mark their step done
  for (int i = 0; i < n; ++i) {
    fastmc_ctx* h = handles[i];
    QueueSlot& q = h->q[slot];
    HIPCHK(hipSetDevice(h->device));
    q.landed = q.hist_landed = 0;
    if (i == 0 && want_powers) TRY(slot_land(h, q, h->gather_buf, (size_t)n_local * n));
    if (i == 0 && nbins > 0) TRY(slot_land_hist(h, q, nbins));
    TRY(slot_mark_done(h, q));
  }
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_comm_abort(fastmc_t* h) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  ncclComm_t c = nullptr;
  // not while another thread is handing this communicator to RCCL (see g_enq_mu); never longer than two seconds
  std::unique_lock<std::timed_mutex> enq(g_enq_mu[cslot(h)], std::defer_lock);
  (void)enq.try_lock_for(std::chrono::seconds(2));
  {
    std::lock_guard<std::mutex> g(g_comm_mu);
    c = g_comm[cslot(h)].comm;
    g_comm[cslot(h)] = DeviceComm();
  }
  g_abort_gen[cslot(h)].fetch_add(1);
  if (c && g_rccl.lib) {
    hipSetDevice(h->device);
    ncclResult_t r = g_rccl.CommAbort(c);
    if (r != ncclSuccess) return fail(FASTMC_ECOMM, std::string("ncclCommAbort: ") + g_rccl.GetErrorString(r));
  }
  return 0;
}

extern "C" int fastmc_last_exchange_ms(fastmc_t* h, double* ms) {
  if (!h || !ms) return fail(FASTMC_EINVAL, "null argument");
  *ms = h->ex_ms;
  return 0;
}
#endif

#if FMC_TU == 0
extern "C" int fastmc_comm_destroy(fastmc_t* h) {
  if (!h) return fail(FASTMC_EINVAL, "null handle");
  ncclComm_t c = nullptr;
  std::unique_lock<std::timed_mutex> enq(g_enq_mu[cslot(h)], std::defer_lock);
  (void)enq.try_lock_for(std::chrono::seconds(2));
  {
    std::lock_guard<std::mutex> g(g_comm_mu);
    c = g_comm[cslot(h)].comm;
    g_comm[cslot(h)] = DeviceComm();
  }
  if (c && g_rccl.lib) {
    hipSetDevice(h->device);
    g_rccl.CommDestroy(c);
  }
  return 0;
}
#endif
#endif   // FMC_TU == 0
