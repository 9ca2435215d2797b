/* fastmc.h -- C-ABI of libfastmc.so: the MI355X (gfx950) implementation of the FAST
 * Monte-Carlo hot path.  Plain C types only; loaded with ctypes / cgo / JNI / dlopen.
 *
 * The reference (ojdf/fast, /root/reference) is pure Python and has no FFI.  Each entry
 * point below names the reference code it replaces (file:line, relative to the
 * reference root); the Python binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success or a negative FASTMC_E* code; the message
 *     for the calling thread is returned by fastmc_last_error();
 *   - the caller owns every host buffer; the library copies in during set_* / run and
 *     owns all device memory until fastmc_destroy();
 *   - all host arrays are C-contiguous float64 unless stated otherwise;
 *   - calls on one handle should be serialised by the caller; calls are blocking; different
 *     handles may be driven from different threads (each has its own HIP stream).  The entry points that use a handle's
 *     stream and result bookkeeping (run*, wait, set_results, histogram, result_stats, comm_gather*) take a per-handle lock
 *     with a deadline (FASTMC_HANDLE_BUSY_TIMEOUT seconds, default 30): a second caller waits its turn and then fails with
 *     FASTMC_ESTATE instead of racing the first -- the case that matters is an exchange a deadline thread is still inside
 *     while its caller, having called fastmc_comm_abort (which takes no handle lock), goes on with the handle;
 *   - N <= 8192: every N <= 4096 has a kernel family; beyond it the grids N = 64 P S / 50 P S of S <= 8 sub-rows (7 <= P <= 24:
 *     4608, 5000, 5120, 6000, 6144, 7000, 7168, 8000, 8192 ...) for any window, every other N for windows of up to 256 pixels
 *     (chirp-z kernels, rows in input blocks); Np <= N; fastmc_destroy() parks ONE retired handle per device, whole (stream, buffers), and
 *     fastmc_create() of the same (N, Np, precision) on that device takes it back, reset to the state of a new
 *     handle (sweeps build one short-lived handle per geometry sample); a handle it displaces is freed, except
 *     for the largest work buffer of the device, which is kept for the next handle;
 *   - nothing here ever falls back to a CPU implementation: without a gfx950 device
 *     fastmc_create() fails with FASTMC_ENODEV;
 *   - environment: FASTMC_TEST_STALL_GATHER=1 makes fastmc_comm_gather / _gather_all block (without touching RCCL)
 *     until fastmc_comm_abort is called on the handle's device, then fail with FASTMC_ECOMM: the fault the deadline
 *     tests inject (tests/test_gpu_dist.py); =2 blocks the same way AFTER the collectives of a real communicator have been
 *     enqueued, holding the handle (the abort then meets a thread with RCCL work on its stream);
 *   - environment: FASTMC_DISABLE_RCCL=1 makes the communicator entry points fail with FASTMC_ECOMM (callers exchange
 *     through the host); FASTMC_RCCL_LIB=<path> names the library to load in place of librccl.so.1 (an RCCL build off the
 *     loader's path, or the tests' stand-in tests/stubs/fake_rccl.cpp -- with which, and only with which,
 *     FASTMC_TEST_VIRTUAL_RANKS=1 lets fastmc_comm_init_all take several handles of ONE device as ranks of a clique, so that
 *     the grouped collectives run on a one-GPU box: tests/test_fake_rccl.py); FASTMC_NO_DENSE16=1 (read by fastmc_create) keeps the twelve-wave kernels where the
 *     sixteen-wave dense-image kernels would run (A/B timing; same results); FASTMC_ROWS_PERSIST=0 (read at the first row launch)
 *     gives every tile of a large row launch a workgroup of its own instead of letting the resident workgroups walk the tiles
 *     (A/B timing; same results); FASTMC_COLS_PERSIST=0: the same for the column launches of the 1024-point pipeline.
 *   - environment, kernel choice for A/B timing (read once; same results to rounding): FASTMC_PKS=0 keeps the packed sub-rows off
 *     (staged draws on the grid's one-row-per-wave / chirp-z / 50-lane rows instead); FASTMC_PKS8=0 keeps them to windows of 96 pixels, FASTMC_PKS16=0 to 128; FASTMC_PKS_P16=1024 lets them serve 1024 too
 *     (default: from 2048); FASTMC_PBZ=0 runs the chirp-z grids one wavefront per row as rounds 1-5 did, FASTMC_PBZ_COLS=0 only their
 *     column pass; FASTMC_BLU_P=0 keeps that form to its five sizes of rounds 1-5; FASTMC_GEN64_STAGED=1 stages the float64 generator
 *     through memory everywhere.
 */
#ifndef FASTMC_H
#define FASTMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FASTMC_VERSION 300

#define FASTMC_OK 0
#define FASTMC_EINVAL (-1)   /* bad argument */
#define FASTMC_ENODEV (-2)   /* no usable GPU */
#define FASTMC_EHIP (-3)     /* HIP runtime error (message has the HIP string) */
#define FASTMC_ESTATE (-4)   /* spectrum / pupil not set */
#define FASTMC_ECOMM (-5)    /* RCCL error */

#define FASTMC_F64 0 /* complex128 pipeline: reference precision */
#define FASTMC_F32 1 /* complex64 pipeline, float64 detector sums */

typedef struct fastmc_ctx fastmc_t;

int fastmc_version(void);
const char* fastmc_last_error(void);
int fastmc_device_count(int* n);

/* One handle = one GPU + one (N, Np) problem.
 * Replaces: Fast.init_fftw (fast/fast.py:419-438: plan + IN/OUT buffers) and
 * Fast.init_phs_logamp (fast/fast.py:440-443).  N = Npxls, Np = Npxls_pup. */
int fastmc_create(fastmc_t** h, int device_id, int N, int Np, int precision);
void fastmc_destroy(fastmc_t* h);

/* powerspec: (N, N) residual phase PSD in the reference's fft-shifted layout
 * (`Fast.powerspec`, fast/fast.py:481); df = freq.main.df.  The library stores
 * sqrt(powerspec)*df, i.e. the colouring of fast/fast.py:594 and the `rand * df`
 * of fast/funcs.py:213. */
int fastmc_set_spectrum(fastmc_t* h, const double* powerspec, double df);

/* W = pupil * pupil_mode on the (Np, Np) window (fast/fast.py:649), crop_lo = first
 * row/column of the window = pup_coords[0][0] (fast/fast.py:390), dx = pixel scale. */
int fastmc_set_pupil(fastmc_t* h, const double* W, int crop_lo, double dx);

/* Optional sub-harmonics (fast/funcs.py:225-258, fast/fast.py:598-603).
 * powerspec_sh, fx, fy: (3, 3, 3) [level][row][col]; df: (3,).  NULL disables. */
int fastmc_set_subharm(fastmc_t* h, const double* powerspec_sh, const double* fx,
                       const double* fy, const double* df);

/* Monte-Carlo run with the on-device generator (xoshiro128+ streams seeded by Philox4x32-7
 * blocks keyed on (seed; realisation, row, stream = column mod SL), SL = 64 -- 128 / 256 at 2048 / 4096, 50 S on the
 * 50 P S grids --, two words per state advance, Box-Muller; the log-amplitude and sub-harmonic draws take
 * Philox4x32-10 blocks directly; restated in oracle/devrng.py) -- replaces the body of the chunk loop of
 * Fast.run (fast/fast.py:130-134: compute_phs 589-605 + compute_detector 647-668) and
 * Fast.compute_logamp (fast/fast.py:639-645).
 *
 * Realisation g (one complex N x N FFT) yields two iterations: its real and its
 * imaginary screen (fast/funcs.py:220-221).  Realisations real0 .. real0+n_real-1 are
 * computed; results do not depend on how a range is split over calls or GPUs.
 *   logamp: 2*n_real log-amplitudes chi, ordered like `out`, or NULL to draw them on
 *           the device as N(0, logamp_var) (stream keyed on the global iteration 2g+s).
 *   out:    coherent == 0: 2*n_real float64, out[j] = |a|^2 of the REAL screen of
 *           realisation real0+j, out[n_real+j] = of its IMAGINARY screen (the order of
 *           vstack([Re, Im]) in fast/funcs.py:221);
 *           coherent != 0: 2*n_real complex128 (interleaved re, im), same order. */
int fastmc_run(fastmc_t* h, uint64_t seed, int64_t real0, int64_t n_real,
               const double* logamp, double logamp_var, int coherent, double* out);

/* The same run without the host copy and without waiting: the kernels are enqueued on the handle's stream and the
 * results stay on the device (log-amplitudes drawn on the device).  What follows on the handle is ordered behind them:
 * fastmc_comm_gather / fastmc_comm_gather_all put the exchange on the same stream, so that a sharded step synchronises
 * ONCE, after its all-gather; fastmc_histogram / fastmc_result_stats / fastmc_link_metrics likewise.
 * fastmc_wait(h, out) waits for the stream and copies the results of the last run (2*n_real float64, or complex128
 * when coherent; out may be NULL: wait only).  It also serves after a blocking fastmc_run (fetches the vector again). */
int fastmc_run_async(fastmc_t* h, uint64_t seed, int64_t real0, int64_t n_real, double logamp_var, int coherent);
int fastmc_wait(fastmc_t* h, double* out);

/* Parity mode: the same pipeline fed with host-drawn coefficients (numpy draw order of
 * fast/funcs.py:352-356: all real parts, then all imaginary parts).
 *   coeff_re, coeff_im: (n_real, N, N) standard normals;
 *   sh_re, sh_im: (n_real, 3, 3, 3) sub-harmonic coefficients or NULL;
 *   logamp: 2*n_real values (required: the host drew them, fast/fast.py:123). */
int fastmc_run_coeffs(fastmc_t* h, const double* coeff_re, const double* coeff_im,
                      int64_t n_real, const double* sh_re, const double* sh_im,
                      const double* logamp, int coherent, double* out);

/* Debug / parity: the cropped phase screens themselves, (2*n_real, Np, Np) float64 in
 * the order of `out` above == Fast.phs after compute_phs (fast/fast.py:596-603). */
int fastmc_screens_coeffs(fastmc_t* h, const double* coeff_re, const double* coeff_im,
                          int64_t n_real, const double* sh_re, const double* sh_im,
                          double* phs);
int fastmc_screens(fastmc_t* h, uint64_t seed, int64_t real0, int64_t n_real, double* phs);

/* Debug / parity: the device generator's coefficients for one realisation,
 * (N, N) complex128 interleaved; and device log-amplitude normals. */
int fastmc_rng_coeffs(fastmc_t* h, uint64_t seed, int64_t real, double* coeff_interleaved);
int fastmc_rng_logamp(fastmc_t* h, uint64_t seed, int64_t iter0, int64_t n_iter, double* normals);

/* Frozen-flow time series (TEMPORAL mode; Fast.compute_phs_temporal fast/fast.py:607-637 +
 * compute_detector 647-668).  set_layer_screens keeps the L real N x N layer screens on the
 * device (the reference builds them once, in chunk 0, fast.py:609-615).  temporal_chunk evaluates
 * one chunk of M time steps: xs, ys are the wrapped, sorted sample coordinates (L, M, Np)
 * (fast.py:621-622), roll the numpy.roll amounts (L, 2, M) (fast.py:624-626), logamp the M
 * log-amplitudes of the chunk; out: M float64 (or M complex128 when coherent). */
int fastmc_set_layer_screens(fastmc_t* h, const double* screens, int n_layers);
int fastmc_temporal_chunk(fastmc_t* h, const double* xs, const double* ys, const int32_t* roll, int M,
                          const double* logamp, int coherent, double* out);
/* The phases of the same chunk, (M, Np, Np): the wind-shifted, bilinearly sampled layer screens summed over the layers --
 * what the reference leaves in Fast.phs after a TEMPORAL chunk (fast/fast.py:619-633). */
int fastmc_temporal_phases(fastmc_t* h, const double* xs, const double* ys, const int32_t* roll, int M, double* phs);

/* Make a result vector the handle's resident results (n_iter float64 powers, or n_iter complex128
 * amplitudes when coherent): what fastmc_histogram / fastmc_result_stats / fastmc_link_metrics /
 * fastmc_comm_gather then reduce.  fastmc_run leaves its results resident by itself; a run made of
 * several calls (host-coefficient chunks, fastmc_temporal_chunk, a sharded run after its exchange)
 * assembles FastResult._r on the host (fast/fast.py:127-137) and hands it back with this call. */
int fastmc_set_results(fastmc_t* h, const double* values, int64_t n_iter, int coherent);

/* Fixed-bin histogram of 10*log10(power) of the last run's results kept on the device
 * (FastResult.dB_rel, fast/fast.py:949-951): bins[k] counts lo + k*(hi-lo)/nbins <= x <
 * ..., bins[nbins] = underflow, bins[nbins+1] = overflow.  bins: nbins+2 int64. */
int fastmc_histogram(fastmc_t* h, double lo_db, double hi_db, int nbins, int64_t* bins);

/* Statistics of the last run's results, reduced on the device (FastResult.avg_power_*,
 * scintillation_index, fast/fast.py:965-983; comms.fade_prob, fast/comms.py:171-177):
 * stats = [n, sum r, sum r^2, sum 10 log10 r, min r, max r, count(r < thr[0]), ...], r = power
 * relative to the diffraction limit; thresholds in the same units; n_thr <= 16. */
int fastmc_result_stats(fastmc_t* h, const double* thresholds, int n_thr, double* stats);

/* Link metrics over a power vector, reduced on the device (fast/comms.py:171-262: fade_prob 171-177,
 * fade_dur 180-195, ber_ook 198-222, sep_qam 225-242, ber_qam 245-255, Q 258-262).
 * The vector is `samples` (n host doubles, copied to device `device_id`; h may be NULL), or, when
 * samples == NULL, the last run's results resident on h's device (|a|^2 for a coherent run; n ignored).
 * out: 4 doubles per query --
 *   FASTMC_LM_FADE   p0 = threshold (units of the vector):
 *       [count(x < p0), rising edges (x[i] < p0 <= x[i-1], i >= 1), index of the first sample >= p0 (n if
 *        none), index of the last sample >= p0 (-1 if none)]; the caller forms fade_prob and the mean
 *        duration of the fades that start and end inside the record from these (fast_amd/comms.py).
 *   FASTMC_LM_BER_OOK p0 = Eb/N0 [dB]:        [sum_i Q(s_i * sqrt(10^(p0/10))), mean(x), n, 0], s = x / mean(x)
 *   FASTMC_LM_SEP_QAM p0 = M, p1 = Es/N0 [dB]: [sum_i 4 (c q_i - c^2 q_i^2), mean(x), n, 0],
 *       q_i = Q(sqrt(3/(M-1) 10^(p1/10) s_i^2)), c = (sqrt(M)-1)/sqrt(M). */
#define FASTMC_LM_FADE 0
#define FASTMC_LM_BER_OOK 1
#define FASTMC_LM_SEP_QAM 2
typedef struct {
  int32_t kind;
  double p0, p1;
} fastmc_link_query;
int fastmc_link_metrics(fastmc_t* h, int device_id, const double* samples, int64_t n,
                        const fastmc_link_query* queries, int n_queries, double* out);

/* Timing of the last fastmc_run / fastmc_run_coeffs, measured with HIP events on the
 * library's own stream: total ms, and per kernel family [rows, cols, finalize] ms and
 * launch counts.  times_ms: 4 doubles, launches: 4 int64. */
int fastmc_last_timing(fastmc_t* h, double* times_ms, int64_t* launches);

/* Which kernel family the handle uses:
 *   1 = wave-FFT (N = 64 P with P = 2^k times 1, 3, 5, 7 or 9, P <= 32: 128, 192, 256, 320, 384, 448, 512, 576, 640, 768,
 *       896, 1024, 1152, 1280, 1536, 1792, 2048 / 4096 as interleaved sub-rows of 1024, and 1344, 1728, 1920, 2304, 2560,
 *       2688, 3072, 3456, 3584, 3840 as 2 ... 4 interleaved sub-rows of 448 ... 1536).  128, 256, 512 run as packed rows (8 / 4 / 2
 *       rows per wavefront).
 *       Whatever family a grid belongs to (1, 2 or 3), the device generator's rows of EVERY multiple of 64 from 192 to 4096 except
 *       256, 512 and 1024, with a centred window of up to 128 pixels (six planes of a sub-transform up to 96, eight beyond; up to 256
 *       pixels with all sixteen planes where 256 divides N), run as
 *       packed SUB-ROWS, rows and columns (S interleaved sub-rows
 *       of 256 / 128 / 64 points, S = N / 256, else N / 128, else N / 64; 4 / 8 / 8 rows per wavefront; round 6) -- the generator draws
 *       N / 16 streams per row on these grids (N / 8 on the odd multiples of 64), as it always did on 2048 and 4096; the family
 *       reported here transforms their host coefficients and any other window;
 *   3 = 50-lane FFT (round decimal grids N = 50 P S, P as above and <= 24, S <= 5 interleaved sub-rows: 100, 150, ..., 500,
 *       600, ..., 1000, 1200, 1350, 1400, 1500, 1600, 1750, 1800, 2000, 2100, 2250, 2400, 2500, 2700, 2800, 3000, 3200,
 *       3500, 3600, 4000; Np <= 128, or <= 256 for P = 8, 10, 12, 16, 20, 24).  The device generator draws 50 S streams per
 *       row on these grids, whichever family transforms them;
 *   2 = chirp-z (any other N, odd included, Np <= 256: every 1-D transform as a Bluestein convolution; default for N >= 96).
 *       Windows of up to 128 pixels (round 6): rows and columns in blocks of 128 inputs on the packed 256-point pipeline, four per
 *       wavefront, one pruned inverse transform each; wider windows: one wavefront per row / column with 64 P >= N + Np - 1 points,
 *       P in {4, 8, 12, 16, 24, 28, 32}, rows beyond 2048 points in input blocks;
 *   0 = direct O(N^2 Np) pruned DFT (any N the LDS holds; tiny grids, huge windows, cross-check of the other three).
 * force: -1 query only, 0 / 1 / 2 / 3 select (fails with EINVAL if the family does not serve this (N, Np)). */
int fastmc_kernel_path(fastmc_t* h, int force);

/* ---- Two steps in flight per handle (round 4) ------------------------------------------------------------------------
 * fastmc_run / fastmc_run_async + fastmc_wait leave the device idle while the host collects a step's results and issues
 * the next one.  A handle has two SLOTS (0 and 1), each with its own timing events, pinned host landing buffers and a
 * completion event, so that a caller can keep the stream fed:
 *     run_queued(step 0, slot 0);  for i = 0, 1, ...: { run_queued(step i + 1, slot (i + 1) & 1);  queue_wait(slot i & 1) }
 * The kernels and the exchange of step i, then the kernels of step i + 1 ..., are ordered by the handle's compute stream.  The
 * copies that land a step's results on its slot run on a second, COPY stream behind an event of the compute stream, so that the
 * next step's kernels do not queue up behind them; the device buffers they read (the result vector, the histogram, the gather
 * buffer) are single, so every later writer of those buffers first waits for the completion event of the latest landing copies
 * (fastmc.hip: copy_guard / guard_outputs -- the invariant a new writer of `out`, `hist` or `gather_buf` must keep).
 *   fastmc_run_queued          as fastmc_run_async; fetch != 0 also lands the step's own result vector on the slot;
 *   fastmc_comm_gather_queued  after run_queued on the same slot: enqueue the all-gather of the n_local values per rank
 *                              (want_powers != 0) and / or the all-reduced dB histogram (nbins > 0) and land them on the slot;
 *   fastmc_comm_gather_all_queued  the same for the N handles of one process (ncclCommInitAll order); rank 0's slot lands;
 *   fastmc_queue_wait          wait for the slot's step; copy what it landed into out (capacity out_cap doubles) and hist
 *                              (hist_cap int64: nbins + 2); either may be NULL.  Returns the number of doubles landed.
 *                              fastmc_last_timing / fastmc_last_exchange_ms then describe that step.
 * Errors as everywhere: a busy slot, a missing run_queued, an aborted communicator are FASTMC_ESTATE / FASTMC_ECOMM. */
int fastmc_run_queued(fastmc_t* h, uint64_t seed, int64_t real0, int64_t n_real, double logamp_var, int coherent, int slot,
                      int fetch);
int fastmc_comm_gather_queued(fastmc_t* h, int64_t n_local, int want_powers, double lo_db, double hi_db, int nbins, int slot);
int fastmc_comm_gather_all_queued(fastmc_t* const* handles, int n, int64_t n_local, int want_powers, double lo_db, double hi_db,
                                  int nbins, int slot);
int fastmc_histogram_queued(fastmc_t* h, double lo_db, double hi_db, int nbins, int slot);   /* of the slot's own results, no exchange */
int fastmc_queue_wait(fastmc_t* h, int slot, double* out, int64_t out_cap, int64_t* hist, int hist_cap);

/* ---- numpy's normal stream on the device (GPU_RNG 'numpy'; round 4) ---------------------------------------------------
 * The reference draws everything from one sequential stream, funcs._R = numpy.random.default_rng(seed) (fast/funcs.py:21,
 * 352-365): PCG64 feeding numpy's ziggurat, where a normal consumes one 64-bit word or, 2.2 % of the time, more.  These
 * entry points reproduce that stream on the device (fast_amd/csrc/fmc_npstream.h: classify every word as a potential start,
 * chain the tiles' transfer maps, write -- in one kernel with a decoupled look-back when the array fits a device buffer of
 * its own, in three passes otherwise), so that a run with a given SEED returns the reference's own numbers at GPU speed.
 * Environment (A/B and tests; same results): FASTMC_NPS_ONEPASS_MAX_GB (default 16) bounds that buffer; FASTMC_NPS_THREEPASS=1
 * forces the three-pass form, FASTMC_NPS_GENERAL_SCAN=1 its in-order scan; FASTMC_NPS_TWO_STREAMS=1 puts the generator of
 * chunk c + 1 on a stream of its own; FASTMC_NPS_TEST_OVERFLOW=k makes the first fastmc_run_npstream call of the process with
 * more than k chunks report chunk k as given up (the caller's redo-with-numpy path); FASTMC_GEN64_STAGED=1 stages the float64
 * device generator through HBM.
 *   fastmc_npstream_set_tables   the 256-entry ziggurat tables (wi, ki, fi) of the numpy that is installed, read out of it
 *                                by fast_amd/npnormal.py (they are not in this library);
 *   fastmc_npstream_normals      one array: out[0 ... n) = Generator(PCG64 at state_inc).normal(size = n); state_inc =
 *                                {state lo, state hi, inc lo, inc hi} as numpy's bit_generator.state holds them; state_after
 *                                (lo, hi), the words consumed, and *overflow != 0 when the device gave up (a normal spanning
 *                                more than 16 words across a tile edge, ...: the caller then draws with numpy itself);
 *   fastmc_run_npstream          chunks [0, n_chunks) of fast.Fast.run's loop (fast/fast.py:130-134): per chunk the real parts
 *                                of chunk_real realisations, then their imaginary parts (funcs.py:352-356), then the
 *                                sub-harmonic draws if sub-harmonics are set (fast.py:598-603); logamp_dev_scaled: the run's
 *                                log-amplitudes as fastmc_npstream_logamp left them on the device.  out: [n_chunks][2 chunk_real]
 *                                (x2 when coherent) as run_coeffs returns each chunk.  *bad_chunk = -1, or the first chunk whose
 *                                draw overflowed: results before it stand, state_after is the state at ITS start.
 *   fastmc_npstream_logamp       the 2 n_iter normals fast.py:639-645 draws before the chunks; keeps the first n_iter, scaled
 *                                by sqrt(logamp_var), on the device for fastmc_run_npstream and copies them to logamp (host). */
int fastmc_npstream_set_tables(int device_id, const double* wi, const uint64_t* ki, const double* fi);
int fastmc_npstream_normals(fastmc_t* h, const uint64_t state_inc[4], int64_t n, double* out, uint64_t state_after[2],
                            uint64_t* consumed, uint32_t* overflow);
int fastmc_npstream_logamp(fastmc_t* h, const uint64_t state_inc[4], int64_t n_iter, double logamp_var, double* logamp,
                           uint64_t state_after[2], uint32_t* overflow);
int fastmc_run_npstream(fastmc_t* h, const uint64_t state_inc[4], int64_t n_chunks, int64_t chunk_real, int64_t logamp_offset,
                        int coherent, double* out, uint64_t state_after[2], int64_t* bad_chunk);

/* Names of the row and column kernels the handle launched last, as c++filt prints the instantiations (e.g.
 * "k_rows_wave<double, 16, 2, 0, 1, 4>"; empty before the first run): bench.py prices the instruction mix of what actually
 * ran (fast_amd/kernel_isa_stats.json is keyed by these names).  rows / cols: caller's buffers of `cap` bytes each. */
int fastmc_last_kernels(fastmc_t* h, char* rows, char* cols, int cap);

/* Effective shader clock of the handle's last row-kernel launch (wave family, k_rows_wave): the first lane of the launch's middle
 * workgroup reads the shader-clock counter and the constant-rate counter before and after its rows; *ghz = shader ticks per second
 * in GHz, *span_us = the time between the two readings -- that workgroup's life: the whole launch when the launch's workgroups stay
 * and walk its tiles (large launches: milliseconds), one tile otherwise (tens of microseconds).  bench.py reports it as
 * `clock.effective_GHz`, so that a row time can be told apart from a box's clock (MI355X throttles under sustained float64 work:
 * 2.0-2.3 GHz against the nominal 2.4).  Blocks until the handle's stream is idle.  FASTMC_ESTATE when the handle's LAST row launch
 * was not a stamping one (another kernel family, or none yet): stamps of an earlier launch are never reported.  No counterpart in the
 * reference. */
int fastmc_last_clock(fastmc_t* h, double* ghz, double* span_us);

/* Shape of the result vector resident on the device -- what fastmc_wait copies out: *n_iter iterations (0: no results yet),
 * *coherent != 0: complex amplitudes (2 doubles per iteration).  Lets a caller size the buffer it hands to fastmc_wait after
 * any of fastmc_run / fastmc_run_async / fastmc_set_results. */
int fastmc_last_result_shape(fastmc_t* h, int64_t* n_iter, int* coherent);

/* The precision the handle COMPUTES in: FASTMC_F64 or FASTMC_F32.  fastmc_create honours FASTMC_F32 on the wave family's
 * fixed grids and the direct family only; a float32 request on a chirp-z, 50-lane or run-time-split grid runs the float64
 * kernels, and this getter says so. */
int fastmc_precision(fastmc_t* h);

/* Realisations in flight per launch (batch).  0 = library default. */
int fastmc_set_batch(fastmc_t* h, int batch);
/* The batch a run of this handle uses as it stands (the value set, or the library's default for its grid and window: a `V`
 * slab of up to 2 GiB in whole workgroup rounds): a run of n realisations is n / batch launches of `batch` and one of the
 * remainder.  Measurement hook: bench.py prints its launch plan with it, so that a profile's per-dispatch times can be read
 * against the realisations each dispatch held. */
int fastmc_get_batch(fastmc_t* h, int* batch);

/* Precision of the DEVICE generator (fastmc_run / fastmc_run_async / fastmc_screens; not of the transform, which
 * fastmc_create fixes).
 *   FASTMC_F64: the reference's precision (fast/funcs.py:352-356 draws 53-bit normals, fast/fast.py:594 colours in float64).  Four
 *     32-bit words per coefficient from ONE advance of its xoshiro128+ stream (round 5; fmc_core.h: next4) make a uniform with 53
 *     significant bits down to 2^-64 and a 56-bit angle; float64 log / sqrt / sincos in ~60 instructions per coefficient
 *     (fast_amd/csrc/fmc_gen64.h: table-driven log, seeded cubic square root, table + rotation for the angle; draws within 3e-15
 *     of the libm restatement), FUSED into the row kernels of every FFT family (wave, packed, 50-lane, run-time-split, chirp-z)
 *     wherever its 6 KB of tables fit the LDS (no coefficient passes through device memory) and staged through device memory
 *     otherwise (the direct kernels; coloured in float64 by the host-coefficient kernels).  THE STATE fastmc_create LEAVES ON A
 *     FLOAT64 HANDLE (round 5: a handle draws at the precision it computes in), what `fast_amd.Fast` selects (GPU_RNG_PRECISION
 *     'auto') and what bench.py times.
 *   FASTMC_F32 (what a float32 handle starts with; on a float64 handle the opt-in shortcut): 24-bit uniforms, hardware float32
 *     log / sqrt / sin / cos, float32 colouring, fused into the row kernels; ~1.7 x the rate at 1024^2.
 * Both restated in oracle/devrng.py; the leading 32 / 24 bits of the float64 draw's uniform and angle are the float32 draw's words.
 * fastmc_rng_coeffs / fastmc_rng_logamp return the draws of the precision in force. */
int fastmc_set_rng_precision(fastmc_t* h, int precision);

/* ---- AO-residual power spectrum (Fast.compute_powerspec, fast/fast.py:445-492) ---- */
#define FASTMC_NOAO 0
#define FASTMC_AO 1
#define FASTMC_TT 2
#define FASTMC_LGSAO 3

typedef struct {
  int32_t N;            /* Npxls */
  int32_t n_layers;     /* L */
  double dx;            /* pixel scale [m] */
  double wvl;           /* wavelength [m] */
  double L0, l0;        /* outer / inner scale [m] (L0 may be +inf) */
  int32_t ao_mode;      /* FASTMC_NOAO .. FASTMC_LGSAO (fast/ao_power_spectra.py:232-267) */
  int32_t alias;        /* include Jol_alias_openloop (ao_power_spectra.py:163-223), lmax=kmax=5 */
  double noise;         /* WFS noise variance; >0 enables Jol_noise_openloop (148-161) */
  double d_wfs;         /* sub-aperture pitch [m] */
  double t_loop, t_exp; /* loop delay, WFS exposure [s] */
  double dtheta[2];     /* point-ahead [arcsec] */
  const double* cn2;    /* (L,) zenith-corrected cn2 dh */
  const double* h;      /* (L,) zenith-corrected heights */
  const double* wind;   /* (L, 2) wind vectors */
  int32_t mask_mode;    /* mask_lf (ao_power_spectra.py:119-141) evaluated on the device: 1 zonal, 2 modal
                           radial cut (modal_mult), 3 modal Zernike (zmax, D_ground); 0 = use lf_mask below */
  int32_t zmax;         /* highest Noll index (mask_mode 3) */
  double modal_mult;
  double D_ground;      /* telescope diameter for the Zernike filters (mask_mode 3, LGSAO) */
  const double* lf_mask;      /* (N, N) mask as float64 when mask_mode == 0, else NULL */
  const double* pupil_filter; /* (N, N) funcs.pupil_filter (funcs.py:308-315) or NULL */
  const double* lgs_z;        /* (N, N) zernike_squared_filter(Z<=4) for LGSAO, or NULL = evaluate on the device */
  const double* simpson_w;    /* (N,) Simpson weights of the frequency axis (funcs.py:100-115) */
  int64_t pupil_filter_token; /* 0, or a caller-chosen name of the pupil_filter array: while consecutive calls on a device
                                 carry the same non-zero token (and N) the filter already on the device is used and the
                                 pointer is not read (the filter depends on the aperture only; sweeps over geometry reuse it) */
} fastmc_ps_params;

#define FASTMC_PS_NSCALARS 6 /* aniso_servo, alias, noise, fitting, phs_var, logamp_var */

/* Outputs (any may be NULL): powerspec (N,N); per_layer (L,N,N); logamp_ps (N,N); lf_mask_out
 * (N,N) the mask used; scalars: FASTMC_PS_NSCALARS values in the order above, then L
 * phs_var_weights; kernel_ms: HIP-event time of the kernels. */
int fastmc_powerspec(int device_id, const fastmc_ps_params* p, double* powerspec,
                     double* per_layer, double* logamp_ps, double* lf_mask_out, double* scalars,
                     double* kernel_ms);

/* The same evaluation on h's device, left there: the spectrum becomes the handle's colouring tables (as
 * fastmc_set_spectrum(h, powerspec, df) would make them) without crossing PCIe, and powerspec, the log-amplitude spectrum
 * and the mask stay resident for fastmc_powerspec_get (which = 0, 1, 2; each (N, N)).  This is what a sweep pays per
 * geometry sample (fast/complete_orbit_simulation.py:217-228 builds one Fast per sample; compute_powerspec is 12-15 s of
 * each in the reference).  scalars / kernel_ms as above. */
int fastmc_powerspec_set(fastmc_t* h, const fastmc_ps_params* p, double df, double* scalars, double* kernel_ms);
int fastmc_powerspec_get(fastmc_t* h, int which, double* out);

/* The terms of that assembly as the reference keeps them on the object (fast/fast.py:448-472; any may be NULL):
 * turb (L,N,N) = funcs.turb_powerspectrum_vonKarman (fast/funcs.py:138-173); g_ao (L,N,N) = G_AO_PAOLA
 * (fast/ao_power_spectra.py:225-270; all ones for FASTMC_NOAO); alias (L,N,N) = Jol_alias_openloop (163-223;
 * zeros when p->alias is off or NOAO); noise (N,N) = Jol_noise_openloop (148-161; zeros when p->noise == 0). */
int fastmc_powerspec_terms(int device_id, const fastmc_ps_params* p, double* turb, double* g_ao,
                           double* alias, double* noise);

/* ---- multi-GPU result exchange: RCCL over xGMI ----
 * Iterations are independent (fast/fast.py:130-134 loops over chunks, 589-605 draws each chunk afresh), so the
 * realisation range is cut into one contiguous piece per GPU and the only exchange is at the end of a run:
 * an all-gather of the per-iteration powers (8 B each) and an all-reduce of the dB histogram.  The reference has no
 * counterpart (single process, single thread).
 * A communicator belongs to a DEVICE of this process and serves every handle on that device (sweeps build many
 * short-lived handles); collectives on one device's communicator must be serialised by the caller.
 *   one process drives n devices:  fastmc_comm_init_all(handles, n) (ncclCommInitAll), then fastmc_comm_gather_all;
 *   one process per GPU:           rank 0 calls fastmc_comm_unique_id, the launcher distributes the 128 bytes, every
 *                                  rank calls fastmc_comm_init (ncclCommInitRank), then fastmc_comm_gather.
 * FASTMC_DISABLE_RCCL=1 in the environment makes the init calls fail with FASTMC_ECOMM (callers then exchange
 * through the host: fast_amd/dist.py, fast_amd/multi.py). */
int fastmc_comm_unique_id(uint8_t id128[128]);
int fastmc_comm_init(fastmc_t* h, const uint8_t id128[128], int world_size, int rank);
/* handles[i] becomes rank i of an n-rank communicator clique; one handle per device, distinct devices. */
int fastmc_comm_init_all(fastmc_t* const* handles, int n);
/* world size and rank of the communicator of h's device (0 and -1 when there is none). */
int fastmc_comm_world(fastmc_t* h, int* world_size, int* rank);
/* All-gather of each rank's last-run results (n_local float64 values per rank, equal on all ranks) and all-reduce
 * (sum) of its histogram, on the device buffers, then copied to the host arrays (either may be NULL). */
int fastmc_comm_gather(fastmc_t* h, int64_t n_local, double* all_powers /* world*n_local */,
                       int64_t* hist /* nbins+2, in: unused, out: global */, double lo_db,
                       double hi_db, int nbins);
/* The same exchange for the n handles of fastmc_comm_init_all (same handles, same order), issued from one thread
 * as one RCCL group per collective; the host arrays are filled from rank 0's copy. */
int fastmc_comm_gather_all(fastmc_t* const* handles, int n, int64_t n_local, double* all_powers, int64_t* hist,
                           double lo_db, double hi_db, int nbins);
/* Destroys the communicator of h's device (no-op when there is none). */
int fastmc_comm_destroy(fastmc_t* h);
/* Aborts the communicator of h's device (ncclCommAbort: outstanding collectives are torn down and their kernels leave
 * the stream) and forgets it; may be called from another thread while fastmc_comm_gather / _gather_all of that device
 * is blocked -- that call then returns FASTMC_ECOMM.  What a caller does when a collective misses its deadline
 * (fast_amd/multi.py, fast_amd/dist.py) or when a clique was only half built; never waits for peers.  No-op without
 * a communicator. */
int fastmc_comm_abort(fastmc_t* h);
/* HIP-event time of the collectives of the last fastmc_comm_gather / _gather_all on h's stream, from the moment the
 * stream reached them (i.e. after this device's kernels) to their completion: transfer plus the wait for the slowest
 * peer.  ms: one double. */
int fastmc_last_exchange_ms(fastmc_t* h, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* FASTMC_H */
