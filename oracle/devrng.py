"""Oracle restatement of the DEVICE generator (test infrastructure, not product).

The GPU path can draw its complex Gaussian coefficients on the device
(fast_amd/csrc/fmc_core.h: philox4x32_10; fmc_kernels.h: box_muller, draw_pair).  That
generator has no counterpart in the reference (which uses numpy's PCG64 stream,
fast/funcs.py:21,352-356); this module restates OUR definition in numpy/float64 so that the
device path can be checked deterministically, not only statistically:

  stream (g, ky, l = kx mod SL): xoshiro128+ seeded with Philox4x32-7(ctr=(ky*SL+l, STREAM, g_lo, g_hi), key=seed),
  SL = stream_lanes(N) (64 for most grids)
  coefficient (ky, l + SL j) = BM(a_j, b_j), (a_j, b_j) = (s0 + s3, s1 + s2) of the state after j advances
  (two words per state advance; jointly equidistributed over the period)
  float32 draw (opt-in since round 5):  BM(a, b) = sqrt(-2 ln((a+.5)/2^32)) * exp(2 pi i (b >> 9)/2^23)
  float64 generator (the default):      four words per advance (xoshiro128p_next4: a, b and two multiply-add scrambles a2, b2),
                                        BM64 = sqrt(-2 ln(RNE(a 2^32 + (a2|1)) 2^-64)) * exp(2 pi i ((b >> 8) 2^32 + b2) 2^-56)
  (log-amplitude and sub-harmonic draws use Philox blocks directly)

Philox4x32-10 is the published Random123 algorithm (Salmon et al., SC'11); `tests/test_oracle_devrng.py`
pins this implementation to Random123's known-answer vectors.
The device evaluates BM in float32 with hardware log2/sqrt/sin/cos; agreement is ~1e-6 absolute.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
STREAM_SCREEN, STREAM_LOGAMP, STREAM_SUBHARM = 0, 1, 2
STREAM_SUBHARM_LO = 4      # second Philox blocks of the sub-harmonic draws at float64 precision (fmc_kernels.h)


SEED_ROUNDS = 7   # fmc_kernels.h: FMC_SEED_ROUNDS -- Philox rounds of the block that seeds a coefficient stream


def philox4x32_10(c0, c1, c2, c3, k0, k1, rounds=10):
    """Philox4x32 with `rounds` rounds (10: the Random123 default the known-answer tests pin; the stream seeds use
    SEED_ROUNDS).  Vectorised over the counter words (uint32 arrays or scalars); key words are ints."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(rounds):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def box_muller(a, b):
    u = (a.astype(np.float64) + 0.5) / 2.0 ** 32
    t = (np.asarray(b).astype(np.uint64) >> np.uint64(9)).astype(np.float64) / 2.0 ** 23   # top 23 bits (fmc_kernels.h: angle_turns)
    r = np.sqrt(-2.0 * np.log(u))
    return r * np.cos(2 * np.pi * t) + 1j * r * np.sin(2 * np.pi * t)


def box_muller_f64(a, b, a2, b2):
    """fmc_kernels.h: box_muller_f64 -- the float64 generator (GPU_RNG_PRECISION 'f64', round-5 definition): four 32-bit
    words make one complex normal,
      u = RNE(a 2^32 + (a2 | 1)) 2^-64,   t = ((b >> 8) 2^32 + b2) 2^-56 turns,   sqrt(-2 ln u) exp(2 pi i t):
    53 significant bits of the uniform at every magnitude down to 2^-64, a 56-bit angle; their leading 32 / 24 bits are the
    words (a, b >> 8) of the float32 draw (`box_muller`)."""
    a, b, a2, b2 = (np.asarray(w).astype(np.uint64) for w in (a, b, a2, b2))
    v = ((a << np.uint64(32)) | (a2 | np.uint64(1))).astype(np.float64)         # uint64 -> float64: round to nearest even
    u = v * 2.0 ** -64
    r = np.sqrt(-2.0 * np.log(u))
    # sin / cos of 2 pi t: the 56-bit angle reduced EXACTLY to the nearest quarter turn (|rem| <= 2^53 is a float64; x = 2 pi
    # 2^-56 rem carries one rounding of a small angle)
    T = ((b >> np.uint64(8)) << np.uint64(32)) | b2
    q = (T + (np.uint64(1) << np.uint64(53))) >> np.uint64(54)
    rem = (T - (q << np.uint64(54))).astype(np.int64)                            # wraps to the signed remainder
    x = rem.astype(np.float64) * (2 * np.pi * 2.0 ** -56)
    sn, cs = np.sin(x), np.cos(x)
    qi = q.astype(np.int64) & 3
    cos_t = np.where(qi == 0, cs, np.where(qi == 1, -sn, np.where(qi == 2, -cs, sn)))
    sin_t = np.where(qi == 0, sn, np.where(qi == 1, cs, np.where(qi == 2, -sn, -cs)))
    return r * cos_t + 1j * r * sin_t


def xoshiro128p_next(s):
    """One step of xoshiro128+ on four uint32 arrays (in place); returns the output words."""
    s0, s1, s2, s3 = s
    r = (s0 + s3).astype(np.uint32)
    t = (s1 << np.uint32(9)).astype(np.uint32)
    s2 ^= s0
    s3 ^= s1
    s1 ^= s2
    s0 ^= s3
    s2 ^= t
    s[3] = ((s3 << np.uint32(11)) | (s3 >> np.uint32(21))).astype(np.uint32)
    return r


def xoshiro128p_next2(s):
    """Two words from ONE state advance (fmc_core.h: xoshiro128p::next2): a = s0 + s3, b = s1 + s2."""
    a = (s[0] + s[3]).astype(np.uint32)
    b = (s[1] + s[2]).astype(np.uint32)
    xoshiro128p_next(s)
    return a, b


def xoshiro128p_next4(s):
    """Four words from ONE state advance (fmc_core.h: xoshiro128p::next4): (a, b) as next2,
    a2 = ((a mod 2^24) 0x9E3779 + s0) | 1, b2 = (b mod 2^24) 0x85EBCB + s1 (mod 2^32)."""
    a = (s[0] + s[3]).astype(np.uint32)
    b = (s[1] + s[2]).astype(np.uint32)
    m24 = np.uint64(0xFFFFFF)
    a2 = (((a.astype(np.uint64) & m24) * np.uint64(0x9E3779) + s[0]) & MASK).astype(np.uint32) | np.uint32(1)
    b2 = (((b.astype(np.uint64) & m24) * np.uint64(0x85EBCB) + s[1]) & MASK).astype(np.uint32)
    xoshiro128p_next(s)
    return a, b, a2, b2


def _radix_ok(P):
    """fmc_core.h: mr_supported_P -- 2^k times 1, 3, 5, 7 or 9, 2 <= P <= 32."""
    return 2 <= P <= 32 and P // (P & -P) in (1, 3, 5, 7, 9)


def wave_rt_split(N):
    """fmc_core.h: wave_rt_split -- grids N = 64 q with q not a radix of the wave family: S = 2 ... 4 sub-rows when that
    leaves 7 <= P <= 24 (1344, 1728, 1920, 2304, 2560, 2688, 3072, 3456, 3584, 3840), else 0."""
    if N % 64 or N in (2048, 4096):
        return 0
    q = N // 64
    if _radix_ok(q):
        return 0
    for S in range(2, (8 if N > 4096 else 4) + 1):          # beyond 4096: up to eight sub-rows (8192 = 8 x 1024)
        if q % S == 0 and 7 <= q // S <= 24 and _radix_ok(q // S):
            return S
    return 0


def spec_split(N):
    """fmc_core.h: spec_split -- sub-rows per row of the wave kernels, which fixes the stream layout."""
    return 4 if N == 4096 else (2 if N == 2048 else (wave_rt_split(N) or 1))


def mr_split(N):
    """fmc_core.h: mr_split -- sub-rows S of the 50-lane kernel family (N = 50 P S, P = 2^k times 1, 3, 5, 7 or 9): 1 for
    P <= 24, else the smallest S <= 5 that leaves 7 <= P <= 24; 0 when N is not a size of the family (sizes of the wave
    family, N = 64 P', stay there)."""
    ok = _radix_ok
    if N < 100 or N % 50:
        return 0
    if N in (2048, 4096) or (N % 64 == 0 and (ok(N // 64) or wave_rt_split(N))):
        return 0
    q = N // 50
    if q <= 24:
        return 1 if ok(q) else 0
    for S in range(2, (8 if N > 4096 else 5) + 1):          # beyond 4096: up to eight sub-rows (8000 = 8 x 1000)
        if q % S == 0 and 7 <= q // S <= 24 and ok(q // S):
            return S
    return 0


def mr_supported(N):
    return mr_split(N) > 0


def pk_grid(N):
    """fmc_core.h: pk_grid -- grids of the packed rows (eight / four / two rows per wavefront)."""
    return N in (128, 256, 512)


def pks_split(N):
    """fmc_core.h: pks_split -- grids of the packed SUB-ROWS (round 6): N = S * 256, else S * 128, else S * 64 -- every multiple of
    64 from 192 to 8192 except the packed grids (128, 256, 512) and the grids of the P = 16 rows (1024, 2048, 4096); 0 otherwise."""
    if N % 64 or N < 192 or N > 8192 or N in (256, 512, 1024, 2048, 4096):
        return 0
    if N % 256 == 0:
        return N // 256
    if N % 128 == 0:
        return N // 128
    return N // 64


def stream_lanes(N):
    """fmc_core.h: stream_lanes -- generator streams per row: N / 16 on the packed grids (128, 256, 512) and on the grids of
    the packed sub-rows of 256 / 128 points (384, 640, 768, ... 3968: multiples of 128): sixteen draws per stream; N / 8 on those of 64 points
    (192, 320, 448, ... 4032: odd multiples of 64): eight draws; 50 S on the 50-lane grids, else 64 * spec_split(N)."""
    if pks_split(N) and N % 128:        # sub-rows of 64 points (odd multiples of 64): eight draws per stream
        return N // 8
    if pk_grid(N) or pks_split(N):
        return N // 16
    return 50 * mr_split(N) if mr_supported(N) else 64 * spec_split(N)


def _stream_states(seed, g, N, stream):
    SL = stream_lanes(N)
    lanes = min(SL, N)
    ky, l = np.meshgrid(np.arange(N), np.arange(lanes), indexing="ij")
    x = philox4x32_10(ky * SL + l, stream, g & 0xFFFFFFFF, g >> 32, seed & 0xFFFFFFFF, seed >> 32, rounds=SEED_ROUNDS)
    s = [np.array(w, dtype=np.uint64).astype(np.uint32) for w in x]
    zero = (s[0] | s[1] | s[2] | s[3]) == 0
    s[0][zero] = 1
    return s


def device_coefficients_f64(seed, g, N):
    """(N, N) complex coefficients of realisation g with the generator at float64 precision (fastmc_set_rng_precision
    FASTMC_F64; == fastmc_rng_coeffs then): the SAME streams as `device_coefficients`, four words per state advance
    (`xoshiro128p_next4`) combined by `box_muller_f64`."""
    SL = stream_lanes(N)
    s = _stream_states(seed, g, N, STREAM_SCREEN)
    out = np.empty((N, N), dtype=complex)
    with np.errstate(over="ignore"):
        for j in range((N + SL - 1) // SL):
            c = box_muller_f64(*xoshiro128p_next4(s))
            w = min(SL, N - SL * j)
            out[:, SL * j:SL * j + w] = c[:, :w]
    return out


def device_coefficients(seed, g, N):
    """(N, N) complex coefficients of realisation g (== fastmc_rng_coeffs).

    SL = stream_lanes(N) streams per row (64; 128 / 256 at 2048 / 4096; 8 / 16 / 32 at 128 / 256 / 512; N / 16 at 192, 320, 448, 576, 640, 768, 896, 1152, 1280, 1536, 1792; 50 S on the 50 P S grids).  Stream (g, ky, L = kx mod SL): state = Philox4x32-7(ctr =
    (ky*SL + L, STREAM_SCREEN, g_lo, g_hi), key = seed) (s0 := 1 if the block is all zero); coefficient
    (ky, L + SL j) = BM(s0 + s3, s1 + s2) of the xoshiro128+ state after j advances (fmc_core.h: xoshiro128p::next2,
    fmc_kernels.h: row_stream / draw_words).  The device colours these float32 normals with sqrt(powerspec) * df
    rounded to float32 (fmc_kernels.h: draw_coloured); the restatement keeps float64 throughout (difference ~6e-8
    relative per coefficient, inside the 2e-3 bar of the device-generator tests)."""
    SL = stream_lanes(N)
    lanes = min(SL, N)
    ky, l = np.meshgrid(np.arange(N), np.arange(lanes), indexing="ij")
    x = philox4x32_10(ky * SL + l, STREAM_SCREEN, g & 0xFFFFFFFF, g >> 32, seed & 0xFFFFFFFF, seed >> 32, rounds=SEED_ROUNDS)
    s = [np.array(w, dtype=np.uint64).astype(np.uint32) for w in x]
    zero = (s[0] | s[1] | s[2] | s[3]) == 0
    s[0][zero] = 1
    out = np.empty((N, N), dtype=complex)
    with np.errstate(over="ignore"):
        for j in range((N + SL - 1) // SL):
            a, b = xoshiro128p_next2(s)
            c = box_muller(a, b)
            w = min(SL, N - SL * j)
            out[:, SL * j:SL * j + w] = c[:, :w]
    return out


def device_logamp_normals(seed, it0, n, f64=False):
    """Log-amplitude normals of iterations it0 ... it0 + n - 1 (one Philox4x32-10 block each); f64: all four words of the
    block (fmc_kernels.h: draw_logamp_normal_f64), else the float32 draw's two."""
    it = np.arange(it0, it0 + n, dtype=np.uint64)
    x0, x1, x2, x3 = philox4x32_10(0, STREAM_LOGAMP, it & MASK, it >> np.uint64(32), seed & 0xFFFFFFFF, seed >> 32)
    return box_muller_f64(x0, x1, x2, x3).real if f64 else box_muller(x0, x1).real


def device_subharm_coefficients(seed, g, f64=False):
    """(3, 3, 3) complex coefficients of realisation g (pairs (m, m+14)); f64: low bits from the blocks of STREAM_SUBHARM_LO."""
    m = np.arange(14)
    x0, x1, x2, x3 = philox4x32_10(m, STREAM_SUBHARM, g & 0xFFFFFFFF, g >> 32, seed & 0xFFFFFFFF, seed >> 32)
    out = np.empty(27, dtype=complex)
    if f64:
        y0, y1, y2, y3 = philox4x32_10(m, STREAM_SUBHARM_LO, g & 0xFFFFFFFF, g >> 32, seed & 0xFFFFFFFF, seed >> 32)
        out[:14] = box_muller_f64(x0, x1, y0, y1)
        out[14:] = box_muller_f64(x2, x3, y2, y3)[:13]
    else:
        out[:14] = box_muller(x0, x1)
        out[14:] = box_muller(x2, x3)[:13]
    return out.reshape(3, 3, 3)
