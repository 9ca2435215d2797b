"""CPU: the oracle's restatement of OUR device generator (oracle/devrng.py).

The generator has no counterpart in the reference (numpy PCG64, fast/funcs.py:21,352-356), so it is pinned to the
published algorithms it is built from and to its own statistical quality:
  * Philox4x32 at 7 and 10 rounds against Random123's known-answer vectors (kat_vectors of the Random123
    distribution, Salmon et al. SC'11);
  * xoshiro128+ against a scalar transcription of Blackman & Vigna's reference code;
  * the two-words-per-state-advance output (s0 + s3, s1 + s2) and the 23-bit angle: moments, tails, uniformity and
    independence of the resulting normals on 2 x 512^2 draws (the GPU suite repeats this on the device's own draws).
"""
import numpy as np
import pytest

from oracle import devrng

KAT = [  # rounds, counter, key -> output (Random123 kat_vectors: "philox4x32 R c0 c1 c2 c3 k0 k1  o0 o1 o2 o3")
    (7, (0, 0, 0, 0), (0, 0), (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)),
    (7, (0xffffffff,) * 4, (0xffffffff,) * 2, (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)),
    (7, (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a)),
    (10, (0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    (10, (0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    (10, (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("rounds,ctr,key,want", KAT)
def test_philox_known_answers(rounds, ctr, key, want):
    got = devrng.philox4x32_10(*ctr, *key, rounds=rounds)
    assert tuple(int(x) for x in got) == want


def _xoshiro_scalar(s, n):
    """Blackman & Vigna, xoshiro128plus.c: result = s[0] + s[3]; t = s[1] << 9; s[2] ^= s[0]; s[3] ^= s[1];
    s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 11)."""
    M = 0xFFFFFFFF
    s = list(s)
    out = []
    for _ in range(n):
        out.append(((s[0] + s[3]) & M, (s[1] + s[2]) & M))
        t = (s[1] << 9) & M
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = ((s[3] << 11) | (s[3] >> 21)) & M
    return out


def test_xoshiro_pair_output_follows_the_reference_engine():
    seed = (0x12345678, 0x9abcdef0, 0x0fedcba9, 0x87654321)
    s = [np.array([w], dtype=np.uint32) for w in seed]
    want = _xoshiro_scalar(seed, 40)
    with np.errstate(over="ignore"):
        for a_w, b_w in want:
            a, b = devrng.xoshiro128p_next2(s)
            assert (int(a[0]), int(b[0])) == (a_w, b_w)
        # the single-word form is the engine's own output
        s2 = [np.array([w], dtype=np.uint32) for w in seed]
        assert [int(devrng.xoshiro128p_next(s2)[0]) for _ in range(5)] == [a for a, _ in want[:5]]


def test_box_muller_definition():
    a = np.array([0, 1, 2 ** 31, 2 ** 32 - 1], dtype=np.uint32)
    b = np.array([0, 2 ** 9, 2 ** 31, 2 ** 32 - 1], dtype=np.uint32)
    c = devrng.box_muller(a, b)
    u = (a.astype(float) + 0.5) / 2 ** 32
    t = (b.astype(np.uint64) >> np.uint64(9)).astype(float) / 2 ** 23
    np.testing.assert_allclose(c, np.sqrt(-2 * np.log(u)) * np.exp(2j * np.pi * t), rtol=1e-14, atol=1e-14)
    assert abs(c[0]) == pytest.approx(np.sqrt(-2 * np.log(0.5 / 2 ** 32)))      # 6.66 sigma: the largest radius


def test_generator_statistics():
    from scipy import stats
    N = 512
    c = np.stack([devrng.device_coefficients(2024, g, N) for g in range(2)])
    z = np.concatenate([c.real.ravel(), c.imag.ravel()])
    n = z.size
    assert abs(z.mean()) < 5 / np.sqrt(n)
    assert abs(z.var() - 1) < 5 * np.sqrt(2 / n)
    assert abs(stats.skew(z)) < 5 * np.sqrt(6 / n)
    assert abs(stats.kurtosis(z)) < 5 * np.sqrt(24 / n)
    for k in (3.0, 4.0):
        expect = n * 2 * stats.norm.sf(k)
        assert abs(np.count_nonzero(np.abs(z) > k) - expect) < 6 * np.sqrt(expect) + 3
    # the two words of one state advance: radius and angle of the SAME coefficient, and of neighbouring steps
    u = np.exp(-np.abs(c) ** 2 / 2)
    t = (np.angle(c) + np.pi) / (2 * np.pi)
    assert stats.kstest(u.ravel()[::5], "uniform").pvalue > 1e-3
    assert stats.kstest(t.ravel()[::5], "uniform").pvalue > 1e-3

    def chi2_z(a, b, bins=32):
        H, _, _ = np.histogram2d(a.ravel(), b.ravel(), bins=bins, range=[[0, 1], [0, 1]])
        e = a.size / bins ** 2
        dof = bins ** 2 - 1
        return (((H - e) ** 2 / e).sum() - dof) / np.sqrt(2 * dof)
    S = 64                                           # stream step: coefficient j -> j + 1 of one stream is kx -> kx + 64
    for a, b in ((u, t), (u[:, :, :-S], u[:, :, S:]), (t[:, :, :-S], t[:, :, S:]), (u[:, :, :-S], t[:, :, S:]),
                 (t[:, :, :-S], u[:, :, S:]), (u[:, :, :-1], u[:, :, 1:]), (u[:, :-1], u[:, 1:])):
        assert abs(chi2_z(a, b)) < 5

    def corr(a, b):
        return abs(np.mean(a * b)) * np.sqrt(a.size)
    re, im = c.real, c.imag
    assert corr(re[:, :, :-1], re[:, :, 1:]) < 5 and corr(re[:, :, :-S], re[:, :, S:]) < 5
    assert corr(re[:, :-1], re[:, 1:]) < 5 and corr(re[0], re[1]) < 5 and corr(re, im) < 5


def test_stream_layout_of_the_50_lane_grids():
    """N = 50 P S grids (fmc_core.h: mr_split / stream_lanes) are drawn as 50 S streams per row, the others as 64 S'."""
    assert [(n, devrng.mr_split(n)) for n in range(2, 4097) if devrng.mr_supported(n)] == [
        (100, 1), (150, 1), (200, 1), (250, 1), (300, 1), (350, 1), (400, 1), (450, 1), (500, 1), (600, 1), (700, 1), (800, 1),
        (900, 1), (1000, 1), (1200, 1), (1350, 3), (1400, 2), (1500, 3), (1600, 2), (1750, 5), (1800, 2), (2000, 2), (2100, 3),
        (2250, 5), (2400, 2), (2500, 5), (2700, 3), (2800, 4), (3000, 3), (3200, 4), (3500, 5), (3600, 3), (4000, 4)]
    assert devrng.stream_lanes(1000) == 50 and devrng.stream_lanes(1024) == 64 and devrng.stream_lanes(2048) == 128
    assert devrng.stream_lanes(2000) == 100 and devrng.stream_lanes(2500) == 250 and devrng.stream_lanes(4000) == 200
    assert devrng.stream_lanes(550) == 64 and devrng.stream_lanes(164) == 64 and devrng.stream_lanes(1100) == 64
    # packed rows (four / two rows per wavefront): sixteen draws per stream
    assert devrng.stream_lanes(256) == 16 and devrng.stream_lanes(512) == 32 and devrng.stream_lanes(128) == 8 and devrng.stream_lanes(2048) == 128
    # the packed sub-rows (round 6): N = 256 S, S = 3, 5, 7 -- sixteen draws per stream, as on the packed grids
    assert devrng.stream_lanes(768) == 48 and devrng.stream_lanes(1280) == 80 and devrng.stream_lanes(1792) == 112
    assert devrng.stream_lanes(1536) == 96 and devrng.stream_lanes(896) == 56 and devrng.stream_lanes(1152) == 72 and devrng.stream_lanes(640) == 40
    assert devrng.stream_lanes(576) == 72 and devrng.stream_lanes(448) == 56 and devrng.stream_lanes(320) == 40 and devrng.stream_lanes(192) == 24      # eight draws per stream
    assert devrng.stream_lanes(384) == 24 and devrng.stream_lanes(1024) == 64
    # wave-family grids with a run-time sub-row count: 64 S streams per row
    assert [(n, devrng.wave_rt_split(n)) for n in range(2, 4097) if devrng.wave_rt_split(n)] == [
        (1344, 3), (1728, 3), (1920, 3), (2304, 2), (2560, 2), (2688, 3), (3072, 2), (3456, 3), (3584, 4), (3840, 3)]
    # ... for their host-coefficient rows; their generator layout is the packed sub-rows' (fmc_core.h: pks_rt -- N / 16 streams of
    # sixteen draws with 256 / 128-point sub-rows, N / 8 of eight with 64-point ones)
    assert devrng.stream_lanes(3072) == 192 and devrng.stream_lanes(1344) == 168 and devrng.stream_lanes(3584) == 224
    assert devrng.stream_lanes(1728) == 216 and devrng.stream_lanes(1920) == 120 and devrng.stream_lanes(2688) == 168 and devrng.stream_lanes(3840) == 240
    assert all(devrng.pks_split(n) for n in range(2, 4096) if devrng.wave_rt_split(n))
    # ... and every other multiple of 64 up to 8192 (chirp-z and 50-lane grids among them)
    assert devrng.stream_lanes(3200) == 200 and devrng.stream_lanes(1600) == 200 and devrng.stream_lanes(960) == 120 and devrng.stream_lanes(704) == 88
    assert devrng.stream_lanes(2112) == 264 and devrng.stream_lanes(2240) == 280 and devrng.stream_lanes(4032) == 504 and devrng.stream_lanes(8192) == 512 and devrng.stream_lanes(4160) == 520 and devrng.stream_lanes(5000) == 250 and devrng.stream_lanes(3968) == 248 and devrng.stream_lanes(2816) == 176
    assert devrng.stream_lanes(1400) == 100 and devrng.stream_lanes(800) == 50 and devrng.stream_lanes(4096) == 256
    c2 = devrng.device_coefficients(5, 0, 1400)[:3]
    assert np.isfinite(c2).all() and len(np.unique(c2.ravel())) == 3 * 1400
    c = devrng.device_coefficients(5, 0, 100)
    assert np.isfinite(c).all() and len(np.unique(c.ravel())) == 100 * 100
    # column kx belongs to stream kx mod 50 at step kx // 50: the first 50 columns are the first draw of every stream
    c64 = devrng.device_coefficients(5, 0, 96)          # 64 streams per row: a different layout, different values
    assert not np.allclose(c[:96, :50], c64[:, :50])


def test_split_layouts_cover_the_grid():
    """2048 / 4096 draw 128 / 256 streams per row (fmc_core.h: spec_split); every coefficient is drawn once."""
    for N in (2048,):
        c = devrng.device_coefficients(5, 0, N)
        assert np.isfinite(c).all() and len(np.unique(c[:4].ravel())) == 4 * N


def test_float64_generator_restatement_extends_the_float32_one():
    """oracle/devrng.py: device_coefficients_f64 (GPU_RNG_PRECISION 'f64') -- standard complex normals whose leading bits
    are the float32 generator's: the two restatements agree to ~2^-24 relative in u and t."""
    N = 256
    a = devrng.device_coefficients(11, 3, N)
    b = devrng.device_coefficients_f64(11, 3, N)
    assert np.abs(a - b).max() < 2e-5 and np.abs(a - b).max() > 0
    z = np.concatenate([devrng.device_coefficients_f64(s, g, N).ravel() for s in (1, 2) for g in (0, 5)])
    z = np.concatenate([z.real, z.imag])
    n = z.size
    assert abs(z.mean()) < 5 / np.sqrt(n) and abs(z.var() - 1) < 5 * np.sqrt(2 / n)
    assert abs(np.mean(z ** 4) - 3) < 5 * np.sqrt(96 / n)
    la32, la64 = devrng.device_logamp_normals(3, 10, 1000), devrng.device_logamp_normals(3, 10, 1000, f64=True)
    assert np.abs(la32 - la64).max() < 2e-5 and abs(la64.std() - 1) < 0.1
    s32, s64 = devrng.device_subharm_coefficients(3, 7), devrng.device_subharm_coefficients(3, 7, f64=True)
    assert s64.shape == (3, 3, 3) and np.abs(s32 - s64).max() < 2e-5
    # extremes of the words stay finite: u in (0, 1], tails to 9.4 sigma (u >= 2^-64)
    top = np.array([0xFFFFFFFF], dtype=np.uint32)
    zero = np.array([0], dtype=np.uint32)
    assert np.isfinite(devrng.box_muller_f64(zero, zero, zero, zero)).all() and abs(devrng.box_muller_f64(zero, zero, zero, zero)[0]) > 9.4
    assert np.isfinite(devrng.box_muller_f64(top, top, top, top)).all() and abs(devrng.box_muller_f64(top, top, top, top)[0]) < 1e-7


def test_four_words_of_one_state_advance():
    """fmc_core.h: xoshiro128p::next4 -- the float64 generator takes (a, b, a2, b2) from ONE state advance: a2 and b2 (the bits
    below the 32 leading ones of the uniform and of the angle) must be uniform, independent of (a, b), of each other and of the
    neighbouring steps' words; a2 is odd by construction.  Known answers first (hand-evaluated from the definition)."""
    s = [np.array([w], dtype=np.uint32) for w in (1, 2, 3, 4)]
    a, b, a2, b2 = (int(w[0]) for w in devrng.xoshiro128p_next4(s))
    assert (a, b) == (5, 5)
    assert a2 == (5 * 0x9E3779 + 1) | 1 and b2 == 5 * 0x85EBCB + 2
    assert [int(w[0]) for w in s] == [int(w[0]) for w in _advance_by_hand(1, 2, 3, 4)]
    # statistics over the streams of two realisations at 512^2 (2 x 512 rows x 64 streams x 8 steps)
    N, SL = 512, 64
    words = []
    for g in range(2):
        st = devrng._stream_states(2025, g, N, devrng.STREAM_SCREEN)
        with np.errstate(over="ignore"):
            words.append(np.stack([np.stack(devrng.xoshiro128p_next4(st)) for _ in range(N // SL)]))      # (steps, 4, N, SL)
    w = np.concatenate(words, axis=2).astype(np.float64) / 2.0 ** 32                                      # (steps, 4, 2N, SL) in [0, 1)
    a, b, a2, b2 = (w[:, i] for i in range(4))
    assert (np.concatenate(words, axis=2)[:, 2] & 1).all()
    from scipy import stats
    for x in (a, b, a2, b2):
        assert stats.kstest(x.ravel()[::7], "uniform").pvalue > 1e-3

    def chi2_z(x, y, bins=32):
        H, _, _ = np.histogram2d(x.ravel(), y.ravel(), bins=bins, range=[[0, 1], [0, 1]])
        e = x.size / bins ** 2
        dof = bins ** 2 - 1
        return (((H - e) ** 2 / e).sum() - dof) / np.sqrt(2 * dof)
    pairs = [(a, a2), (b, b2), (a, b2), (b, a2), (a2, b2), (a2[:-1], a2[1:]), (b2[:-1], b2[1:]), (a2[:-1], a[1:]), (b2[:-1], b[1:]),
             (a[:-1], a2[1:]), (b[:-1], b2[1:])]
    for x, y in pairs:
        assert abs(chi2_z(x, y)) < 5
    # the low HALF of each extra word too (bits 0 ... 15 of b2 feed the angle below 2^-40, bits 1 ... 15 of a2 the uniform below 2^-48)
    lo = lambda x: (x * 2.0 ** 16) % 1.0      # noqa: E731
    for x, y in ((a, lo(a2)), (b, lo(b2)), (lo(a2), lo(b2)), (lo(a2)[:-1], lo(a2)[1:]), (lo(b2)[:-1], lo(b2)[1:])):
        assert abs(chi2_z(x, y)) < 5
    # the float64 uniform and angle as the generator forms them: the part BELOW the float32 draw's bits is uniform and independent of it
    wi = np.concatenate(words, axis=2).astype(np.uint64)
    u_lo = wi[:, 2].astype(np.float64) / 2.0 ** 32                       # bits 33 ... 64 of u
    t_lo = wi[:, 3].astype(np.float64) / 2.0 ** 32                       # bits 25 ... 56 of t
    assert abs(chi2_z(a, u_lo)) < 5 and abs(chi2_z(b, t_lo)) < 5 and abs(chi2_z(u_lo, t_lo)) < 5


def _advance_by_hand(s0, s1, s2, s3):
    """xoshiro128 state transition written out (Blackman & Vigna's reference formulation)."""
    M = 0xFFFFFFFF
    t = (s1 << 9) & M
    s2 ^= s0
    s3 ^= s1
    s1 ^= s2
    s0 ^= s3
    s2 ^= t
    s3 = ((s3 << 11) | (s3 >> 21)) & M
    return [np.array([x], dtype=np.uint32) for x in (s0, s1, s2, s3)]
