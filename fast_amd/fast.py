"""`fast_amd.Fast(config).run()` -> `FastResult`: the reference's public surface
(fast/fast.py:20-140, 931-994) with the Monte-Carlo hot path on an MI355X.

    sim = fast_amd.Fast(params)      # host init (host.py) + AO-residual PSD on the GPU
    res = sim.run()                  # batched screens + detector on the GPU
    res.dB_rel, res.power, res.scintillation_index, sim.I, sim.link_budget, sim.powerspec ...

Differences from the reference, all deliberate:
  * no CPU path: libfastmc.so and a gfx950 device are required (FastMCError otherwise);
  * `GPU_RNG`: 'device' (default) draws the coefficients on the GPU with a counter-based
    generator -- results depend only on (SEED, iteration index), not on NCHUNKS, batch size or
    the number of GPUs; 'host' draws them with numpy in the reference's order, so the same
    SEED reproduces the reference's `result._r` to ~1e-10 (parity mode, PCIe-bound); 'numpy' draws
    THAT stream -- numpy's PCG64 + ziggurat, word for word -- on the GPU: the reference's numbers
    for its SEED without a host draw (fast_amd/npnormal.py, csrc/fmc_npstream.h);
  * `GPU_PRECISION`: 'f64' (default, complex128 like the reference) or 'f32';
  * `GPU_RNG_PRECISION`: 'auto' (default: the pipeline's precision, i.e. 'f64' unless GPU_PRECISION is 'f32'), 'f64' -- the
    device generator's normals and the colouring multiply at the reference's float64 precision (funcs.py:352-356, fast.py:594),
    fused into the row kernels of every FFT family, staged through device memory by the direct kernels -- or 'f32', the opt-in
    shortcut (float32 normals and colouring into the complex128 transform: ~1.7 x the rate at 1024^2, not the reference's arithmetic);
  * `TEMPORAL` (frozen-flow time series, fast.py:607-637): the layer screens, the bilinear shifts and
    the detector run on the GPU; its draws are always numpy's, in the reference's order (the
    series is sequential and tiny), so the same SEED reproduces the reference;
  * `FFTW` / `FFTW_THREADS` are accepted; the transform always has the arithmetic of the reference's FFTW branch
    (funcs.py:212-215), a warning says so once when `FFTW` is False (the reference's default, aotools ift2);
  * `GPU_DEVICES: [0, 1, ...]`: one process drives several GPUs (fast_amd/multi.py); under a multi-rank launcher
    (RANK / WORLD_SIZE in the environment) the iterations are sharded over the ranks instead (fast_amd/dist.py).
    Either way the result vector is identical to a single-GPU run.  No torch anywhere.
"""
import logging

import numpy

from . import _lib, conf, dist, fitsio, host, multi, rendezvous

logger = logging.getLogger(__name__)

# Host generator of parity mode: module-global like the reference's funcs._R (funcs.py:21),
# so a second Fast with SEED=None continues the stream.
_R = numpy.random.default_rng()


_BRANCH_WARNED = set()


def _warn_transform_branch(fftw):
    """Say once per process which of the reference's two transform branches is reproduced.  The reference's DEFAULT
    is `FFTW: False` (fast/conf.py:71) = `aotools.fouriertransform.ift2` (fast/funcs.py:216-218); its `FFTW: True`
    branch is fftshift -> unnormalised forward DFT -> fftshift (funcs.py:212-215).  The GPU path always has the
    FFTW-branch arithmetic (the only branch written out in the reference itself): for the same SEED a
    reference run with `FFTW: False` sees point-mirrored screens of the same statistics, not the same `_r`."""
    key = bool(fftw)
    if key in _BRANCH_WARNED:
        return
    _BRANCH_WARNED.add(key)
    if fftw:
        logger.info("FFTW flag: the FFT runs on the GPU with the arithmetic of the reference's FFTW branch (funcs.py:212-215)")
    else:
        logger.warning("FFTW is False: the reference would use aotools' ift2 (funcs.py:216-218); the GPU path computes the "
                       "FFTW-branch transform (funcs.py:212-215) -- same statistics, but not the same per-iteration values "
                       "as a reference run with FFTW False and the same SEED (set FFTW True there to compare seeds)")


class Fast():
    """Drop-in for `fast.Fast` on the Monte-Carlo path.  `params`: config-file name or dict."""

    def __init__(self, params):
        self.conf = conf.ConfigParser(params)
        self.params = p = self.conf.config
        self.Niter, self.Nchunks = p['NITER'], p['NCHUNKS']
        self.seed = p['SEED']
        self.temporal, self.dt = p['TEMPORAL'], p['DT']
        if self.seed != None:
            self.set_seed(self.seed)
        self.init_logging()

        # (the size limit belongs to the GPU kernels: a run that has opted into the numpy path and gets it has none, like the reference)
        host_only = False
        if p['GPU_FALLBACK']:
            from . import hostpath
            host_only = hostpath.unavailable_reason() is not None
        prob = host.build_problem(p, size_limit=not host_only)          # raises the reference's config Exceptions
        self._prob = prob
        self.Niter_per_chunk = prob.M
        atm, pup = prob.atm, prob.pup
        # attributes of the reference object (fast.py:49-64 and the init_* methods)
        for k in ("zenith_correction", "h", "cn2", "L", "dtheta", "paa", "wind_dir", "wind_vector", "wind_speed",
                  "r0", "theta0", "tau0", "rytov_variance", "r0_los", "theta0_los", "tau0_los", "rytov_variance_los"):
            setattr(self, k, getattr(atm, k))
        self.L0, self.l0 = p['L0'], p['l0']
        self.power, self.wvl, self.k = p['POWER'], p['WVL'], prob.k
        self.D_ground, self.obsc_ground, self.D_sat, self.obsc_sat = p['D_GROUND'], p['OBSC_GROUND'], p['D_SAT'], p['OBSC_SAT']
        self.dx, self.Npxls, self.Npxls_pup = prob.dx, prob.N, prob.Np
        self.subharmonics = prob.subharm
        self.ao_mode, self.Dsubap, self.tloop, self.texp = prob.ao_mode, prob.d_wfs, p['TLOOP'], p['TEXP']
        self.Zmax, self.alias, self.noise, self.modal, self.modal_mult = prob.zmax, p['ALIAS'], p['NOISE'], prob.modal, prob.modal_mult
        self.dx_sat, self.pupil_sat = pup.dx_sat, pup.pupil_sat           # (pupil, pupil_mode, pupil_filter: properties below)
        self.pupil_mode_sat, self.W0, self.W0_sat = pup.pupil_mode_sat, pup.W0, pup.W0_sat
        self.pup_coords = pup.pup_coords
        self.link_budget, self.diffraction_limit = prob.link_budget, prob.diffraction_limit
        self.logamp = numpy.zeros((self.Niter))
        # bookkeeping names of the reference object that a caller may read (fast.py:78-80, 106, 245-257, 528-531)
        self.F0 = numpy.inf
        self.fftw, self.nthreads, self.fftw_objs = p['FFTW'], p['FFTW_THREADS'], None
        self.shifts = self.shifts_sh = None
        if hasattr(atm, "wind_correction"):
            self.wind_correction = atm.wind_correction

        self.precision = p['GPU_PRECISION']
        self.rng_mode = p['GPU_RNG']
        if self.precision not in ('f64', 'f32'):
            raise Exception("GPU_PRECISION must be 'f64' or 'f32'")
        if self.rng_mode not in ('device', 'host', 'numpy'):
            raise Exception("GPU_RNG must be 'device', 'host' or 'numpy'")
        if p['GPU_RNG_PRECISION'] not in ('auto', 'f32', 'f64'):
            raise Exception("GPU_RNG_PRECISION must be 'auto', 'f32' or 'f64'")
        devs = p['GPU_DEVICES']
        if devs is not None:
            devs = [int(d) for d in (devs if isinstance(devs, (list, tuple, numpy.ndarray)) else [devs])]
            if not devs:
                raise Exception("GPU_DEVICES must name at least one device")
            if p['GPU_DEVICE'] is not None and int(p['GPU_DEVICE']) != devs[0]:
                raise Exception("GPU_DEVICE and GPU_DEVICES disagree: give one of them")
        # The backend: libfastmc.so on a GPU -- or, ONLY when the caller opted in with GPU_FALLBACK and the library or a device is
        # missing, numpy (fast_amd/hostpath.py), announced the way the reference announces its own fall-back (fast/fast.py:107-110)
        self._be, self.backend = _lib, 'gpu'
        if p['GPU_FALLBACK']:
            from . import hostpath
            why = hostpath.unavailable_reason()
            if why is not None:
                logger.warning(f"{why}, falling back to numpy (GPU_FALLBACK)")
                self._be, self.backend = hostpath, 'host'
                if self.rng_mode != 'host':
                    logger.warning(f"GPU_RNG '{self.rng_mode}' needs a device: the draws are numpy's, in the reference's order (GPU_RNG 'host')")
                    self.rng_mode = 'host'
                self.precision = 'f64'
        if self.backend == 'host':
            self.device, self.devices = -1, [-1]
            _warn_transform_branch(p['FFTW'])
            self._group = multi.DeviceGroup(self.Npxls, self.Npxls_pup, 'f64', [-1], exchange="host", factory=lambda d: self._be.Handle(self.Npxls, self.Npxls_pup))
        else:
            self.device = (_lib.default_device() if p['GPU_DEVICE'] is None else int(p['GPU_DEVICE'])) if devs is None else devs[0]
            self.devices = [self.device] if devs is None else devs
            _warn_transform_branch(p['FFTW'])
            # one handle per device, one thread each (fast_amd/multi.py); the first handle also serves everything that is
            # not sharded (statistics, histogram of the assembled vector, TEMPORAL and host-generator modes)
            self._group = multi.DeviceGroup(self.Npxls, self.Npxls_pup, self.precision, self.devices)
        self._handle = self._group.handles[0]
        self.precision = getattr(self._handle, 'precision', self.precision)     # what the handle computes in (see _lib.Handle)
        if p['GPU_BATCH']:
            self._group.set_batch(p['GPU_BATCH'])
        # the generator at the reference's precision unless the caller chose the float32 pipeline or the float32 draw
        self.rng_precision = p['GPU_RNG_PRECISION'] if p['GPU_RNG_PRECISION'] != 'auto' else ('f64' if self.precision == 'f64' else 'f32')
        self._group.each(lambda h, i: h.set_rng_precision(self.rng_precision))
        if p['GPU_KERNELS'] != 'auto':
            if p['GPU_KERNELS'] not in _lib.KERNEL_PATHS:
                raise Exception("GPU_KERNELS must be 'auto', 'wave', 'lanes50', 'chirpz' or 'direct'")
            self._group.each(lambda h, i: h.kernel_path(_lib.KERNEL_PATHS[p['GPU_KERNELS']]))
        self.compute_powerspec()
        # (every multiple of 64 from 192 to 4096 has the packed sub-rows for device-generator runs, whatever family holds its other rows)
        packed_sub_rows = self.Npxls % 64 == 0 and 192 <= self.Npxls <= 4096 and self.rng_mode == 'device'
        if self._handle.kernel_path() not in (1, 3) and self.Npxls >= 128 and not self.temporal and not packed_sub_rows:
            below = [n for n in host.ROUND_UP_SIZES if n <= self.Npxls][-1:]
            above = [n for n in host.ROUND_UP_SIZES if n >= self.Npxls][:1]
            near = ', '.join(str(n) for n in below + above)
            if self._handle.kernel_path() == 2:
                logger.info(f"NPXLS = {self.Npxls} is not 64 P: chirp-z kernels (about 3x the work per row of the plain FFT "
                            f"kernels); nearest plain sizes: {near}")
            else:
                logger.warning(f"NPXLS = {self.Npxls} runs on the direct O(N^2 Np) kernels (about 10x slower than the "
                               f"FFT kernels); nearest fast grid sizes: {near}")
        self._group.set_pupil(prob.W, pup.crop_lo, self.dx)
        if self.subharmonics:
            self._group.set_subharm(self.powerspec_subharm, self._sh_fx, self._sh_fy, self._sh_df)

    # ------------------------------------------------------------------ init pieces
    def init_logging(self):
        logging.basicConfig(filename=self.params['LOGFILE'], level=logging.getLevelName(self.params['LOGLEVEL']),
                            format="[%(levelname)s] %(name)s.%(funcName)s | %(message)s")

    def set_seed(self, seed):
        global _R
        _R = numpy.random.default_rng(seed)

    def compute_powerspec(self, per_layer=None):
        """AO-residual phase PSD and its Simpson integrals on the GPU (replaces fast.py:445-492).  The spectrum is
        evaluated on every device of the object and STAYS there as the handle's colouring tables
        (fastmc_powerspec_set): no N x N grid crosses PCIe unless `powerspec`, `logamp_powerspec`, `lf_mask` /
        `hf_mask` are read (fetched on first use) or the (L, N, N) per-layer grids are needed (TEMPORAL,
        `powerspec_per_layer`)."""
        logger.info("Computing (residual) phase power spectra")
        if per_layer is None:
            per_layer = bool(self.temporal)
        prob, p, atm = self._prob, self.params, self._prob.atm
        args = (prob.dx, prob.wvl, p['L0'], p['l0'], prob.ao_mode, p['ALIAS'], p['NOISE'], prob.d_wfs, p['TLOOP'], p['TEXP'],
                atm.dtheta, atm.cn2, atm.h, atm.wind_vector, prob.pup.pupil_filter, prob.simpson_w)
        kw = dict(lf_mask=None, modal=prob.modal, modal_mult=prob.modal_mult, zmax=prob.zmax, D_ground=p['D_GROUND'])
        outs = self._group.each(lambda h, i: self._be.powerspec_set(h, prob.df, *args, pupil_filter_token=prob.pup.token, **kw))
        out = outs[0]
        self._grids = {}
        for k in ("aniso_servo_error", "alias_error", "noise_error", "fitting_error", "phs_var", "logamp_var", "phs_var_weights"):
            setattr(self, k, out[k])
        self.powerspec_kernel_ms = out["kernel_ms"]
        self._per_layer = None
        if per_layer:
            self._per_layer = self._fetch_per_layer()
        if self.subharmonics:
            self.powerspec_subharm, self._sh_fx, self._sh_fy, self._sh_df, sh = host.subharm_spectrum(prob)
            # the bookkeeping of fast.py:494-526, as the reference leaves it on the object
            self.powerspec_subharm_per_layer, self.lf_mask_subharm = sh.per_layer, sh.lf_mask
            self.turb_lo, self.G_ao_lo, self.alias_subharm, self.noise_subharm = sh.turb, sh.G, sh.alias, sh.noise
            self.phs_var_subharm, self.phs_var_weights_sh = sh.phs_var, sh.phs_var_weights
        else:
            self.powerspec_subharm = self.phs_var_subharm = self.phs_var_weights_sh = None
        self.temporal_powerspec = None
        self.temporal_logamp_powerspec = prob.temporal.logamp_powerspec if self.temporal else None
        if self.temporal:
            self.pixel_shifts = prob.temporal.pixel_shifts

    def _ps_args(self):
        prob, p, atm = self._prob, self.params, self._prob.atm
        args = (prob.dx, prob.wvl, p['L0'], p['l0'], prob.ao_mode, p['ALIAS'], p['NOISE'], prob.d_wfs, p['TLOOP'], p['TEXP'],
                atm.dtheta, atm.cn2, atm.h, atm.wind_vector, prob.pup.pupil_filter, prob.simpson_w)
        kw = dict(lf_mask=None, modal=prob.modal, modal_mult=prob.modal_mult, zmax=prob.zmax, D_ground=p['D_GROUND'])
        return args, kw

    def _fetch_per_layer(self):
        """(L, N, N) per-layer grids by the stand-alone evaluation: the handles' colouring tables, the cached grids and the
        scalars on the object are left alone (a spectrum installed through the `powerspec` setter stays installed)."""
        args, kw = self._ps_args()
        pl = self._be.powerspec(self._prob.N, *args, per_layer=True, device=self.device, **kw)["powerspec_per_layer"]
        pl.flags.writeable = False
        return pl

    def _grid(self, which):
        """Host copy of a grid kept on the device, read-only: the device tables are what run() uses, so an in-place edit
        of the copy would be silently ignored -- replace the spectrum through the `powerspec` setter instead."""
        if which not in self._grids:
            g = self._handle.powerspec_get(which)
            g.flags.writeable = False
            self._grids[which] = g
        return self._grids[which]

    # ---- pupil weights: plain attributes of the reference object, read where they are used -- `pupil * pupil_mode` in every
    # compute_detector (fast.py:647-649), `pupil_filter` in compute_powerspec (fast.py:488-492) -- so a caller may replace them
    # after construction.  Here the weights live on the devices: a replacement is uploaded; a new pupil_filter takes effect at
    # the next compute_powerspec(), as in the reference.  (The arrays of a geometry are shared between objects and read-only:
    # the first replacement gives this object a description of its own.)
    def _own_pupil(self):
        import copy
        if self.temporal:
            # (the frozen-flow set-up -- high-resolution pupil-filter spline, temporal log-amplitude spectrum -- was built from the
            # computed pupil at construction, as the reference builds it in __init__: fast.py:394-405, 538-587)
            logger.warning("pupil weights replaced on a TEMPORAL object: the temporal log-amplitude spectrum keeps the pupil it was built with")
        if not getattr(self, "_pupil_owned", False):
            self._prob.pup = copy.copy(self._prob.pup)
            self._pupil_owned = True
        return self._prob.pup

    def _set_weights(self, pupil=None, mode=None):
        pup = self._own_pupil()
        Np = self.Npxls_pup
        for name, a in (("pupil", pupil), ("pupil_mode", mode)):
            if a is not None:
                a = numpy.array(a, dtype=float)
                if a.shape != (Np, Np):
                    raise ValueError(f"{name} must be ({Np}, {Np})")
                a.flags.writeable = False
                setattr(pup, name, a)
        self._prob.W = pup.pupil * pup.pupil_mode
        self._group.set_pupil(self._prob.W, pup.crop_lo, self.dx)

    pupil = property(lambda self: self._prob.pup.pupil, lambda self, v: self._set_weights(pupil=v),
                     doc="(Np, Np) aperture on the pupil grid (fast.py:379-392); assignable")
    pupil_mode = property(lambda self: self._prob.pup.pupil_mode, lambda self, v: self._set_weights(mode=v),
                          doc="(Np, Np) fibre / launch mode on the pupil grid; assignable")

    @property
    def pupil_filter(self):
        """(N, N) |FT(pupil x mode)|^2-type filter of the log-amplitude spectrum (funcs.py:300-312); assignable -- an (N, N)
        array or a scalar -- and read by the next compute_powerspec()."""
        return self._prob.pup.pupil_filter

    @pupil_filter.setter
    def pupil_filter(self, value):
        pup = self._own_pupil()
        N = self.Npxls
        a = numpy.array(numpy.broadcast_to(numpy.asarray(value, dtype=float), (N, N)))
        a.flags.writeable = False
        pup.pupil_filter = a
        host._PUPIL_TOKEN += 1
        pup.token = host._PUPIL_TOKEN          # a new array: the device-side cache must not serve the old one

    @property
    def powerspec(self):
        """(N, N) residual phase PSD, fft-shifted layout (fast.py:481); fetched from the GPU on first use."""
        return self._grid("powerspec")

    @powerspec.setter
    def powerspec(self, value):
        """The reference reads `self.powerspec` in every chunk (fast.py:593-594), so a caller may replace it between
        construction and `run()`; here the new grid is uploaded to every device as the colouring tables."""
        ps = numpy.ascontiguousarray(value, dtype=float)
        if ps.shape != (self.Npxls, self.Npxls):
            raise ValueError(f"powerspec must be ({self.Npxls}, {self.Npxls})")
        self._group.set_spectrum(ps, self._prob.df)
        ps = ps.copy()
        ps.flags.writeable = False
        self._grids["powerspec"] = ps

    @property
    def logamp_powerspec(self):
        return self._grid("logamp_powerspec")

    @property
    def lf_mask(self):
        """mask_lf (ao_power_spectra.py:119-141), evaluated on the device; same dtype as the reference's."""
        if "lf_mask_typed" not in self._grids:
            m = self._grid("lf_mask")
            self._grids["lf_mask_typed"] = m if (self._prob.modal and self._prob.zmax is not None) else m.astype(numpy.int64)
        return self._grids["lf_mask_typed"]

    @property
    def hf_mask(self):
        return 1 - self.lf_mask

    # ------------------------------------------------------------------ Monte Carlo
    def run(self):
        """The chunk loop of fast.py:115-140 as GPU launches; returns a FastResult."""
        M, half = self.Niter_per_chunk, self.Niter_per_chunk // 2
        coherent = bool(self.params['COHERENT'])
        I = numpy.zeros((self.Nchunks, M), dtype=complex if coherent else float)
        if self.temporal:
            self._run_temporal(I, coherent)
        elif self.rng_mode == 'host':
            self._run_host_rng(I, coherent)
        elif self.rng_mode == 'numpy':
            self._run_numpy_rng(I, coherent)
        else:
            seed = self.seed if self.seed is not None else int(numpy.random.SeedSequence().generate_state(2, numpy.uint32).view(numpy.uint64)[0])
            n_real = self.Niter // 2
            tr = self._transport()
            if tr is not None and self.seed is None:
                # every rank must draw from the same generator: rank 0's entropy seed wins
                seed = int.from_bytes(tr.rdzv.broadcast(int(seed).to_bytes(8, "little"), src=0), "little")
            self._device_seed = seed
            if tr is None:
                # this process's devices (one or several): contiguous pieces, one thread per device
                out = self._group.run(seed, 0, n_real, None, float(self.logamp_var), coherent)
            else:
                # one process per GPU: this rank's contiguous realisation range, then one exchange (one synchronisation,
                # under a deadline; fast_amd/dist.py: step_sharded)
                out, _, self.exchange_info = dist.step_sharded(self._handle, tr, seed, 0, n_real, float(self.logamp_var), coherent)
            re, im = out[:n_real].reshape(self.Nchunks, half), out[n_real:].reshape(self.Nchunks, half)
            I[:, :half], I[:, half:] = re, im
            # the log-amplitudes the device drew, in iteration order (global iteration 2g+s)
            chi = self._handle.rng_logamp(seed, 0, self.Niter) * numpy.sqrt(self.logamp_var)
            la = numpy.empty((self.Nchunks, M))
            la[:, :half] = chi[0::2].reshape(self.Nchunks, half)
            la[:, half:] = chi[1::2].reshape(self.Nchunks, half)
            self.logamp[:] = la.ravel()
        self.random_iters = I[-1]
        self.timing = self._handle.last_timing()
        self.result = FastResult(I.flatten(), self.diffraction_limit)
        if self.temporal or self.rng_mode in ('host', 'numpy') or getattr(self, '_tr', None) is not None or self._group.world > 1:
            # a run made of several library calls: hand the assembled vector back so that histogram(),
            # result_stats() and the fast_amd.comms reductions see all of it, not the last chunk / shard
            self._handle.set_results(self.result._r)
        self.I = self.result.power
        logger.info(self.result)
        return self.result

    def _run_host_rng(self, I, coherent):
        """Parity mode: numpy draws in the reference's order (fast.py:123,593,600; funcs.py:352-365)."""
        M, half, N = self.Niter_per_chunk, self.Niter_per_chunk // 2, self.Npxls
        re = _R.normal(0, 1, size=(self.Niter,))
        _R.normal(0, 1, size=(self.Niter,))
        self.logamp[:] = re * numpy.sqrt(self.logamp_var)
        for i in range(self.Nchunks):
            if i == self.Nchunks - 1:
                # `phs` re-draws the last chunk from here on demand (the reference keeps its screens, fast.py:596-603)
                self._last_chunk_state = _R.bit_generator.state
            cr = _R.normal(0, 1, size=(half, N, N))
            ci = _R.normal(0, 1, size=(half, N, N))
            sr = si = None
            if self.subharmonics:
                sr = _R.normal(0, 1, size=(half, 3, 3, 3))
                si = _R.normal(0, 1, size=(half, 3, 3, 3))
            I[i] = self._handle.run_coeffs(cr, ci, self.logamp[i * M:(i + 1) * M], coherent, sr, si)

    def _run_numpy_rng(self, I, coherent):
        """The reference's numbers for the reference's SEED at GPU speed: numpy's own stream -- `funcs._R`, PCG64 + ziggurat, in the
        reference's order (fast.py:123, 593, 600; funcs.py:352-365) -- drawn ON THE DEVICE (fast_amd/csrc/fmc_npstream.h,
        fast_amd/npnormal.py).  The module generator `_R` is left exactly where the reference's would be after the run.  A chunk
        the device gives up on (it says so; never observed) is drawn by numpy itself; a numpy whose normal() is not the ziggurat
        restated here, or a generator that is not PCG64, makes the whole run fall back to `GPU_RNG: 'host'` with a warning."""
        from . import npnormal
        M, half, N = self.Niter_per_chunk, self.Niter_per_chunk // 2, self.Npxls
        h = self._handle
        try:
            sw = npnormal.state_words(_R.bit_generator)
            if self.precision != 'f64':
                raise RuntimeError("the numpy-stream generator feeds the float64 pipeline")
            la, after, ovf = h.npstream_logamp(sw, self.Niter, float(self.logamp_var))
            if ovf:
                raise RuntimeError(f"the device gave up on the log-amplitude draws (flags {ovf})")
        except RuntimeError as e:          # incl. _lib.FastMCError
            logger.warning(f"GPU_RNG 'numpy' is unavailable ({e}); drawing on the host (GPU_RNG 'host')")
            return self._run_host_rng(I, coherent)
        self.logamp[:] = la
        inc = sw[2:]
        sw = numpy.concatenate([after, inc])

        def rstate(words):       # the dict numpy takes for a PCG64 at these state words
            st = _R.bit_generator.state
            st["state"]["state"] = (int(words[1]) << 64) | int(words[0])
            return st          # (has_uint32 / uinteger -- a buffered 32-bit half-word -- stay as they are: normal() never touches them)
        c, last = 0, self.Nchunks - 1
        while c < self.Nchunks:
            if c == last:
                self._last_chunk_state = rstate(sw)           # `phs` re-draws the last chunk from here on demand
            n = (last - c) if c < last else 1
            out, after, bad = h.run_npstream(sw, n, half, c * M, coherent)
            good = n if bad < 0 else bad
            I[c:c + good] = out[:good]
            c += good
            sw = numpy.concatenate([after, inc])
            if bad >= 0:
                logger.warning(f"chunk {c}: the device gave up on numpy's stream; drawing this chunk with numpy")
                _R.bit_generator.state = rstate(sw)
                if c == last:
                    self._last_chunk_state = _R.bit_generator.state
                cr = _R.normal(0, 1, size=(half, N, N))
                ci = _R.normal(0, 1, size=(half, N, N))
                sr = si = None
                if self.subharmonics:
                    sr = _R.normal(0, 1, size=(half, 3, 3, 3))
                    si = _R.normal(0, 1, size=(half, 3, 3, 3))
                I[c] = h.run_coeffs(cr, ci, self.logamp[c * M:(c + 1) * M], coherent, sr, si)
                c += 1
                sw = npnormal.state_words(_R.bit_generator)
        _R.bit_generator.state = rstate(sw)

    def _run_temporal(self, I, coherent):
        """Frozen-flow series (fast.py:607-637; funcs.py:367-375): numpy draws in the reference's
        order; screens, bilinear shifts and detector on the GPU."""
        N, Np, M, prob = self.Npxls, self.Npxls_pup, self.Niter_per_chunk, self._prob
        tps = self.temporal_logamp_powerspec
        r = _R.normal(0, 1, size=(self.Niter,)) + 1j * _R.normal(0, 1, size=(self.Niter,))
        r *= numpy.sqrt(tps / tps.sum())
        series = numpy.fft.fftshift(numpy.fft.fft(numpy.fft.fftshift(r)))
        self.logamp[:] = (series.T * numpy.sqrt(self.logamp_var)).real
        # chunk 0: one real screen per layer from (L, N, N) coefficients (double=False)
        L = self.powerspec_per_layer.shape[0]
        cr = _R.normal(0, 1, size=(L, N, N))
        ci = _R.normal(0, 1, size=(L, N, N))
        # the L layer screens in ONE batched launch: the coefficients are coloured here exactly as the reference colours them
        # (rand *= sqrt(powerspec_per_layer), fast.py:610-612) and transformed with a unit spectrum (the `* df` of
        # funcs.py:213 is the handle's amplitude); window = the whole grid
        full = self._full_window_handle()
        amp = numpy.sqrt(self.powerspec_per_layer)
        scrns = full.screens_coeffs(cr * amp, ci * amp)[:L]
        self._handle.set_layer_screens(scrns)
        pup = prob.pup.pup_coords.astype(float)
        interp = pup[numpy.newaxis, :, numpy.newaxis, :] + self.pixel_shifts[:, :, :, numpy.newaxis]
        self.interp_coords = interp                      # fast.py:617 (chunk 0)
        for i in range(self.Nchunks):
            coord, shifts = host.temporal_coords(interp, N)
            I[i] = self._handle.temporal_chunk(coord[:, 0], coord[:, 1], shifts, self.logamp[i * M:(i + 1) * M], coherent)
            self._last_temporal = (coord[:, 0], coord[:, 1], shifts)
            interp = interp + self.pixel_shifts[:, :, -1, numpy.newaxis, numpy.newaxis]
            self.interp_coords = interp                  # advanced after every chunk, as fast.py:635 leaves it

    def _full_window_handle(self):
        """A handle whose window is the whole N x N grid with a unit spectrum (layer screens of TEMPORAL runs); kept on
        the object: every run() of a time series needs it again."""
        if getattr(self, "_full", None) is None:
            N = self.Npxls
            self._full = self._be.Handle(N, N, self.precision, self.device)
            self._full.set_pupil(numpy.ones((N, N)), 0, self.dx)
            self._full.set_spectrum(numpy.ones((N, N)), self._prob.df)
        return self._full

    def _transport(self):
        """The result exchange of a one-process-per-GPU run, or None.  GPU_SHARD: 'auto' (default) shards when a
        launcher started this process as one of several ranks (WORLD_SIZE > 1 with RANK / MASTER_ADDR / MASTER_PORT in
        the environment); True requires that; False never shards (every rank runs its own simulation, as the sweeps do)."""
        mode = self.params.get('GPU_SHARD', 'auto')
        if mode is False:
            return None
        if getattr(self, '_tr', None) is None:
            rdzv = rendezvous.from_env()
            if rdzv is None:
                if mode is True:
                    raise Exception("GPU_SHARD=True needs a multi-rank launch (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)")
                return None
            if self._group.world > 1:
                raise Exception("GPU_DEVICES with several devices inside a multi-rank launch: use one device per rank")
            self._tr = dist.make_transport(self._handle, rdzv)
        return self._tr

    def histogram(self, lo_db=-60.0, hi_db=10.0, nbins=4096):
        """Histogram of dB_rel of the last run, computed on the device."""
        return self._handle.histogram(lo_db, hi_db, nbins)

    def _psd_terms(self):
        """turb_powerspec, G_ao, alias_powerspec, noise_powerspec of fast.py:448-472, fetched from the GPU on first use."""
        if getattr(self, "_terms", None) is None:
            prob, p, atm = self._prob, self.params, self._prob.atm
            t = self._be.powerspec_terms(prob.N, prob.dx, prob.wvl, p['L0'], p['l0'], prob.ao_mode, p['ALIAS'], p['NOISE'],
                                     prob.d_wfs, p['TLOOP'], p['TEXP'], atm.dtheta, atm.cn2, atm.h, atm.wind_vector,
                                     prob.pup.pupil_filter, prob.simpson_w, lf_mask=None, modal=prob.modal,
                                     modal_mult=prob.modal_mult, zmax=prob.zmax, D_ground=p['D_GROUND'], device=self.device)
            noao = prob.ao_mode == 'NOAO'
            # the reference keeps plain scalars where a term is switched off (fast.py:464-465, 471-472; G_AO_PAOLA
            # returns 1 for NOAO, ao_power_spectra.py:235-236)
            if noao:
                t["G_ao"] = 1
            if noao or not p['ALIAS']:
                t["alias_powerspec"] = 0.
            if noao or not p['NOISE'] > 0:
                t["noise_powerspec"] = 0.
            self._terms = t
        return self._terms

    turb_powerspec = property(lambda self: self._psd_terms()["turb_powerspec"])
    G_ao = property(lambda self: self._psd_terms()["G_ao"])
    alias_powerspec = property(lambda self: self._psd_terms()["alias_powerspec"])
    noise_powerspec = property(lambda self: self._psd_terms()["noise_powerspec"])

    @property
    def powerspec_per_layer(self):
        """(L, N, N) residual PSD per turbulence layer (fast.py:478-479); fetched from the GPU on first use."""
        if self._per_layer is None:
            self._per_layer = self._fetch_per_layer()
        return self._per_layer

    @property
    def freq(self):
        """Angular spatial-frequency grids like the reference's `SpatialFrequencies` (fast.py:814-875):
        freq.main.{fx_axis, fy_axis, fx, fy, fabs, f, df}, the same names on `freq` itself, and
        freq.subharm when sub-harmonics are on.  Built on first use (three N x N arrays)."""
        if getattr(self, "_freq", None) is None:
            from types import SimpleNamespace
            prob = self._prob
            fx, fy, fabs = host.mesh(prob.axis)
            main = SimpleNamespace(fx_axis=prob.axis, fy_axis=prob.axis, f=prob.axis, df=prob.df, dfx=prob.df, dfy=prob.df,
                                   fx=fx, fy=fy, fabs=fabs, freq_per_layer=False)
            fr = SimpleNamespace(N=prob.N, dx=prob.dx, main=main, fx=fx, fy=fy, fabs=fabs, f=prob.axis, df=prob.df)
            if self.subharmonics:
                ax = host.subharm_axes(prob.N, prob.dx)
                sfx, sfy, sfabs = host.mesh(ax)
                fr.subharm = SimpleNamespace(fx_axis=ax, fy_axis=ax, f=ax, df=ax[..., 1] - ax[..., 0], fx=sfx, fy=sfy,
                                             fabs=sfabs, freq_per_layer=False)
            self._freq = fr
        return self._freq

    @property
    def phs(self):
        """Phase screens of the LAST chunk, (NITER/NCHUNKS, Np, Np), as `Fast.phs` holds them after `run()` in the
        reference (fast.py:596-603, 633).  Recomputed on the GPU when read: device-generator runs from the seed, host-generator
        runs by re-drawing the last chunk from the generator state kept at its start, TEMPORAL runs from the last chunk's
        sample coordinates and the layer screens still resident on the device."""
        if not hasattr(self, "result"):
            raise AttributeError("phs is available after run()")
        half = self.Niter_per_chunk // 2
        if self.temporal:
            return self._handle.temporal_phases(*self._last_temporal)
        if self.rng_mode in ('host', 'numpy'):
            N = self.Npxls
            rng = numpy.random.default_rng()
            rng.bit_generator.state = self._last_chunk_state
            cr = rng.normal(0, 1, size=(half, N, N))
            ci = rng.normal(0, 1, size=(half, N, N))
            sr = si = None
            if self.subharmonics:
                sr = rng.normal(0, 1, size=(half, 3, 3, 3))
                si = rng.normal(0, 1, size=(half, 3, 3, 3))
            return self._handle.screens_coeffs(cr, ci, sr, si)
        return self._handle.screens(self._device_seed, (self.Nchunks - 1) * half, half)

    def result_stats(self, thresholds_dB_rel=()):
        """Summary statistics of the last run reduced on the device (mean, scintillation index, fade
        probabilities below the given dB_rel thresholds): nothing per-iteration crosses PCIe."""
        return self._handle.result_stats([10 ** (t / 10) for t in thresholds_dB_rel])

    def compute_mean_irradiance(self, onaxis=True):
        """Analytic (non Monte-Carlo) mean coupled flux, fast.py:736-761; its N x N transforms on the GPU."""
        return host.mean_irradiance(self.powerspec, self._prob.W, self.dx, self._prob.df, self.diffraction_limit, onaxis,
                                    self.device, backend=self._be)

    def make_header(self, params):
        """Result-file header cards (fast.py:771-807)."""
        hdr = {}
        hdr['ZENITH'] = params['ZENITH_ANGLE']
        hdr['WVL'] = int(params['WVL'] * 1e9)
        hdr['OTRSCALE'] = str(params['L0']) if numpy.isinf(params['L0']) else params['L0']
        hdr['INRSCALE'] = params['l0']
        hdr['POWER'] = params['POWER']
        hdr['PAA'] = self.paa
        hdr['AO_MODE'] = self.ao_mode
        hdr['TLOOP'] = params['TLOOP']
        hdr['TEXP'] = params['TEXP']
        hdr['DSUBAP'] = params['DSUBAP']
        hdr['ALIAS'] = str(params['ALIAS'])
        hdr['NOISE'] = params['NOISE']
        hdr['D_GND'] = params['D_GROUND']
        hdr['OBSC_GND'] = params['OBSC_GROUND']
        hdr['D_SAT'] = params['D_SAT']
        hdr['OBSC_SAT'] = params['OBSC_SAT']
        hdr['AXICON'] = str(params['AXICON'])
        hdr['W0'] = self.W0
        hdr['L_SAT'] = self.L
        hdr['H_SAT'] = params['H_SAT']
        hdr['DX'] = self.dx
        hdr['NPXLS'] = self.Npxls
        hdr['NITER'] = self.Niter
        hdr['R0'] = self.r0
        hdr['THETA0'] = self.theta0
        hdr['TAU0'] = self.tau0
        hdr["DIFFLIM"] = self.diffraction_limit
        if self.seed != None:
            hdr["SEED"] = self.seed
        return hdr

    def save(self, fname, **kwargs):
        """Write `result.power` and the header as a FITS file (fast.py:809-812)."""
        logger.info(f"Saving results to {fname}")
        fitsio.writeto(fname, self.result.power, header=self.make_header(self.params), **kwargs)

    def calc_zenith_correction(self, zenith_angle):
        return 1 / numpy.cos(numpy.radians(zenith_angle))


def load(fname):
    """Result file -> FastResult (fast.py:998-1002; data are stored in watts)."""
    hdr, data = fitsio.read(fname)
    return FastResult(data / hdr['DIFFLIM'], hdr['DIFFLIM'], header=hdr)


class FastResult():
    '''
    Per-iteration coupled power relative to the diffraction limit, with lazy unit
    conversions (same attributes as fast/fast.py:931-994):

        dB_rel, dB_abs, dBm, power [W], scintillation_index, avg_power_*
    '''
    def __init__(self, random_iters, diffraction_limit, header=None):
        self._r = random_iters
        self._dl = diffraction_limit
        if header != None:
            self.hdr = header

    dB_rel = property(lambda s: 10 * numpy.log10(s._r))
    dB_abs = property(lambda s: 10 * numpy.log10(s._r * s._dl))
    dBm = property(lambda s: 10 * numpy.log10(s._r * s._dl / 1e-3))
    power = property(lambda s: s._dl * s._r)
    scintillation_index = property(lambda s: (s._r / s._r.mean()).var())
    avg_power_W = property(lambda s: s.power.mean())
    avg_power_dBm = property(lambda s: 10 * numpy.log10(s.avg_power_W / 1e-3))
    avg_power_dB_rel = property(lambda s: 10 * numpy.log10((s.power / s._dl).mean()))
    avg_power_dB_abs = property(lambda s: 10 * numpy.log10(s.avg_power_W))

    def __str__(self):
        return ("FAST result statistics:\n"
                f"            Avg. power (W): {self.avg_power_W}\n"
                f"            Avg. power (dBm): {self.avg_power_dBm}\n"
                f"            Avg. power (dB_rel): {self.avg_power_dB_rel}\n"
                f"            Avg. power (dB_abs): {self.avg_power_dB_abs}\n"
                f"            Scintillation index: {self.scintillation_index}\n        ")
