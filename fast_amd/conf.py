"""Configuration: dict or .py file defining `p`, defaults filled in with a warning.
Mirrors fast/conf.py:11-116 of the reference (same keys, same defaults), plus the GPU keys."""
import importlib.util
import logging

import numpy

logger = logging.getLogger(__name__)

DEFAULTS = {
    # simulation
    'NPXLS': 'auto', 'DX': 'auto', 'NITER': 1000, 'SUBHARM': False, 'FFTW': False, 'FFTW_THREADS': 1,
    'NCHUNKS': 10, 'TEMPORAL': False, 'DT': 0.001, 'LOGFILE': None, 'LOGLEVEL': "INFO", 'SEED': None,
    # transmitter / receiver
    'W0': "opt", 'D_GROUND': 1.0, 'OBSC_GROUND': 0, 'D_SAT': 0.1, 'OBSC_SAT': 0, 'WVL': 1550e-9,
    'AXICON': False, 'POWER': 1, 'SMF': True,
    # turbulence and link
    'H_SAT': 36e6, 'L_SAT': None, 'H_TURB': numpy.array([0, 10e3]), 'CN2_TURB': numpy.array([100e-15, 100e-15]),
    'WIND_SPD': numpy.array([10, 10]), 'WIND_DIR': numpy.array([90., 0.]), 'L0': numpy.inf, 'l0': 1e-06,
    'ZENITH_ANGLE': 0, 'PROP_DIR': 'up', 'DTHETA': [4, 0], 'TRANSMISSION': 1,
    # adaptive optics
    'AO_MODE': 'AO', 'DSUBAP': 0.02, 'TLOOP': 0.001, 'TEXP': 0.001, 'ALIAS': True, 'NOISE': 0.0,
    'MODAL': False, 'MODAL_MULT': 1, 'ZMAX': None,
    # comms
    'COHERENT': False, 'MODULATION': None, 'EsN0': None,
}

# Keys of the MI355X backend (not in the reference).  Filled silently.
GPU_DEFAULTS = {
    'GPU_PRECISION': 'f64',   # 'f64' (complex128 pipeline, reference precision) or 'f32'
    'GPU_FALLBACK': False,    # True: when libfastmc.so or a GPU is missing, warn (as fast/fast.py:107-110 does for pyfftw) and compute
                              # on the host with numpy (fast_amd/hostpath.py); default: raise, nothing ever falls back silently
    'GPU_RNG': 'device',      # 'device': Philox on the GPU; 'host': numpy draws, reference order (parity mode);
                              # 'numpy': the SAME stream as 'host' (the reference's numbers for its SEED) drawn on the GPU
    'GPU_RNG_PRECISION': 'auto',  # device generator: 'auto' = the pipeline's precision (GPU_PRECISION; 'f64' unless that says otherwise);
                              # 'f64': 53-bit normals and float64 colouring like the reference's funcs.py:352-356 / fast.py:594,
                              # fused into the row kernels; 'f32' (opt-in shortcut, ~1.7x the rate at 1024^2): 24-bit uniforms,
                              # hardware float32 Box-Muller, float32 colouring -- NOT the reference's arithmetic
    'GPU_DEVICE': None,       # HIP device index; None -> LOCAL_RANK or 0
    'GPU_DEVICES': None,      # list of HIP device indices driven by THIS process (one thread each, fast_amd/multi.py);
                              # None -> [GPU_DEVICE]
    'GPU_BATCH': 0,           # realisations in flight per launch (0 = library default)
    'GPU_KERNELS': 'auto',    # kernel family: 'auto' (wave FFT where NPXLS = 64 P, 50-lane FFT where NPXLS = 50 P S, chirp-z for
                              # other sizes, direct for tiny grids / huge windows) | 'wave' | 'lanes50' | 'chirpz' | 'direct'
                              # (O(N^2 Np) cross-check)
    'GPU_ROUND_NPXLS': False, # with NPXLS 'auto': False (default since round 6) = the reference's own auto-sized grid (fast/fast.py:167-189:
                               # 164 for the shipped example -- a drop-in keeps the user's grid; the chirp-z kernels serve any size);
                               # True = round it up to the next fast-kernel size (opt-in, logged); 'auto' = round when nothing ties the
                               # run to the reference's exact grid (GPU_RNG 'device', not TEMPORAL: the default of rounds 3-5)
    'GPU_SHARD': 'auto',      # shard iterations over the ranks of a multi-process launch (RANK / WORLD_SIZE in the
                              # environment; fast_amd/rendezvous.py): 'auto' | True | False
}


def _read_py_config(path):
    """A config file is a Python module that defines the dict `p` (fast/conf.py:92-101)."""
    if not path.endswith(".py"):
        raise Exception("Require .py config file")
    spec = importlib.util.spec_from_file_location("", path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module.p


class ConfigParser:
    """`.config`: the user's dict (or the `p` of a .py file), completed in place: reference keys with a
    warning per missing key (fast/conf.py:103-116), GPU keys silently.  `.fname`, `.defaults` as in the reference."""

    def __init__(self, fname_or_dict):
        if isinstance(fname_or_dict, dict):
            self.fname, self.config = None, fname_or_dict
        elif isinstance(fname_or_dict, str):
            self.fname, self.config = fname_or_dict, _read_py_config(fname_or_dict)
        else:
            raise Exception("Either config file name or params dict required")
        self.defaults = DEFAULTS
        missing = [k for k in DEFAULTS if k not in self.config]
        for key in missing:
            logger.warning(f"Config parameter {key} not defined in {self.fname}, setting default value of {DEFAULTS[key]}")
            self.config[key] = DEFAULTS[key]
        for key, val in GPU_DEFAULTS.items():
            self.config.setdefault(key, val)
