"""fast_amd -- the Monte-Carlo hot path of FAST (ojdf/fast) on AMD MI355X (gfx950).

Public surface = the reference's for this path: `Fast`, `FastResult`, `conf`, `turbulence_models`.
Compute lives in libfastmc.so (hand-written HIP behind the C-ABI of include/fastmc.h),
loaded with ctypes; there is no CPU fallback.
"""
from .fast import Fast, FastResult, load
from . import conf
from . import turbulence_models
from . import host
from . import _lib
from . import comms
from . import funcs
from . import dist
from . import multi
from . import rendezvous
from . import sweep
from ._lib import FastMCError

__version__ = "0.2.0"
__all__ = ["Fast", "FastResult", "load", "conf", "turbulence_models", "host", "comms", "funcs", "dist", "multi",
           "rendezvous", "sweep", "FastMCError"]
