"""`import fast` -> the MI355X implementation of FAST's Monte-Carlo path (package `fast_amd`).

A user of ojdf/fast keeps `fast.Fast(config).run()`, `fast.FastResult`, `fast.conf`,
`fast.turbulence_models`, FITS `save` / `load`, and the link metrics of `fast.comms` that reduce the power
vector (fade_prob, fade_dur, ber_ook, sep_qam, ber_qam, Q), and `fast.funcs.make_phase_fft` /
`make_phase_subharm`; the symbol simulator, GMI and orbit
tools are not provided here -- see DESIGN.md section 6.
"""
from fast_amd import Fast, FastResult, FastMCError, load, conf, turbulence_models, comms, funcs  # noqa: F401
from fast_amd import __version__  # noqa: F401

# `from fast.comms import ber_ook`, `import fast.turbulence_models` ... resolve to the fast_amd modules
import sys as _sys
for _name, _mod in (("comms", comms), ("conf", conf), ("turbulence_models", turbulence_models), ("funcs", funcs)):
    _sys.modules.setdefault(__name__ + "." + _name, _mod)
del _sys, _name, _mod
