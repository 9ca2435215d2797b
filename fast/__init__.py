"""`import fast` -> the MI355X implementation of FAST's Monte-Carlo path (package `fast_amd`).

A user of ojdf/fast keeps `fast.Fast(config).run()`, `fast.FastResult`, `fast.conf`,
`fast.turbulence_models`; everything outside that path (comms, orbit tools, FITS I/O) is not
provided here -- see DESIGN.md section 6.
"""
from fast_amd import Fast, FastResult, FastMCError, load, conf, turbulence_models  # noqa: F401
from fast_amd import __version__  # noqa: F401
