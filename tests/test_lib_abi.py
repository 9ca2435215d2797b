"""CPU: libfastmc.so loads, exports every symbol include/fastmc.h declares, and refuses to
compute without a GPU (no CPU fallback)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT
import fast_amd
from fast_amd import _lib


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "fast_amd", "csrc"), "all"], check=True)
    return _lib.lib()


def test_header_symbols_are_exported(built):
    hdr = open(os.path.join(ROOT, "include", "fastmc.h")).read()
    declared = set(re.findall(r"\b(fastmc_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"fastmc_ctx", "fastmc_t"}
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(built, name), name
    assert built.fastmc_version() == 300


def test_no_cpu_fallback_without_gpu(built):
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(fast_amd.FastMCError, match="no HIP device|no CPU fallback"):
        _lib.Handle(64, 22)
    with pytest.raises(fast_amd.FastMCError):
        fast_amd.Fast({"NPXLS": 64, "DX": 0.01, "NITER": 4, "NCHUNKS": 1, "D_GROUND": 0.2, "LOGLEVEL": "ERROR"})


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fast_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def _build_c_client(tmp_path):
    import subprocess
    exe = str(tmp_path / "c_smoke")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "c_smoke.c"), "-o", exe, "-L", os.path.join(ROOT, "fast_amd"), "-lfastmc",
           "-Wl,-rpath," + os.path.join(ROOT, "fast_amd"), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_links(built, tmp_path):
    """include/fastmc.h compiles as C99 and a C client links against libfastmc.so (no compute here)."""
    _build_c_client(tmp_path)


@pytest.mark.gpu
def test_plain_c_client_runs(built, tmp_path):
    import subprocess
    r = subprocess.run([_build_c_client(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C-ABI OK" in r.stdout and "split_maxdiff=0" in r.stdout
    # kernel families and the one-process communicator calls from plain C
    assert "path(200)=3" in r.stdout and ("comm rc=0 world=1 rank=0 gathered_diff=0 hist=800" in r.stdout or "comm rc=-" in r.stdout)
