"""CPU: the wave-FFT pipeline of fmc_wavefft.h, executed lane by lane on the host, against a
naive long-double DFT with the reference's fftshift semantics (fast/funcs.py:213-215)."""
import os
import subprocess

from conftest import ROOT


def test_lane_emulation_matches_naive_dft():
    exe = os.path.join(ROOT, "fast_amd", "emu_wavefft")
    subprocess.run(["make", "-C", os.path.join(ROOT, "fast_amd", "csrc"), "emu"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "EMU OK" in out.stdout and "FAIL" not in out.stdout


def test_lane_emulation_under_sanitizers():
    """Same program built with -fsanitize=address,undefined: every LDS-image index of the pipeline
    (exchange buffers, tables) stays in bounds for all P, NS, window positions exercised."""
    exe = os.path.join(ROOT, "fast_amd", "emu_wavefft_asan")
    subprocess.run(["make", "-C", os.path.join(ROOT, "fast_amd", "csrc"), "emu-asan"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "EMU OK" in out.stdout
