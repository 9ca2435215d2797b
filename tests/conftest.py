import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def params_from_json(s):
    """Inverse of tools/capture_golden/capture.py:params_to_json."""
    raw = json.loads(str(s))
    out = {}
    for k, v in raw.items():
        if isinstance(v, dict) and "__ndarray__" in v:
            out[k] = np.array(v["__ndarray__"], dtype=float)
        elif isinstance(v, dict) and "__float__" in v:
            out[k] = float(v["__float__"])
        else:
            out[k] = v
    out.setdefault("GPU_ROUND_NPXLS", False)     # captured configs are compared with the reference on ITS grid
    return out


E2E_CASES = ["ao_alias", "noao", "noao_L0", "tt", "noise", "noalias_noise", "modal", "modal_zmax",
             "lgsao", "subharm", "subharm_ao", "coherent", "down", "obsc", "axicon", "w0fixed",
             "lsat_aniso", "oddNp", "autosize", "oddN", "npxls100", "npxls150", "npxls200"]


@pytest.fixture(scope="session")
def golden():
    return load_golden
