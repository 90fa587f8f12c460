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
    """Golden vectors generated from the reference by oracle/make_golden.py (plain arrays only)."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), "finite/non-finite pattern differs"
    if not fin.any():
        return 0.0
    return float(np.max(np.abs(got[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1e-300)))


def ensure_library_built():
    """Build libmcd_hip.so in-tree when it is missing or older than its sources (hipcc cross-compiles
    for gfx950 without a GPU).  Returns the path."""
    import subprocess
    csrc = os.path.join(ROOT, "mcmc_dynamics_amd", "csrc")
    lib = os.path.join(ROOT, "mcmc_dynamics_amd", "libmcd_hip.so")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))] + \
           [os.path.join(ROOT, "include", "mcd.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(f) > os.path.getmtime(lib) for f in srcs):
        subprocess.run(["make", "-C", csrc, "-j4"], check=True, capture_output=True)
    return lib


@pytest.fixture(scope="session")
def built_library():
    return ensure_library_built()
