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
