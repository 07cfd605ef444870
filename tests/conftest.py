import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "motif-learn_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Vectors captured from the reference by oracle/make_golden.py."""
    with np.load(os.path.join(ROOT, "tests", "golden", "zps_golden.npz")) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def golden_high():
    """Reference outputs at n_max 28 / 32 / 36 on structured inputs (oracle/make_golden_high_orders.py)."""
    with np.load(os.path.join(ROOT, "tests", "golden", "zps_high_golden.npz")) as f:
        return {k: f[k] for k in f.files}


def rel_close(got, ref, rtol=1e-6, atol_scale=1e-12):
    """Parity criterion of SURVEY 8c: elementwise rtol=1e-6 with an absolute floor of
    1e-12 * max|ref| that only matters for moments that cancel to ~0."""
    ref = np.asarray(ref)
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol_scale * np.abs(ref).max())
