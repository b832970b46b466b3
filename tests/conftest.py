import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """fp64 vectors computed by the reference itself (tests/golden/make_golden.py)."""
    return np.load(os.path.join(GOLDEN_DIR, "ref_golden.npz"))


def norm_rel(got, ref):
    """SURVEY.md Appendix C parity metric: per trajectory and axis,
    max_k |c - c_ref| / max_k |c_ref| over all segments; returns the worst one.
    got/ref: [N, M, 4, nc] or [M, 4, nc]."""
    got = np.asarray(got)
    ref = np.asarray(ref)
    if got.ndim == 3:
        got, ref = got[None], ref[None]
    num = np.abs(got - ref).max(axis=(1, 3))
    den = np.abs(ref).max(axis=(1, 3))
    den = np.where(den == 0, 1.0, den)
    return float((num / den).max())


@pytest.fixture(scope="session")
def ctx7():
    from drone_path_planning_python_amd import Context
    c = Context(device_id=0, order=7, max_segments=4096)
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx9():
    from drone_path_planning_python_amd import Context
    c = Context(device_id=0, order=9, max_segments=4096)
    yield c
    c.close()
