"""Parity of the HIP solve (K1) through the C-ABI against the reference's golden
vectors and the oracle.  Tolerance: 1e-6 norm-relative on the coefficients
(BASELINE.json north_star); observed errors are ~1e-12 and the tighter bound
1e-9 is asserted too so a regression is visible long before the gate."""
import numpy as np
import pytest

from conftest import norm_rel

pytestmark = pytest.mark.gpu

TOL = 1e-6        # north-star gate
TIGHT = 1e-9      # what the algorithm actually delivers (regression tripwire)

SINGLE = ["cfg1", "testdata", "m1", "m2", "t0quirk", "path49"]
BATCH = ["cfg2", "cfg2s", "m20", "m3"]


@pytest.mark.parametrize("name", SINGLE)
def test_single_trajectory_golden(ctx7, golden, name):
    wp, t, ref = golden[name + "_wp"], golden[name + "_t"], golden[name + "_coef"]
    coef, dur, status = ctx7.solve_batch(wp[None], t[None])
    assert status[0] == 0
    err = norm_rel(coef[0], ref)
    assert err <= TOL, err
    assert err <= (1e-7 if name == "path49" else TIGHT), err
    np.testing.assert_array_equal(dur[0], golden[name + "_dur"])


@pytest.mark.parametrize("name", BATCH)
def test_batch_golden(ctx7, golden, name):
    wp, t, ref = golden[name + "_wp"], golden[name + "_t"], golden[name + "_coef"]
    coef, dur, status = ctx7.solve_batch(wp, t)
    assert (status == 0).all()
    err = norm_rel(coef, ref)
    assert err <= TIGHT, err
    rdur = golden[name + "_dur"]
    np.testing.assert_array_equal(dur, rdur if rdur.ndim == 2 else np.broadcast_to(rdur, dur.shape))
