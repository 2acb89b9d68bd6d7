"""The C leg of the oracle (oracle/mivi_oracle.c, the bench's cpu_baseline "port") against the numpy leg."""
import os

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import oracle as O
from tests.helpers import SEED, make_family


@pytest.fixture(scope="module")
def clib():
    if not os.path.exists(CO.PATH):
        import __graft_entry__ as g
        g.build()
    return CO.load()


@pytest.mark.parametrize("family", [O.MEANFIELD, O.FULLRANK])
@pytest.mark.parametrize("ent", range(5))
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 3e-5)])
def test_c_oracle_matches_numpy_oracle(clib, family, ent, dtype, tol):
    rng = np.random.default_rng(family * 10 + ent)
    d, M = 37, 21
    _, q = make_family(rng, d, family)
    tm, ts = rng.normal(size=d), rng.uniform(0.5, 2.0, size=d)
    eps = O.philox_normal(SEED, 4, d, 0, M, f64=(dtype == np.float64))
    ref = O.estimate_gradient(O.destructure(q), d, family, O.DiagNormalTarget(tm, ts), eps, ent)
    v, g = CO.estimate_gradient(clib, dtype, family, d, M, O.destructure(q), eps, tm, ts, ent)
    assert abs(v - ref["value"]) <= tol * abs(ref["value"])
    assert np.linalg.norm(g - ref["grad"]) <= tol * max(1.0, np.linalg.norm(ref["grad"]))


def test_c_oracle_eps_stream(clib):
    for dtype, f64, tol in ((np.float32, False, 4e-6), (np.float64, True, 1e-14)):
        e = CO.fill_eps(clib, dtype, SEED, 9, 37, 11, m_offset=5)
        assert np.max(np.abs(e - O.philox_normal(SEED, 9, 37, 5, 16, f64=f64))) < tol
