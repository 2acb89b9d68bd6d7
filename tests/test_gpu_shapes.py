"""Ragged, degenerate and large shapes through the C ABI vs the oracle on identical eps (edge cases the reference's
own tests touch only implicitly: d = 1, a single sample, dimensions that are no multiple of any tile, d >> BASELINE)."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED, make_family, make_problem

pytestmark = pytest.mark.gpu

CASES = [
    (avi.FULLRANK, 2048, 64, "diag", 0, np.float32), (avi.FULLRANK, 2048, 64, "diag", 3, np.float32),
    (avi.FULLRANK, 1000, 100, "dense", 0, np.float32), (avi.FULLRANK, 1537, 33, "diag", 4, np.float32),
    (avi.FULLRANK, 1537, 33, "diag", 4, np.float64), (avi.FULLRANK, 1024, 64, "diag", 3, np.float64),
    (avi.MEANFIELD, 1 << 20, 8, "diag", 0, np.float32), (avi.MEANFIELD, 100003, 7, "diag", 3, np.float32),
    (avi.FULLRANK, 1, 1, "diag", 0, np.float32), (avi.FULLRANK, 3, 1000, "diag", 2, np.float32),
    (avi.MEANFIELD, 1, 1, "diag", 2, np.float64), (avi.FULLRANK, 33, 4096, "dense", 3, np.float32),
    (avi.FULLRANK, 2, 1, "dense", 1, np.float64), (avi.MEANFIELD, 5, 3, "funnel", 3, np.float32),
    # second-generation full-rank route: unsplit product with the STL riders (d = 2048 / 512 / 256: 16 / 4 / 2 chain blocks per half),
    # dense target on the unsplit product
    (avi.FULLRANK, 2048, 128, "diag", 3, np.float32), (avi.FULLRANK, 512, 256, "dense", 3, np.float32),
    (avi.FULLRANK, 256, 128, "diag", 4, np.float32), (avi.FULLRANK, 2048, 512, "diag", 3, np.float32),
    (avi.FULLRANK, 1024, 1024, "diag", 0, np.float32),
    # beyond d * n_mc = 1024 * 512: the unsplit 64 x 64 product (k_fr_prod64) -- dense target (two products), a row-block count that
    # is no multiple of 4 (plain block -> tile map), the STL estimator behind it
    (avi.FULLRANK, 1024, 1024, "dense", 0, np.float32), (avi.FULLRANK, 1152, 512, "diag", 0, np.float32),
    (avi.FULLRANK, 1536, 512, "dense", 2, np.float32), (avi.FULLRANK, 2048, 384, "diag", 4, np.float32),
    (avi.FULLRANK, 4096, 512, "diag", 0, np.float32),   # 64 x 64 tiles for both contractions (k_fr_prod64 + k_fr_vjp64)
    # the shapes bench.py times: north star with the STL estimators (k_stl_* fed by the riders of the
    # north-star-size k_fr_prod32) and with the dense target (k_fr_prod32<DENSE> at 1024 x 256)
    (avi.FULLRANK, 1024, 256, "diag", 3, np.float32), (avi.FULLRANK, 1024, 256, "diag", 4, np.float32),
    (avi.FULLRANK, 1024, 256, "dense", 0, np.float32), (avi.FULLRANK, 1024, 256, "dense", 3, np.float32),
]


@pytest.mark.parametrize("family,d,M,kind,ent,dtype", CASES)
def test_shape(family, d, M, kind, ent, dtype):
    rng = np.random.default_rng(d + M)
    q, q_o = make_family(rng, d, family, dtype)
    prob, tgt = make_problem(rng, kind, d, dtype)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(dtype, family, d, M, ent, SEED)
    ctx.set_problem(prob)
    _, eps = ctx.sample(params, 3)
    v, g = ctx.estimate_gradient(params, 3)
    ctx.synchronize()
    ref = O.estimate_gradient(O.destructure(q_o), d, family, tgt, eps.cpu().numpy().astype(np.float64), ent)
    vt, gt = (1e-5, 2e-5) if dtype == np.float32 else (1e-12, 1e-11)
    assert abs(float(v.item()) - ref["value"]) <= vt * max(abs(ref["value"]), 1.0)
    assert np.linalg.norm(g.cpu().numpy() - ref["grad"]) <= gt * max(np.linalg.norm(ref["grad"]), 1.0)
    ctx.close()
