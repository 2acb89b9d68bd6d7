"""Randomised shape / option sweep through the C ABI vs the oracle on identical eps.  Fixed seed (reproducible), ~60
cases drawn over family x dtype x estimator x target x (d, M) including sizes that are not multiples of any tile, the
partials + finalize route and the graph-batched route."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED, make_family, make_problem

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        family = int(rng.integers(0, 2))
        dtype = (np.float32, np.float64)[int(rng.integers(0, 2))]
        ent = int(rng.integers(0, 5))
        kind = ("diag", "dense", "funnel", "logreg0", "logreg1")[int(rng.integers(0, 5))]
        d = int(rng.choice([2, 3, 5, 17, 31, 32, 33, 64, 65, 100, 129, 200, 257]))
        M = int(rng.choice([1, 2, 7, 16, 31, 32, 33, 64, 100, 257]))
        if kind.startswith("logreg"):
            d = min(d, 65)
        out.append((family, dtype, ent, kind, d, M))
    return out


@pytest.mark.parametrize("family,dtype,ent,kind,d,M", _cases(60, 20260928))
def test_fuzz(family, dtype, ent, kind, d, M):
    rng = np.random.default_rng(d * 1009 + M * 13 + ent)
    q, q_o = make_family(rng, d, family, dtype, mu_scale=0.3 if kind != "diag" else 1.0)
    prob, tgt = make_problem(rng, kind, d, dtype)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(dtype, family, d, M, ent, SEED)
    ctx.set_problem(prob)
    idx = int(rng.integers(0, 1 << 40))
    _, eps = ctx.sample(params, idx)
    ref = O.estimate_gradient(O.destructure(q_o), d, family, tgt, eps.cpu().numpy().astype(np.float64), ent)
    vt, gt = (2e-5, 4e-5) if dtype == np.float32 else (1e-11, 1e-10)
    gs = max(np.linalg.norm(ref["grad"]), 1.0)

    def check(v, g, scale=1.0):
        assert abs(float(v.item()) - ref["value"]) <= scale * vt * max(abs(ref["value"]), 1.0)
        assert np.linalg.norm(g.cpu().numpy() - ref["grad"]) <= scale * gt * gs

    # the built-in logistic regression has two f32 kernel families (chosen by problem size): cover both
    routes = (1, 2) if kind.startswith("logreg") and dtype == np.float32 else (0,)
    for route in routes:
        ctx.set_logreg_route(route)
        check(*ctx.estimate_gradient(params, idx))
        check(*ctx.finalize(params, ctx.estimate_partials(params, idx)), 2.0)       # shard route
        p = ctx.to_device(params)
        v, g = ctx.empty(1), ctx.empty(ctx.params_len)
        ctx.estimate_gradient_n(p, idx - 2 if idx >= 2 else 0, 3 if idx >= 2 else 1, v, g)   # batched route, last = idx
        ctx.synchronize()                                                                  # (short batch: eager chain for the full-rank family)
        if idx >= 2:
            check(v, g)
        if idx >= 6:
            ctx.estimate_gradient_n(p, idx - 6, 7, v, g)                                   # captured graph, last = idx
            ctx.synchronize()
            check(v, g)
    ctx.close()
