"""Random mid-size sweep over the mean-field kernels (k_mf_main and the launch-free loops): d up to 4096, sample counts on both sides of the
256-column block, f32 / f64, diagonal-Gaussian and funnel targets, the five estimators -- value / gradient against the fp64 oracle and the
chained-step entry against the step-by-step sequence (bitwise: these loops run the single calls' arithmetic; the sticking-the-landing
estimators to an ulp in a few entries)."""
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
        d = int(rng.choice([6, 40, 130, 256, 500, 1024, 1500, 2048, 4096]))
        M = int(rng.choice([1, 4, 33, 64, 200, 256, 300, 512]))
        ent = int(rng.integers(0, 5))
        kind = ("diag", "funnel")[int(rng.integers(0, 2))]
        dt = ("float32", "float64")[int(rng.integers(0, 2))]
        out.append((d, M, ent, kind, dt))
    return out


@pytest.mark.parametrize("d,M,ent,kind,dt", _cases(24, 20261002))
def test_meanfield_sweep(d, M, ent, kind, dt):
    dtype = np.float32 if dt == "float32" else np.float64
    rng = np.random.default_rng(d * 13 + M + ent)
    q, q_o = make_family(rng, d, avi.MEANFIELD, dtype, mu_scale=0.3 if kind == "funnel" else 1.0)
    prob, tgt = make_problem(rng, kind, d, dtype)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(dtype, avi.MEANFIELD, d, M, ent, SEED)
    ctx.set_problem(prob)
    idx = int(rng.integers(0, 1 << 30))
    _, eps = ctx.sample(params, idx)
    v, g = ctx.estimate_gradient(params, idx)
    ref = O.estimate_gradient(O.destructure(q_o), d, avi.MEANFIELD, tgt, eps.cpu().numpy().astype(np.float64), ent)
    vt, gt = (2e-5, 4e-5) if dtype == np.float32 else (1e-11, 1e-10)
    assert abs(float(v.item()) - ref["value"]) <= vt * max(abs(ref["value"]), 1.0), (float(v.item()), ref["value"])
    assert np.linalg.norm(g.cpu().numpy() - ref["grad"]) <= gt * max(np.linalg.norm(ref["grad"]), 1.0)
    T = 6
    pa = ctx.to_device(params).clone()
    st = ctx.empty(2 * pa.numel()).zero_()
    for t in range(T):
        v1, g1 = ctx.estimate_gradient(pa, 70 + t)
        ctx.adam_update(pa, g1, st, t + 1, 1e-3)
        ctx.clip_scale(pa, 1e-5)
    pb = ctx.to_device(params).clone()
    st2 = ctx.empty(2 * pb.numel()).zero_()
    ctx.optimize_steps(pb, st2, 70, 0, T, 1, 1e-3, 1e-5, None)
    ctx.synchronize()
    a_, b_ = pa.cpu().numpy(), pb.cpu().numpy()
    if ent in (3, 4) and kind == "diag":   # sticking-the-landing estimators in k_mf_sgd_loop: a few entries one ulp apart (fp contraction of the residual term)
        ulp = np.spacing(np.abs(a_).astype(dtype))
        assert np.all(np.abs(a_ - b_) <= 4 * ulp) and np.mean(a_ != b_) < 0.02
    else:
        assert np.array_equal(a_, b_)
    ctx.close()
