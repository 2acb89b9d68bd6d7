"""Random mid-size sweep over the one-estimate full-rank kernels (second generation: d a multiple of 32, f32; the solve kernels of the
sticking-the-landing estimators; first generation for the rest) -- the shapes an optimiser step runs at, between the small fuzz
(tests/test_gpu_fuzz.py) and the BASELINE sizes: value / gradient against the fp64 oracle, plus the chained-step entry against the
step-by-step sequence (bitwise on the graph of launches, to rounding on the row-owning launch-free loop)."""
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
        d = int(rng.choice([96, 160, 224, 320, 416, 544, 800, 1056, 72, 200]))
        M = int(rng.choice([8, 32, 64, 96, 160, 288]))
        ent = int(rng.integers(0, 5))
        kind = ("diag", "dense")[int(rng.integers(0, 2))]
        out.append((d, M, ent, kind))
    return out


@pytest.mark.parametrize("d,M,ent,kind", _cases(24, 20261001))
def test_single_call_sweep(d, M, ent, kind):
    rng = np.random.default_rng(d * 17 + M + ent)
    q, q_o = make_family(rng, d, avi.FULLRANK, np.float32, mu_scale=0.4)
    prob, tgt = make_problem(rng, kind, d, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
    ctx.set_problem(prob)
    idx = int(rng.integers(0, 1 << 30))
    _, eps = ctx.sample(params, idx)
    v, g = ctx.estimate_gradient(params, idx)
    ref = O.estimate_gradient(O.destructure(q_o), d, avi.FULLRANK, tgt, eps.cpu().numpy().astype(np.float64), ent)
    assert abs(float(v.item()) - ref["value"]) <= 2e-5 * max(abs(ref["value"]), 1.0), (float(v.item()), ref["value"])
    assert np.linalg.norm(g.cpu().numpy() - ref["grad"]) <= 4e-5 * max(np.linalg.norm(ref["grad"]), 1.0)
    gC = g.cpu().numpy()[d:].reshape(d, d, order="F")
    assert np.all(np.triu(gC, 1) == 0.0)
    # five chained Adam + ClipScale steps in one call == the step-by-step sequence (bitwise)
    T = 5
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
    if M <= 32 and kind == "diag":   # the row-owning launch-free loop (k_fr_rows_loop: FMAs on resident rows): the sequence to rounding
        a_, b_ = pa.cpu().numpy().astype(np.float64), pb.cpu().numpy().astype(np.float64)
        assert np.linalg.norm(a_ - b_) <= 5e-6 * np.linalg.norm(a_)
    else:
        assert np.array_equal(pa.cpu().numpy(), pb.cpu().numpy())
    ctx.close()
