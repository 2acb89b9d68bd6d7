"""Generic Stacked bijector around any target (mivi_set_bijector_stacked): README.md:76-82,91-119,
docs/src/tutorials/constrained.md:154-196 (`TransformedLogDensityProblem(prob, binv)`).  Oracle: oracle.StackedBijectorTarget."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED, OraclePlugin, make_family, make_problem, rel_err

pytestmark = pytest.mark.gpu


def _blocks(d):
    return [(0, 3, "exp"), (3, d // 2, "identity"), (d // 2, d // 2 + 5, "exp")]   # the rest: no block = identity


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
@pytest.mark.parametrize("kind", ["diag", "dense", "callback", "logreg0"])
@pytest.mark.parametrize("ent", [0, 3], ids=["closedform", "stl"])
def test_stacked_bijector_matches_oracle(kind, family, dtype, ent):
    d, M = 40, 24
    rng = np.random.default_rng(31)
    q, _ = make_family(rng, d, family, dtype, mu_scale=0.3)
    prob_a, tgt_o = make_problem(rng, "diag" if kind == "callback" else kind, d, dtype)
    if kind == "callback":
        prob_a = OraclePlugin(tgt_o)              # the plugin sees CONSTRAINED samples, like the reference's wrapped problem
    blocks = _blocks(d)
    tgt = O.StackedBijectorTarget(tgt_o, blocks)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(dtype, family, d, M, ent, SEED)
    ctx.set_problem(avi.TransformedProblem(prob_a, avi.StackedBijector(blocks)))
    _, eps = ctx.sample(params, 3)
    v, g = ctx.estimate_gradient(params, 3)
    ref = O.estimate_gradient(params.astype(np.float64), d, family, tgt, eps.cpu().numpy().astype(np.float64), ent)
    vtol, gtol = (1e-5, 2e-5) if dtype == np.float32 else (1e-12, 1e-10)
    assert abs(float(v.item()) - ref["value"]) <= vtol * abs(ref["value"])
    assert rel_err(g.cpu().numpy(), ref["grad"]) < gtol
    # value-only route (estimate_objective) carries the log-Jacobian too
    vo = ctx.estimate_objective(params, 3, n_samples=M, entropy=2)
    ro = O.estimate_objective(O.restructure(params.astype(np.float64), d, family), tgt, eps.cpu().numpy().astype(np.float64), 2)
    assert abs(float(vo.item()) - ro) <= vtol * abs(ro)
    # removing the bijector restores the plain target
    ctx.set_problem(prob_a)
    v2, g2 = ctx.estimate_gradient(params, 3)
    ref2 = O.estimate_gradient(params.astype(np.float64), d, family, tgt_o, eps.cpu().numpy().astype(np.float64), ent)
    assert abs(float(v2.item()) - ref2["value"]) <= vtol * abs(ref2["value"])
    assert rel_err(g2.cpu().numpy(), ref2["grad"]) < gtol
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
def test_c5_funnel_through_the_generic_bijector_equals_the_baked_funnel(family, dtype):
    """BASELINE config 5's target re-expressed: constrained funnel under Stacked([exp on [0,1), identity]) == FunnelProblem."""
    d, M, ent = (2048, 64, 3) if family == avi.MEANFIELD else (256, 64, 3)
    q = avi.MeanFieldGaussian(np.zeros(d, dtype), np.ones(d, dtype)) if family == avi.MEANFIELD else \
        avi.FullRankGaussian(np.zeros(d, dtype), np.eye(d, dtype=dtype))
    params, _ = avi.destructure(q)
    baked = avi.MiviContext(dtype, family, d, M, ent, SEED)
    baked.set_problem(avi.FunnelProblem(d, 1.5))
    vb, gb = baked.estimate_gradient(params, 9)
    gen = avi.MiviContext(dtype, family, d, M, ent, SEED)
    gen.set_problem(avi.TransformedProblem(avi.FunnelConstrainedProblem(d, 1.5), avi.StackedBijector([(0, 1, "exp"), (1, d, "identity")])))
    vg, gg = gen.estimate_gradient(params, 9)
    tol = 1e-6 if dtype == np.float32 else 1e-12
    assert abs(float(vg.item()) - float(vb.item())) <= tol * abs(float(vb.item()))
    assert rel_err(gg.cpu().numpy(), gb.cpu().numpy()) < (5e-6 if dtype == np.float32 else 1e-11)
    baked.close(); gen.close()


def test_bijector_argument_checks():
    ctx = avi.MiviContext(np.float32, avi.MEANFIELD, 8, 4, 0, SEED)
    ctx.set_problem(avi.DiagNormalProblem(np.zeros(8, np.float32), np.ones(8, np.float32)))
    for bad in ([(0, 9, "exp")], [(0, 4, "exp"), (3, 6, "identity")], [(-1, 2, "exp")]):
        with pytest.raises(avi.MiviError):
            ctx.set_bijector(avi.StackedBijector(bad))
    ctx.close()


@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
@pytest.mark.parametrize("rule", ["descent", "adam"])
def test_device_loop_honours_the_bijector(family, rule):
    """the launch-free mean-field loop has no Stacked-bijector handling, so a TransformedProblem must NOT take it.
    `optimize(device_loop=True)` == the host-driven `step` loop, and both differ from the untransformed problem."""
    import warnings
    d, T = 12, 9
    rng = np.random.default_rng(11)
    mu, sig = rng.uniform(0.5, 1.5, size=d).astype(np.float32), rng.uniform(1.5, 2.5, size=d).astype(np.float32)
    base = avi.DiagNormalProblem(mu, sig)
    prob = avi.TransformedProblem(base, avi.StackedBijector([(0, 4, "exp"), (4, d, "identity")]))
    q0 = (avi.MeanFieldGaussian(np.zeros(d, np.float32), np.full(d, 0.3, np.float32)) if family == avi.MEANFIELD
          else avi.FullRankGaussian(np.zeros(d, np.float32), 0.3 * np.eye(d, dtype=np.float32)))
    opt = avi.Descent(1e-3) if rule == "descent" else avi.Adam(1e-2)
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=8, optimizer=opt, averager=avi.NoAveraging(), operator=avi.ClipScale())
    res = {}
    for name, pr, dev in (("dev", prob, True), ("host", prob, False), ("plain", base, True)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _, info, st = avi.optimize(avi.PhiloxRNG(5), alg, T, pr, q0, device_loop=dev)
        res[name] = (st["params"].cpu().numpy().copy(), np.array([i["elbo"] for i in info]))
    assert np.array_equal(res["dev"][0], res["host"][0])
    assert np.allclose(res["dev"][1], res["host"][1], rtol=1e-6)
    assert not np.allclose(res["dev"][0], res["plain"][0], rtol=1e-4)   # the bijector changed the problem


@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
def test_estimate_gradient_n_honours_the_bijector(family):
    """mivi_estimate_gradient_n under a bijector == the last of n single calls (and != the untransformed estimate)."""
    d, M, n = 16, 32, 7
    rng = np.random.default_rng(12)
    q, _ = make_family(rng, d, family, np.float32, mu_scale=0.3)
    prob_a, _ = make_problem(rng, "diag", d, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, family, d, M, 0, SEED)
    ctx.set_problem(avi.TransformedProblem(prob_a, avi.StackedBijector([(0, 5, "exp")])))
    p = ctx.to_device(params)
    v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    ctx.estimate_gradient_n(p, 20, n, v, g)
    v1, g1 = ctx.estimate_gradient(p, 20 + n - 1)
    assert abs(float(v.item()) - float(v1.item())) <= 1e-6 * abs(float(v1.item()))
    assert rel_err(g.cpu().numpy(), g1.cpu().numpy()) < 1e-6
    ctx.set_problem(prob_a)
    v0, _ = ctx.estimate_gradient(p, 20 + n - 1)
    assert abs(float(v0.item()) - float(v1.item())) > 1e-3 * abs(float(v1.item()))
    ctx.close()
