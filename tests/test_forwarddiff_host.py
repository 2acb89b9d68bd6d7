"""The host-side gradient provider for order-0 targets (advancedvi.jl_amd/forwarddiff.py, problems.ADgradient) and the capability
dispatch of `init` (repgradelbo.jl:50-62) -- CPU only.  The reference's own entry problems declare LogDensityOrder{0}: the README model
(README.md:42-66) and the benchmark target (bench/benchmarks.jl:25-41)."""
import logging

import numpy as np
import pytest

import advancedvi_jl_amd as avi
from advancedvi_jl_amd import forwarddiff as FD
from advancedvi_jl_amd import objectives as OBJ
from oracle import oracle as O
from tests.helpers import BenchDist, ReadmeLogReg, readme_bijector


def _fd(f, x, h=1e-6):
    return np.array([(f(x + h * e) - f(x - h * e)) / (2 * h) for e in np.eye(x.size)])


@pytest.mark.parametrize("chunk", [1, 5, 64])
def test_readme_model_gradient_matches_the_oracles_closed_form(chunk):
    """d/dtheta of the README's logdensity by dual numbers == the oracle's hand-derived gradient of the same model, chained through the
    exp bijector (oracle.LogRegTarget "lognormal_exp_bijector" is README model + TransformedLogDensityProblem, README.md:91-119)."""
    rng = np.random.default_rng(5)
    n, p = 200, 12
    X = np.hstack([rng.normal(size=(n, p - 1)), np.ones((n, 1))])
    y = (rng.uniform(size=n) < 0.5).astype(float)
    model = ReadmeLogReg(X, y)
    tgt = O.LogRegTarget(X, y, "lognormal_exp_bijector")
    for _ in range(3):
        eta = rng.normal(size=p + 1) * 0.7
        theta = eta.copy(); theta[p] = np.exp(eta[p])
        v, g = FD.value_and_gradient(model.logdensity, theta, chunk)
        # eta-space: logdensity(theta(eta)) + eta_p;  d/deta_p = theta_p * dl/dtheta_p + 1
        g_eta = g.copy(); g_eta[p] = g[p] * theta[p] + 1.0
        l_ref, g_ref = tgt.logdensity_and_gradient(eta)
        assert abs(v + eta[p] - l_ref) <= 1e-12 * abs(l_ref)
        assert np.allclose(g_eta, g_ref, rtol=1e-11, atol=1e-12)


def test_rules_against_finite_differences():
    rng = np.random.default_rng(1)
    A = rng.normal(size=(4, 4))

    def f(x):
        z = np.concatenate([x[:2], np.exp(x[2:])])
        w = np.where(z > 0.5, z ** 2, np.tanh(z))
        m = z.reshape(2, 2)
        return ((A @ z) @ w + np.linalg.norm(z) + np.mean(np.maximum(z, 0.3)) + (z @ A).sum() + (m.T @ m).sum()
                + np.log1p(np.square(z)).sum() + np.sqrt(np.abs(z) + 1.0).sum() / (1.0 + z[0] * z[0]) - np.logaddexp(z[1], 2.0 * z[2]))

    x = np.array([0.2, 0.9, -0.3, 0.4])
    v, g = FD.value_and_gradient(f, x, chunk=3)
    assert v == pytest.approx(f(x), rel=1e-14)
    assert np.allclose(g, _fd(f, x), rtol=1e-6, atol=1e-7)


def test_an_operation_without_a_rule_raises():
    with pytest.raises(TypeError, match="no differentiation rule"):
        FD.value_and_gradient(lambda x: np.sort(x).sum(), np.arange(3.0))
    with pytest.raises(TypeError, match="no differentiation rule"):
        FD.value_and_gradient(lambda x: np.floor(x).sum(), np.arange(3.0))


def test_adgradient_wrapper_is_an_order_1_problem():
    prob = BenchDist(10)
    assert avi.capabilities(prob) < avi.LogDensityOrder(1)
    ad = avi.ADgradient("forwarddiff", prob)
    assert not (avi.capabilities(ad) < avi.LogDensityOrder(1)) and avi.dimension(ad) == 10
    x = np.linspace(-1, 1, 10)
    l, g = ad.logdensity_and_gradient(x)
    assert l == pytest.approx(prob.logdensity(x)) and np.allclose(g, -(x - 5.0))
    assert prob.grad_calls == 0          # the wrapped problem's own gradient is never used: order 0 means "differentiate through logdensity"
    with pytest.raises(ValueError):
        avi.ADgradient("zygote", prob)


def test_capability_dispatch_wraps_order_0_and_emits_the_references_info(caplog):
    """repgradelbo.jl:50-57: order 0 -> @info + AD through logdensity; order >= 1 -> the problem itself."""
    p = 4
    model = ReadmeLogReg(np.ones((3, p)), np.ones(3))
    trans = avi.TransformedProblem(model, readme_bijector(p))
    with caplog.at_level(logging.INFO, logger="advancedvi_jl_amd"):
        ad_prob = OBJ._ad_problem(avi.AutoMIVI(), trans)
    assert "is less than LogDensityOrder{1}()" in caplog.text and "directly differentiate through `LogDensityProblems.logdensity`" in caplog.text
    assert isinstance(ad_prob, avi.TransformedProblem) and isinstance(ad_prob.prob, avi.ADgradient) and ad_prob.bijector is trans.bijector
    caplog.clear()
    with caplog.at_level(logging.INFO, logger="advancedvi_jl_amd"):
        assert isinstance(OBJ._ad_problem(avi.AutoMIVI(), BenchDist(3), announce=False), avi.ADgradient)
        builtin = avi.DiagNormalProblem(np.zeros(3), np.ones(3))
        assert OBJ._ad_problem(avi.AutoMIVI(), builtin) is builtin
    assert caplog.text == ""
    with pytest.raises(TypeError, match="is less than LogDensityOrder"):
        OBJ._ad_problem(avi.AutoMIVI(target_ad=None), BenchDist(3))
