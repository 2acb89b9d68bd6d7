"""Pins the CPU oracle (oracle/oracle.py) against every known-answer test the reference holds for the
hot path, restated with the reference's own tolerances, plus AD-of-the-forward
(torch CPU autograd standing in for the reference's AD backends) and the Philox known answers.
No GPU, no product code."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import oracle_torch as OT
from tests.helpers import SEED, make_family, make_problem


def test_philox4x32_10_known_answers():
    """Random123 kat_vectors (philox4x32 10 rounds)."""
    kat = [
        ([0, 0, 0, 0], (0, 0), [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, (0xFFFFFFFF, 0xFFFFFFFF), [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], (0xA4093822, 0x299F31D0),
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, out in kat:
        got = O.philox4x32_10(np.array(ctr, dtype=np.uint32), key)
        assert [int(x) for x in got] == out


def _test_normal(d=5, fullrank=False):
    """normal_meanfield / normal_fullrank: test/models/normal.jl:36-75 (mu = 5, sigma0 = 0.3)."""
    mu = np.full(d, 5.0)
    if fullrank:
        L = 0.3 * np.eye(d)
        return O.DenseNormalTarget(mu, L), O.MvLocationScale(mu, L)
    return O.DiagNormalTarget(mu, np.full(d, 0.3)), O.MvLocationScale(mu, np.full(d, 0.3))


@pytest.mark.parametrize("fullrank", [False, True])
@pytest.mark.parametrize("M", [1, 10])
def test_stl_gradient_zero_at_optimum(fullrank, M):
    """test/algorithms/klminrepgraddescent.jl:66-87: norm(grad) ~ 0, atol 1e-5, for q = pi."""
    tgt, q = _test_normal(fullrank=fullrank)
    eps = np.random.default_rng(M).normal(size=(5, M))
    r = O.estimate_gradient(O.destructure(q), 5, int(fullrank), tgt, eps, O.ENT_STL)
    assert np.linalg.norm(r["grad"]) < 1e-5
    _, g_ad = OT.value_and_gradient(O.destructure(q), 5, int(fullrank), tgt, eps, O.ENT_STL)
    assert np.linalg.norm(g_ad) < 1e-5


@pytest.mark.parametrize("fullrank", [False, True])
def test_estimate_objective_zero_at_optimum(fullrank):
    """test/algorithms/klminrepgraddescent.jl:36-37: estimate_objective(q = pi, n = 10^5) ~ 0, atol 1e-2;
    the prox variant (klminrepgradproxdescent.jl:36-37) uses atol 1e-3 with the same estimator."""
    tgt, q = _test_normal(fullrank=fullrank)
    eps = O.philox_normal(SEED, 0, 5, 0, 10 ** 5, f64=True)
    Z = O.rand_batch(q, eps)
    # vectorised restatement of estimate_objective for this target (the per-column loop is identical)
    r = (Z - tgt.mean[:, None]) / 0.3
    ell = -0.5 * np.sum(r * r, axis=0) - 5 * np.log(0.3) - 2.5 * O.LOG2PI
    ent = np.mean(0.5 * np.sum(eps * eps, axis=0)) + 2.5 * O.LOG2PI + 5 * np.log(0.3)
    assert abs(-(ell.mean() + ent)) < 1e-2
    # and the literal per-column path on a subset agrees with the vectorised one
    sub = eps[:, :200]
    lit = O.estimate_objective(q, tgt, sub, O.ENT_MONTE_CARLO)
    Zs = O.rand_batch(q, sub)
    rs = (Zs - tgt.mean[:, None]) / 0.3
    vec = -((-0.5 * np.sum(rs * rs, axis=0) - 5 * np.log(0.3) - 2.5 * O.LOG2PI).mean()
            + np.mean(0.5 * np.sum(sub * sub, axis=0)) + 2.5 * O.LOG2PI + 5 * np.log(0.3))
    assert abs(lit - vec) < 1e-10


@pytest.mark.parametrize("fullrank", [False, True])
def test_family_entropy_logpdf_match_mvnormal(fullrank):
    """test/families/location_scale.jl:38-47: logpdf(q,z) ~ logpdf(MvNormal(mu, C C')) and
    entropy(q) ~ entropy(MvNormal) with the reference's scale tril(I + ones/2) (:13)."""
    d = 10
    rng = np.random.default_rng(3)
    loc = rng.normal(size=d)
    scale = np.tril(np.eye(d) + np.ones((d, d)) / 2) if fullrank else np.ones(d)
    q = O.MvLocationScale(loc, scale)
    cov = scale @ scale.T if fullrank else np.diag(scale ** 2)
    sign, logdet = np.linalg.slogdet(cov)
    ent_true = 0.5 * d * (1 + O.LOG2PI) + 0.5 * logdet
    assert abs(O.entropy_closed_form(q) - ent_true) < 1e-10
    z = O.rand_batch(q, rng.normal(size=(d, 1)))[:, 0]
    r = z - loc
    lp_true = -0.5 * r @ np.linalg.solve(cov, r) - 0.5 * logdet - 0.5 * d * O.LOG2PI
    assert abs(O.logpdf(q, z) - lp_true) <= 1e-2 * abs(lp_true)   # the reference's rtol
    assert abs(O.logpdf(q, z) - lp_true) < 1e-9


@pytest.mark.parametrize("fullrank", [False, True])
def test_sample_moments(fullrank):
    """test/families/location_scale.jl:68-97: mean/var/cov of 10^6 samples within rtol 1e-2."""
    d = 10
    rng = np.random.default_rng(4)
    loc = rng.normal(size=d)
    scale = np.tril(np.eye(d) + np.ones((d, d)) / 2) if fullrank else np.ones(d)
    q = O.MvLocationScale(loc, scale)
    Z = O.rand_batch(q, O.philox_normal(SEED, 7, d, 0, 10 ** 6, f64=True))
    cov = scale @ scale.T if fullrank else np.diag(scale ** 2)
    assert np.allclose(Z.mean(axis=1), loc, rtol=1e-2, atol=1e-2)
    assert np.allclose(Z.var(axis=1), np.diag(cov), rtol=1e-2)
    assert np.allclose(np.cov(Z), cov, rtol=1e-2, atol=2e-2)


def test_meanfield_destructure_length_and_roundtrip():
    """test/families/location_scale.jl:146-155: length(params) == 2d and re(params) == q."""
    d = 7
    q = O.MvLocationScale(np.arange(d, dtype=float), np.arange(1, d + 1, dtype=float))
    p = O.destructure(q)
    assert p.shape[0] == 2 * d
    q2 = O.restructure(p, d, O.MEANFIELD)
    assert np.array_equal(q2.location, q.location) and np.array_equal(q2.scale, q.scale)
    # full-rank: [mu; vec(C)] with LowerTriangular re-projection
    C = np.tril(np.arange(1, d * d + 1, dtype=float).reshape(d, d))
    pf = O.destructure(O.MvLocationScale(q.location, C))
    assert pf.shape[0] == d + d * d
    pf_dirty = pf.copy()
    pf_dirty[d + d] = 123.0  # element (0, 1): above the diagonal, must be ignored
    assert np.array_equal(O.restructure(pf_dirty, d, O.FULLRANK).scale, C)


def test_rrule_seam_uses_plugin_gradient_verbatim():
    """test/general/mixedad_logdensity.jl:2-25,37-61: a plugin with a deliberately wrong gradient [1,2,3]
    must see exactly that gradient propagated (pullback = dy' * grad, src/mixedad_logdensity.jl:23-34)."""

    class MixedADTestModel:
        def dimension(self):
            return 3

        def logdensity(self, z):
            return float(-0.5 * z @ z)

        def logdensity_and_gradient(self, z):
            return self.logdensity(z), np.array([1.0, 2.0, 3.0])

    d, M = 3, 4
    q = O.MvLocationScale(np.zeros(d), np.ones(d))
    eps = np.random.default_rng(0).normal(size=(d, M))
    r = O.estimate_gradient(O.destructure(q), d, O.MEANFIELD, MixedADTestModel(), eps, O.ENT_CLOSED_FORM)
    assert np.allclose(r["grad"][:d], -np.array([1.0, 2.0, 3.0]))
    assert np.allclose(r["grad"][d:], -(np.array([1.0, 2.0, 3.0])[:, None] * eps).mean(axis=1) - 1.0)


KINDS = ["diag", "dense", "logreg0", "logreg1", "funnel"]


@pytest.mark.parametrize("family", [O.MEANFIELD, O.FULLRANK])
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("ent", range(5))
def test_closed_form_vjp_equals_ad_of_forward(family, kind, ent):
    """The reference's gradient IS reverse-mode AD of estimate_repgradelbo_ad_forward
    (src/algorithms/repgradelbo.jl:142-149, src/AdvancedVI.jl:57-67)."""
    rng = np.random.default_rng(100 * family + ent)
    d, M = 7, 5
    _, q = make_family(rng, d, family)
    _, tgt = make_problem(rng, kind, d)
    eps = rng.normal(size=(d, M))
    params = O.destructure(q)
    r = O.estimate_gradient(params, d, family, tgt, eps, ent)
    v_ad, g_ad = OT.value_and_gradient(params, d, family, tgt, eps, ent)
    assert abs(r["value"] - v_ad) < 1e-10 * max(1.0, abs(v_ad))
    assert np.max(np.abs(r["grad"] - g_ad)) < 1e-9
    assert abs(O.estimate_repgradelbo_forward(params, d, family, tgt, eps, ent) - v_ad) < 1e-10 * max(1.0, abs(v_ad))
    # shard-additivity of the partial buffer + finalize
    pa = O.estimate_gradient(params, d, family, tgt, eps[:, :2], ent)["partials"]
    pb = O.estimate_gradient(params, d, family, tgt, eps[:, 2:], ent)["partials"]
    v2, g2 = O.finalize_partials(pa + pb, params, d, family, ent, M)
    assert abs(v2 - v_ad) < 1e-10 * max(1.0, abs(v_ad)) and np.max(np.abs(g2 - g_ad)) < 1e-9


def test_target_gradients_by_finite_differences():
    """LogReg / funnel targets appear only in the reference's docs (parity unpinned there): pin the
    restated gradients with central differences."""
    rng = np.random.default_rng(8)
    d = 6
    for kind in KINDS:
        _, tgt = make_problem(rng, kind, d)
        z = rng.normal(size=d) * 0.5
        _, g = tgt.logdensity_and_gradient(z)
        for i in range(d):
            h = 1e-6
            zp, zm = z.copy(), z.copy()
            zp[i] += h
            zm[i] -= h
            fd = (tgt.logdensity(zp) - tgt.logdensity(zm)) / (2 * h)
            assert abs(fd - g[i]) < 1e-5 * max(1.0, abs(g[i])), (kind, i)


def test_clip_scale():
    """src/optimization/clip_scale.jl:18-29."""
    d = 4
    p = np.concatenate([np.zeros(d), np.array([1.0, 1e-9, -2.0, 0.5])])
    assert np.array_equal(O.clip_scale(p, d, O.MEANFIELD, 1e-5)[d:], [1.0, 1e-5, 1e-5, 0.5])
    C = np.tril(np.ones((d, d)))
    C[1, 1] = -1.0
    pf = np.concatenate([np.zeros(d), C.reshape(-1, order="F")])
    out = O.clip_scale(pf, d, O.FULLRANK, 1e-5)[d:].reshape(d, d, order="F")
    assert out[1, 1] == 1e-5 and out[2, 1] == 1.0 and np.all(np.triu(out, 1) == 0)


@pytest.mark.parametrize("family", [O.MEANFIELD, O.FULLRANK])
def test_proximal_location_scale_entropy_stationarity(family):
    """test/general/proximal_location_scale_entropy.jl:3-56: the operator's output L' must satisfy
    grad logabsdet(L') = grad ||L' - L||^2 / (2 eta)  (d = 5, L = I, eta = 1e-2), checked with AD of both sides."""
    import torch
    d, eta = 5, 1e-2
    L = np.eye(d)
    params = np.concatenate([np.zeros(d), np.ones(d) if family == O.MEANFIELD else L.reshape(-1, order="F")])
    out = O.proximal_location_scale_entropy(params, d, family, eta)
    Lp = np.diag(out[d:]) if family == O.MEANFIELD else out[d:].reshape(d, d, order="F")
    x = torch.tensor(Lp, dtype=torch.float64, requires_grad=True)
    left = torch.autograd.grad(torch.linalg.slogdet(torch.tril(x))[1], x)[0].numpy()
    right = ((Lp - L) / eta)
    assert np.allclose(np.diag(left), np.diag(right), rtol=1e-10)       # the diagonal is where logabsdet lives
    assert np.allclose(np.tril(right, -1), 0.0) and np.allclose(out[:d], params[:d])
    # closed form of the scalar problem: c' = (c + sqrt(c^2 + 4 eta)) / 2
    assert np.allclose(np.diag(Lp), (1.0 + np.sqrt(1.0 + 4 * eta)) / 2)


def test_stepsize_from_optimizer_state():
    """src/optimization/proximal_location_scale_entropy.jl:26-42."""
    assert O.stepsize_from_optimizer_state("descent", eta=0.3) == 0.3
    assert np.isclose(O.stepsize_from_optimizer_state("dog", v=4.0, r=3.0), 1.5)
    assert np.isclose(O.stepsize_from_optimizer_state("dowg", v=4.0, r=3.0), 4.5)
    with pytest.raises(ValueError):
        O.stepsize_from_optimizer_state("adam")


def test_gaussian_expectation_gradient_and_hessian_known_answer():
    """test/general/gauss_expected_grad_hess.jl:31-56 (first-order capability): logpi(x) = -x' S x / 2,
    q = N(1, 0.1^2 I): E grad = -S mu, E hess = -S, atol 1e-1 (n = 10^6 there; 2*10^5 of the fixed Philox stream here)."""
    class TestQuad:
        def __init__(self, S):
            self.S = S

        def logdensity_and_gradient(self, x):
            return float(-x @ self.S @ x / 2), -self.S @ x

    S = np.array([[2.0, -0.1], [-0.1, 2.0]])
    q = O.MvLocationScale(np.ones(2), np.diag([0.1, 0.1]))
    u = O.philox_normal(SEED, 0, 2, 0, 200000, f64=True)
    lp, g, H = O.gaussian_expectation_gradient_and_hessian(q, TestQuad(S), u)
    assert np.allclose(g, -S @ np.ones(2), atol=1e-1)
    assert np.allclose(H, -S, atol=1e-1)
    # exact identity behind the estimator: for a quadratic target, C' \ mean(u g') = -(C' \ mean(u z')) S
    z = np.tril(q.scale) @ u + q.location[:, None]
    assert np.allclose(H, -np.linalg.solve(np.tril(q.scale).T, (u @ z.T) / u.shape[1]) @ S, rtol=1e-10)
    with pytest.raises(TypeError):
        O.gaussian_expectation_gradient_and_hessian(O.MvLocationScale(np.ones(2), np.ones(2)), TestQuad(S), u)


def test_gaussian_expectation_gradient_and_hessian_known_answer_second_order():
    """test/general/gauss_expected_grad_hess.jl:45-56 with LogDensityOrder{2}: the same TestQuad also provides
    logdensity_gradient_and_hessian (:16-20 there: (lp, -S x, -S)); the second-order branch (src :61-83) averages the Hessians:
    E hess = -S exactly, E grad = -S mu within atol 1e-1; the gradient and logpi averages equal the first-order branch's on the
    same draws."""
    class TestQuad2:
        def __init__(self, S):
            self.S = S

        def logdensity_and_gradient(self, x):
            return float(-x @ self.S @ x / 2), -self.S @ x

        def logdensity_gradient_and_hessian(self, x):
            return float(-x @ self.S @ x / 2), -self.S @ x, -self.S

    S = np.array([[2.0, -0.1], [-0.1, 2.0]])
    q = O.MvLocationScale(np.ones(2), np.diag([0.1, 0.1]))
    u = O.philox_normal(SEED, 0, 2, 0, 200000, f64=True)
    lp, g, H = O.gaussian_expectation_gradient_and_hessian_order2(q, TestQuad2(S), u)
    assert np.allclose(g, -S @ np.ones(2), atol=1e-1)
    assert np.allclose(H, -S, rtol=1e-12)
    lp1, g1, _ = O.gaussian_expectation_gradient_and_hessian(q, TestQuad2(S), u)
    assert abs(lp - lp1) <= 1e-12 * abs(lp1) and np.allclose(g, g1, rtol=1e-12)
    # the oracle's own Gaussian targets carry the Hessians the built-in device targets write
    rng = np.random.default_rng(2)
    dn = O.DenseNormalTarget(rng.normal(size=3), np.tril(rng.normal(size=(3, 3))) + 2 * np.eye(3))
    x, h = rng.normal(size=3), 1e-5
    fd = np.array([(dn.logdensity_and_gradient(x + h * e)[1] - dn.logdensity_and_gradient(x - h * e)[1]) / (2 * h) for e in np.eye(3)])
    assert np.allclose(dn.logdensity_gradient_and_hessian(x)[2], fd, rtol=1e-6, atol=1e-8)
    dg = O.DiagNormalTarget(rng.normal(size=3), rng.uniform(0.5, 2, size=3))
    fd = np.array([(dg.logdensity_and_gradient(x + h * e)[1] - dg.logdensity_and_gradient(x - h * e)[1]) / (2 * h) for e in np.eye(3)])
    assert np.allclose(dg.logdensity_gradient_and_hessian(x)[2], fd, rtol=1e-6, atol=1e-8)


def test_stacked_bijector_target_chain_rule_and_the_funnel_identity():
    """oracle.StackedBijectorTarget (README.md:76-82,91-119): gradient == central finite differences of its own value, and
    wrapping the constrained funnel with exp on coordinate 0 reproduces FunnelStackedTarget (value and gradient)."""
    rng = np.random.default_rng(5)
    d = 9
    inner = O.DiagNormalTarget(rng.normal(size=d), rng.uniform(0.5, 2.0, size=d))
    tgt = O.StackedBijectorTarget(inner, [(0, 2, "exp"), (2, 5, "identity"), (6, 8, "exp")])
    eta = 0.4 * rng.normal(size=d)
    v, g = tgt.logdensity_and_gradient(eta)
    h = 1e-6
    fd = np.array([(tgt.logdensity(eta + h * e) - tgt.logdensity(eta - h * e)) / (2 * h) for e in np.eye(d)])
    assert np.allclose(g, fd, rtol=1e-7, atol=1e-7)
    assert abs(v - tgt.logdensity(eta)) < 1e-12
    # funnel: constrained + Stacked(exp on s) == the hand-derived unconstrained funnel
    f_c = O.StackedBijectorTarget(O.FunnelConstrainedTarget(d, 1.5), [(0, 1, "exp"), (1, d, "identity")])
    f_u = O.FunnelStackedTarget(d, 1.5)
    v1, g1 = f_c.logdensity_and_gradient(eta)
    v2, g2 = f_u.logdensity_and_gradient(eta)
    assert abs(v1 - v2) <= 1e-12 * max(1.0, abs(v2))
    assert np.allclose(g1, g2, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("alpha", [100, 100.0])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cocob_restatement_passes_the_reference_rule_test(alpha, dtype):
    """test/general/rules.jl:1-31 for COCOB (src/optimization/rules.jl:78-96): single-sample least-squares SGD, 10^4 steps,
    loss(X, w) < loss_0 / 10, eltype preserved."""
    rng = np.random.default_rng(5)
    d, n, T = 10, 1000, 10 ** 4
    w = rng.normal(size=d).astype(dtype)
    X = rng.random((n, d)).astype(dtype)
    w_true = rng.normal(size=d).astype(dtype)
    loss = lambda A, v: float(np.mean((A @ v - A @ w_true) ** 2))
    l0 = loss(X, w)
    st = O.cocob_init(w)
    for t in range(T):
        xi = X[rng.integers(n)]
        g = (2 * (xi @ w - xi @ w_true) * xi).astype(dtype)       # gradient of the one-row loss
        w, st = O.cocob_step(w, g, st, dtype(alpha))
    assert w.dtype == dtype
    assert loss(X, w) < l0 / 10


def test_proximal_operator_negative_diagonal_stays_positive_and_stationary():
    """A scale-diagonal entry an un-clipped step pushed below zero: the proximal operator's exact value is gamma / |c| + O(gamma^2) > 0.  The
    reference's expression c + (sqrt(c^2 + 4 gamma) - c) / 2 (proximal_location_scale_entropy.jl:56) cancels to exactly 0 in Float32 once
    4 gamma < eps c^2.  The ORACLE keeps the reference's literal expression (f64); the DEVICE (csrc/optim_rules.h) evaluates, for c < 0, the
    same value as 2 gamma / (sqrt(c^2 + 4 gamma) - c) -- a deliberate, documented divergence in Float32 behaviour.  Checked: the two forms
    agree to rounding in f64 wherever the literal one is accurate; the stable form is stationary (c' - c = gamma / c') and positive."""
    d = 6
    c = np.array([-3.0, -1e-3, -1.0, 0.5, 2.0, 1e-4])
    params = np.concatenate([np.zeros(d), c])
    for gamma in (1e-2, 1e-5, 1e-9):
        out = O.proximal_location_scale_entropy(params, d, O.MEANFIELD, gamma)[d:]
        assert np.array_equal(out, c + (np.sqrt(c * c + 4 * gamma) - c) / 2.0)         # the literal expression, nothing else
        rt = np.sqrt(c * c + 4 * gamma)
        stable = np.where(c < 0, 2 * gamma / (rt - c), c + (rt - c) / 2.0)             # what the device evaluates
        assert np.all(stable > 0.0) and np.all(out > 0.0)
        assert np.allclose(out, stable, rtol=1e-6, atol=1e-16)
        neg = c < 0                                                                    # (c > 0: c' - c is itself a cancelling difference in f64)
        assert np.allclose((stable - c)[neg], (gamma / stable)[neg], rtol=1e-9, atol=0.0)   # -1/c' + (c' - c)/gamma = 0
        assert np.allclose(stable * (stable - c), gamma, rtol=1e-5, atol=0.0)
    # the failure mode the stable form removes: the literal expression in Float32
    c32, g32 = np.float32(-1.0), np.float32(1e-9)
    assert c32 + (np.sqrt(c32 * c32 + np.float32(4) * g32) - c32) / np.float32(2) == 0.0


def test_logreg_batched_evaluation_equals_the_per_column_loop():
    """oracle.LogRegTarget.logdensity_and_gradient_batch (row-chunked matrix products, what makes BASELINE configs[2]'s n = 10^6 checkable) is
    the per-column restatement to rounding, for both variants, with f32 storage promoted chunk by chunk, and through estimate_gradient."""
    rng = np.random.default_rng(21)
    n, p, M = 1000, 9, 7
    X32 = (rng.normal(size=(n, p)) / 3).astype(np.float32)
    y = (rng.uniform(size=n) < 0.4).astype(np.uint8)
    Z = rng.normal(size=(p + 1, M)) * 0.5
    for variant, adj in (("logsigma_normal", 1.7), ("lognormal_exp_bijector", 1.0)):
        t = O.LogRegTarget(X32, y, variant, adj)
        tb = O.LogRegTarget(X32, y, variant, adj, keep_storage=True)
        ell, G = tb.logdensity_and_gradient_batch(Z, row_chunk=300)
        for m in range(M):
            l1, g1 = t.logdensity_and_gradient(Z[:, m])
            assert abs(ell[m] - l1) <= 1e-12 * abs(l1) and np.allclose(G[:, m], g1, rtol=1e-11, atol=1e-12)
        with pytest.raises(TypeError):
            tb.logdensity_and_gradient(Z[:, 0])
    d = p + 1
    params = np.concatenate([rng.normal(size=d) * 0.1, (0.6 * np.eye(d)).reshape(-1)])
    eps = rng.normal(size=(d, M))
    a = O.estimate_gradient(params, d, O.FULLRANK, t, eps, 0)
    b = O.estimate_gradient(params, d, O.FULLRANK, tb, eps, 0, batch_target=True)
    assert abs(a["value"] - b["value"]) <= 1e-12 * abs(a["value"]) and np.allclose(a["grad"], b["grad"], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("kind", ["logreg0", "logreg1", "funnel", "funnel_constrained"])
def test_second_order_targets_hessian_is_the_derivative_of_the_gradient(kind):
    """Round 6: `logdensity_gradient_and_hessian` of the logistic-regression and funnel restatements (what a LogDensityOrder{2} problem hands
    to gaussian_expectation_gradient_and_hessian!'s second-order branch, gauss_expected_grad_hess.jl:61-83), pinned by central differences of
    the (already pinned) gradients; symmetric."""
    rng = np.random.default_rng(31)
    if kind.startswith("logreg"):
        n, p = 60, 6
        X = rng.normal(size=(n, p)) / 2
        y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
        t = O.LogRegTarget(X, y, "logsigma_normal" if kind == "logreg0" else "lognormal_exp_bijector", 1.7 if kind == "logreg0" else 1.0)
        z = np.concatenate([rng.normal(size=p) * 0.4, [0.3]])
    elif kind == "funnel":
        t = O.FunnelStackedTarget(7, 1.5)
        z = np.concatenate([[0.4], rng.normal(size=6)])
    else:
        t = O.FunnelConstrainedTarget(7, 1.5)
        z = np.concatenate([[1.3], rng.normal(size=6)])
    v, g, H = t.logdensity_gradient_and_hessian(z)
    v1, g1 = t.logdensity_and_gradient(z)
    assert v == v1 and np.array_equal(g, g1) and np.allclose(H, H.T, rtol=0, atol=1e-14)
    h = 1e-5
    Hfd = np.stack([(t.logdensity_and_gradient(z + h * e)[1] - t.logdensity_and_gradient(z - h * e)[1]) / (2 * h) for e in np.eye(z.size)], axis=1)
    assert np.allclose(H, Hfd, rtol=1e-6, atol=1e-7)
