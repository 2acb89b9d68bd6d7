"""GPU parity tests proper: libmivi (through the C ABI) vs the CPU oracle on identical eps.

eps is read back from the device (mivi_sample) and handed to the oracle, so both sides consume the
same random stream (SURVEY.md 8c "identical RNG streams at the eps level"); the stream itself is
checked bit-exactly (Philox words) and to a few ulp (Box-Muller) in test_gpu_rng.py.

Tolerances (stated per the north star, fp32 compute vs fp64 oracle):
    objective value : rel 1e-5 (f32), 1e-12 (f64)
    gradient        : rel-L2 2e-5 (f32), 1e-11 (f64)
"""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED, OraclePlugin, make_family, make_problem, rel_err

pytestmark = pytest.mark.gpu

TOL = {np.float32: (1e-5, 2e-5), np.float64: (1e-12, 1e-11)}
ENTROPIES = [avi.ClosedFormEntropy(), avi.ClosedFormEntropyZeroGradient(), avi.MonteCarloEntropy(),
             avi.StickingTheLandingEntropy(), avi.StickingTheLandingEntropyZeroGradient()]


def run_case(d, M, family, kind, ent, dtype, idx=3, plugin=False, route=None):
    rng = np.random.default_rng(1234 + d + 7 * M)
    q, q_o = make_family(rng, d, family, dtype)
    prob, tgt = make_problem(rng, kind, d, dtype)
    if plugin:
        prob = OraclePlugin(tgt)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(dtype, family, d, M, ent.code, SEED)
    ctx.set_problem(prob)
    Z, eps = ctx.sample(params, idx)
    eps = eps.cpu().numpy().astype(np.float64)
    Z = Z.cpu().numpy().astype(np.float64)
    if kind.startswith("logreg") and dtype == np.float32 and not plugin and route is None:
        # the built-in logistic regression has two f32 kernel families, chosen by problem size: cover both
        ctx.close()
        run_case(d, M, family, kind, ent, dtype, idx, plugin, route=1)
        return run_case(d, M, family, kind, ent, dtype, idx, plugin, route=2)
    if route is not None:
        ctx.set_logreg_route(route)
    value, grad = ctx.estimate_gradient(params, idx)
    value = float(value.item())
    grad = grad.cpu().numpy().astype(np.float64)
    ref = O.estimate_gradient(O.destructure(q_o), d, family, tgt, eps, ent.code)
    vt, gt = TOL[dtype]
    assert rel_err(Z, ref["Z"]) < (1e-6 if dtype == np.float32 else 1e-14)
    assert abs(value - ref["value"]) <= vt * abs(ref["value"]), (value, ref["value"])
    gscale = np.linalg.norm(ref["grad"])
    assert np.linalg.norm(grad - ref["grad"]) <= gt * max(gscale, 1.0), (rel_err(grad, ref["grad"]))
    if family == avi.FULLRANK:  # structural zeros above the diagonal
        gC = grad[d:].reshape(d, d, order="F")
        assert np.all(np.triu(gC, 1) == 0.0)
    # partials route (multi-GPU building block) must agree with the fused route
    part = ctx.estimate_partials(params, idx)
    v2, g2 = ctx.finalize(params, part)
    assert abs(float(v2.item()) - ref["value"]) <= 2 * vt * abs(ref["value"])
    assert np.linalg.norm(g2.cpu().numpy() - ref["grad"]) <= 2 * gt * max(gscale, 1.0)
    assert rel_err(part.cpu().numpy(), ref["partials"]) < (5e-6 if dtype == np.float32 else 1e-12)
    ctx.close()
    return value, grad


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("ent", ENTROPIES, ids=lambda e: type(e).__name__)
@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
def test_diag_gauss_all_estimators(family, ent, dtype):
    run_case(40, 24, family, "diag", ent, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["dense", "logreg0", "logreg1", "funnel"])
@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
def test_targets(family, kind, dtype):
    for ent in (avi.ClosedFormEntropy(), avi.StickingTheLandingEntropy()):
        run_case(33, 17, family, kind, ent, dtype)


@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
def test_plugin_callback_route(family):
    """Generic LogDensityProblems plugin: batched logdensity_and_gradient over the columns of Z
    (src/mixedad_logdensity.jl:23-34 seam)."""
    run_case(12, 9, family, "logreg1", avi.ClosedFormEntropy(), np.float64, plugin=True)
    run_case(12, 9, family, "dense", avi.StickingTheLandingEntropy(), np.float32, plugin=True)


@pytest.mark.parametrize("family,d,M", [(avi.MEANFIELD, 1024, 256), (avi.FULLRANK, 256, 128), (avi.FULLRANK, 1024, 256),
                                        (avi.MEANFIELD, 5, 1), (avi.FULLRANK, 5, 1), (avi.FULLRANK, 70, 300),
                                        (avi.MEANFIELD, 3, 1000)])
def test_sizes_including_ragged(family, d, M):
    """BASELINE sizes (C2, north star) and ragged / minimal shapes (d, M not multiples of the tile)."""
    run_case(d, M, family, "diag", avi.ClosedFormEntropy(), np.float32)
    if d <= 256:
        run_case(d, M, family, "dense", avi.MonteCarloEntropy(), np.float32)


def test_stl_zero_gradient_known_answer():
    """Reference known answer: STL gradient == 0 when q == pi, for any M
    (test/algorithms/klminrepgraddescent.jl:66-87, atol 1e-5)."""
    d = 5
    for family in (avi.MEANFIELD, avi.FULLRANK):
        for M in (1, 10):
            mu = np.full(d, 5.0)
            if family == avi.MEANFIELD:
                q = avi.MeanFieldGaussian(mu, np.full(d, 0.3))
                prob = avi.DiagNormalProblem(mu, np.full(d, 0.3))
            else:
                L = 0.3 * np.eye(d) + 0.05 * np.tril(np.ones((d, d)), -1)
                q = avi.FullRankGaussian(mu, L)
                prob = avi.DenseNormalProblem(mu, L)
            params, _ = avi.destructure(q)
            ctx = avi.MiviContext(np.float64, family, d, M, avi.StickingTheLandingEntropy.code, SEED)
            ctx.set_problem(prob)
            _, grad = ctx.estimate_gradient(params, 0)
            assert np.linalg.norm(grad.cpu().numpy()) < 1e-5
            ctx.close()


def test_estimate_objective_at_optimum_is_zero():
    """estimate_objective(q = pi, n_samples = 10^5) ~ 0, atol 1e-2
    (test/algorithms/klminrepgraddescent.jl:36-37; default monitor entropy = MonteCarloEntropy)."""
    d = 5
    mu = np.full(d, 5.0)
    q = avi.MeanFieldGaussian(mu, np.full(d, 0.3))
    prob = avi.DiagNormalProblem(mu, np.full(d, 0.3))
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), operator=avi.ClipScale())
    rng = avi.PhiloxRNG(SEED)
    for n in (None, 1, 3):
        assert np.isfinite(avi.estimate_objective(rng, alg, q, prob, n_samples=n))
    assert abs(avi.estimate_objective(rng, alg, q, prob, n_samples=10 ** 5)) < 1e-2
    L = 0.3 * np.eye(d) + 0.05 * np.tril(np.ones((d, d)), -1)
    qf = avi.FullRankGaussian(mu, L)
    assert abs(avi.estimate_objective(rng, alg, qf, avi.DenseNormalProblem(mu, L), n_samples=10 ** 5)) < 1e-2


def test_estimate_objective_matches_oracle():
    rng = np.random.default_rng(5)
    for family in (avi.MEANFIELD, avi.FULLRANK):
        for kind in ("diag", "dense", "funnel"):
            d, M = 24, 50
            q, q_o = make_family(rng, d, family, np.float64)
            prob, tgt = make_problem(rng, kind, d, np.float64)
            params, _ = avi.destructure(q)
            ctx = avi.MiviContext(np.float64, family, d, M, 0, SEED)
            ctx.set_problem(prob)
            _, eps = ctx.sample(params, 11)
            for ent in (0, 2):
                v = float(ctx.estimate_objective(params, 11, n_samples=M, entropy=ent).item())
                ref = O.estimate_objective(q_o, tgt, eps.cpu().numpy(), ent)
                assert abs(v - ref) <= 1e-11 * abs(ref)
            ctx.close()


def test_determinism_bitwise():
    """Same seed => bitwise identical results (test/algorithms/klminrepgraddescent.jl:40-57)."""
    for family, d, M in ((avi.MEANFIELD, 1024, 256), (avi.FULLRANK, 256, 64)):
        rng = np.random.default_rng(9)
        q, _ = make_family(rng, d, family, np.float32)
        prob, _ = make_problem(rng, "diag", d, np.float32)
        params, _ = avi.destructure(q)
        outs = []
        for _ in range(2):
            ctx = avi.MiviContext(np.float32, family, d, M, 0, SEED)
            ctx.set_problem(prob)
            v, g = ctx.estimate_gradient(params, 5)
            outs.append((v.cpu().numpy().copy(), g.cpu().numpy().copy()))
            ctx.close()
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_graph_batched_estimates_match_single(dtype):
    for family, d, M in ((avi.MEANFIELD, 256, 64), (avi.FULLRANK, 128, 64), (avi.FULLRANK, 70, 19)):
        rng = np.random.default_rng(10)
        q, _ = make_family(rng, d, family, dtype)
        prob, _ = make_problem(rng, "diag", d, dtype)
        params, _ = avi.destructure(q)
        ctx = avi.MiviContext(dtype, family, d, M, 0, SEED)
        ctx.set_problem(prob)
        p = ctx.to_device(params)
        v, g = ctx.empty(1), ctx.empty(ctx.params_len)
        for count in (3, 9):   # a short batch runs as an eager chain, a longer one as a captured graph
            v1, g1 = ctx.estimate_gradient(p, 7 + count - 1)
            v1, g1 = v1.cpu().numpy().copy(), g1.cpu().numpy().copy()
            ctx.estimate_gradient_n(p, 7, count, v, g)   # estimates 7 .. 7 + count - 1; the last one is left in the buffers
            ctx.synchronize()
            assert np.array_equal(v.cpu().numpy(), v1) and np.array_equal(g.cpu().numpy(), g1), count
        ctx.close()


@pytest.mark.parametrize("ent", [3, 4])
def test_graph_batched_stl_estimates_reuse_the_solve_preparation(ent):
    """Sticking-the-landing estimators in batched calls: inside one call the parameters are fixed, so the parameter-only work is done once
    per call (the batch engine forms C^-T once -- DESIGN.md 3; with MIVI_FB_STL=0 only the chain's first estimate carries the solve's
    riders, bitwise: tests/test_gpu_ab_switches.py).  The last estimate of the batch equals the single call to rounding (2e-6: the
    engine multiplies by the inverse where a single call solves); a call after the parameters changed IN PLACE prepares again."""
    d, M = 256, 128
    rng = np.random.default_rng(21)
    q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
    prob, _ = make_problem(rng, "diag", d, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
    ctx.set_problem(prob)
    p = ctx.to_device(params).clone()
    v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    for scale in (1.0, 1.25):
        if scale != 1.0:
            p[d:] *= scale                                   # the scale matrix changes in place: same buffer, same cached graph
        for count in (4, 9):   # eager chain / captured graph
            v1, g1 = ctx.estimate_gradient(p, 7 + count - 1)
            v1, g1 = v1.cpu().numpy().copy(), g1.cpu().numpy().copy()
            ctx.estimate_gradient_n(p, 7, count, v, g)
            ctx.synchronize()
            assert abs(float(v.item()) - float(v1[0])) <= 1e-6 * abs(float(v1[0])), (scale, count)
            gb, gs = g.cpu().numpy().astype(np.float64), g1.astype(np.float64)
            assert np.linalg.norm(gb - gs) <= 2e-6 * max(1.0, np.linalg.norm(gs)), (scale, count, np.linalg.norm(gb - gs) / np.linalg.norm(gs))
    ctx.close()


def test_nonpositive_scale_and_nonfinite_status():
    """Error conventions: non-finite objective -> the reference's ErrorException (common.jl:83-89)."""
    d = 8
    q = avi.MeanFieldGaussian(np.zeros(d), np.ones(d))
    prob = avi.DiagNormalProblem(np.zeros(d), np.ones(d))
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), optimizer=avi.Descent(1e300), n_samples=4, operator=avi.IdentityOperator())
    with pytest.warns(UserWarning, match="IdentityOperator"):
        with pytest.raises(RuntimeError, match="diverged"):
            avi.optimize(avi.PhiloxRNG(1), alg, 5, prob, q)


@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK])
def test_device_entries_report_sticky_status_on_synchronize(family):
    """Device entries never synchronise: a non-positive scale diagonal (the DomainError ClipScale exists to prevent,
    clip_scale.jl:18-29) is flagged on the device and reported -- once -- by the next mivi_synchronize."""
    d, M = 32, 16
    rng = np.random.default_rng(5)
    q, _ = make_family(rng, d, family, np.float32)
    prob, _ = make_problem(rng, "diag", d, np.float32)
    params, _ = avi.destructure(q)
    params = params.copy()
    params[d + (3 if family == avi.MEANFIELD else 3 * d + 3)] = -0.25
    ctx = avi.MiviContext(np.float32, family, d, M, 0, SEED)
    ctx.set_problem(prob)
    ctx.estimate_gradient(params, 0)
    with pytest.raises(Exception, match="scale diagonal"):
        ctx.synchronize()
    ctx.synchronize()                      # flag was cleared by the read
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["diag", "dense"])
@pytest.mark.parametrize("d,M", [(64, 32), (70, 19), (256, 64)])
def test_fullrank_speculative_eps_prefetch(d, M, kind, dtype):
    """Single calls on the MFMA full-rank path let the VJP kernel of estimate idx also draw eps of idx + 1 and the
    following call skip its eps kernel.  Consecutive indices (prefetch hits), a jump (miss), the partials route and
    an interleaved sample() must all give exactly what a fresh context gives for the same (seed, idx)."""
    rng = np.random.default_rng(99 + d)
    q, _ = make_family(rng, d, avi.FULLRANK, dtype)
    prob, _ = make_problem(rng, kind, d, dtype)
    params, _ = avi.destructure(q)
    ent = avi.MonteCarloEntropy().code

    def fresh(idx):
        c = avi.MiviContext(dtype, avi.FULLRANK, d, M, ent, SEED)
        c.set_problem(prob)
        v, g = c.estimate_gradient(params, idx)
        out = (float(v.item()), g.cpu().numpy().copy())
        c.close()
        return out

    ctx = avi.MiviContext(dtype, avi.FULLRANK, d, M, ent, SEED)
    ctx.set_problem(prob)
    seq = [5, 6, 7, 3, 4, 4, 9]
    want = {i: fresh(i) for i in set(seq)}
    for n, idx in enumerate(seq):
        if n == 4:
            ctx.sample(params, 1234)          # touches the eps buffers: the pending prefetch must be dropped
        if n % 2 == 0:
            v, g = ctx.estimate_gradient(params, idx)
        else:
            v, g = ctx.finalize(params, ctx.estimate_partials(params, idx))
        v, g = float(v.item()), g.cpu().numpy()
        if n % 2 == 0:
            assert v == want[idx][0] and np.array_equal(g, want[idx][1]), (n, idx)
        else:                                  # partials route: same sums, normalised in a different kernel
            assert abs(v - want[idx][0]) <= (2e-6 if dtype == np.float32 else 1e-13) * abs(want[idx][0])
            assert np.linalg.norm(g - want[idx][1]) <= (2e-6 if dtype == np.float32 else 1e-13) * max(np.linalg.norm(want[idx][1]), 1.0)
    ctx.close()
