"""BASELINE.json configs at (or near) their full sizes.
C1  README LogReg n=1000, 32 features + sigma (theta in R^33), mean-field, n_mc=16, fp64 -- through the generic
    LogDensityProblems plugin route (host callback), checked against the oracle.
C2 / NS: covered by tests/test_gpu_parity.py::test_sizes_including_ragged (d=1024, M=256).
C3  hierarchical LogReg, D=512, full-rank, n_mc=128: oracle parity at n=20 000 and, at n=10^6, size-independent
    properties (row duplication with likeadj=1/2 leaves the estimate unchanged; f32 agrees with f64).
C4  the same D=512 full-rank family with n_mc = 1024 sharded 128 per GPU (8 shard contexts on one GPU, partials route):
    oracle parity of the summed partials at n=20 000, shard-sum == single 1024-sample estimate, and the n=10^6 duplication
    property on the sharded route.
C5  funnel d=2048 + Stacked bijector, mean-field, STL, 64 samples per GPU: oracle parity at full size and
    shard-sum == single estimate over the 8 x 64 = 512 global samples."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from advancedvi_jl_amd.distributed import ShardPlan
from oracle import oracle as O
from tests.helpers import SEED, OraclePlugin, rel_err

pytestmark = pytest.mark.gpu

# Gradient tolerance of the f32 logistic-regression path against the fp64 oracle.  The stated bar for f32 is 2e-5
# (tests/test_gpu_parity.py); the two data contractions sum 20 000 (10^6) products per output in f32 on the matrix cores
# (exact 3-way bf16 split, f32 accumulation in 16-row stages), measured rel-L2 error 0.6e-5 .. 1.2e-5 over seeds.
C3_GRAD_RTOL = 2e-5


def test_c1_readme_model_from_logdensity_alone_through_automivi(caplog):
    """BASELINE configs[0] as the reference's README runs it: the model declares LogDensityOrder{0}() and only `logdensity`
    (README.md:42-66), wrapped in the TransformedLogDensityProblem of README.md:91-119; `init` emits the reference's @info and
    differentiates through `logdensity` (repgradelbo.jl:50-57) -- here with the host's forward-mode provider, configs[0]'s "ForwardDiff
    on CPU".  Nothing hands the plugin a gradient: the oracle's closed-form LogReg gradient is only the CHECK."""
    import logging
    from tests.helpers import ReadmeLogReg, readme_bijector
    rng = np.random.default_rng(11)
    n, p = 1000, 32
    X = np.hstack([rng.normal(size=(n, p - 1)), np.ones((n, 1))])           # intercept column, README.md:139-140
    beta = rng.normal(size=p)
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    d, M = p + 1, 16
    model = ReadmeLogReg(X, y)
    prob = avi.TransformedProblem(model, readme_bijector(p))
    assert avi.capabilities(prob) < avi.LogDensityOrder(1)
    q = avi.MeanFieldGaussian(np.zeros(d), np.ones(d))                        # q0 of README.md:184
    params, re = avi.destructure(q)
    obj, ad = avi.RepGradELBO(M), avi.AutoMIVI()
    with caplog.at_level(logging.INFO, logger="advancedvi_jl_amd"):
        st = avi.init(avi.PhiloxRNG(SEED), obj, ad, q, prob, params, re)
    assert "directly differentiate through `LogDensityProblems.logdensity`" in caplog.text
    ctx = st.obj_ad_prep
    out = avi.DiffResult(ctx.empty(1), ctx.empty(ctx.params_len))
    _, eps = ctx.sample(params, 0)
    model.calls = 0
    _, _, info = avi.estimate_gradient_(avi.PhiloxRNG(SEED, 0), obj, ad, out, st, ctx.to_device(params), re)
    assert model.calls == M                                                   # one dual-number sweep (chunk 64 >= 33 partials) per column
    tgt = O.LogRegTarget(X, y, "lognormal_exp_bijector")
    ref = O.estimate_gradient(params, d, O.MEANFIELD, tgt, eps.cpu().numpy(), 0)
    assert abs(out.value() - ref["value"]) <= 1e-12 * abs(ref["value"])
    assert float(info["elbo"]) == -out.value()
    assert rel_err(out.gradient().cpu().numpy(), ref["grad"]) < 1e-11
    # estimate_objective only needs `logdensity` (repgradelbo.jl:112-118): the order-0 problem goes in unwrapped
    v = avi.estimate_objective(avi.PhiloxRNG(SEED, 7), obj, q, prob, n_samples=64)
    assert np.isfinite(v)
    # the README's run (README.md:150-200 in outline): KLMinRepGradDescent + ClipScale from `logdensity` alone; the ELBO improves
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=M, optimizer=avi.Adam(1e-2), operator=avi.ClipScale())
    _, info, _ = avi.optimize(avi.PhiloxRNG(SEED), alg, 60, prob, q)
    assert np.mean([float(i["elbo"]) for i in info[-10:]]) > np.mean([float(i["elbo"]) for i in info[:10]]) + 50.0
    ctx.close()


def test_reference_benchmark_target_order0_optimize():
    """bench/benchmarks.jl:25-94: `Dist(MvNormal(fill(5, 10), I))` declares LogDensityOrder{0}() (it has logdensity_and_gradient, which the
    reference therefore never calls), Float64, Adam(1e-3), ClipScale, both families and both entropies.  Through AutoMIVI unchanged: the
    trajectory equals the one of the built-in device target (same eps stream, f64) to rounding, and the problem's own gradient is unused."""
    from tests.helpers import BenchDist
    d = 10
    for fam in (avi.MEANFIELD, avi.FULLRANK):
        q0 = (avi.MeanFieldGaussian(np.zeros(d), np.ones(d)) if fam == avi.MEANFIELD else avi.FullRankGaussian(np.zeros(d), np.eye(d)))
        for ent in (avi.ClosedFormEntropy(), avi.StickingTheLandingEntropy()):
            alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), optimizer=avi.Adam(1e-3), entropy=ent, operator=avi.ClipScale())
            prob = BenchDist(d)
            qa, ia, _ = avi.optimize(avi.PhiloxRNG(SEED), alg, 40, prob, q0)
            qb, ib, _ = avi.optimize(avi.PhiloxRNG(SEED), alg, 40, avi.DiagNormalProblem(np.full(d, 5.0), np.ones(d)), q0)
            assert prob.grad_calls == 0
            assert np.allclose(qa.location, qb.location, rtol=1e-10, atol=1e-12)
            assert np.allclose(np.asarray(qa.scale), np.asarray(qb.scale), rtol=1e-10, atol=1e-12)
            assert np.allclose([float(i["elbo"]) for i in ia], [float(i["elbo"]) for i in ib], rtol=1e-10)


def test_c1_readme_logreg_plugin_route():
    rng = np.random.default_rng(11)
    n, p = 1000, 32
    X = np.hstack([rng.normal(size=(n, p - 1)), np.ones((n, 1))])           # intercept column, README.md:139-140
    beta = rng.normal(size=p)
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    tgt = O.LogRegTarget(X, y, "lognormal_exp_bijector")                     # README model + exp bijector on sigma
    d, M = p + 1, 16
    q = avi.MeanFieldGaussian(np.zeros(d), np.ones(d))                        # q0 of README.md:184
    params, _ = avi.destructure(q)
    plug = OraclePlugin(tgt)
    ctx = avi.MiviContext(np.float64, avi.MEANFIELD, d, M, 0, SEED)
    ctx.set_problem(plug)
    _, eps = ctx.sample(params, 0)
    v, g = ctx.estimate_gradient(params, 0)
    assert plug.calls == M                                                    # one logdensity_and_gradient per column
    ref = O.estimate_gradient(params, d, O.MEANFIELD, tgt, eps.cpu().numpy(), 0)
    assert abs(float(v.item()) - ref["value"]) <= 1e-12 * abs(ref["value"])
    assert rel_err(g.cpu().numpy(), ref["grad"]) < 1e-11
    # built-in device target gives the same answer as the plugin
    ctx2 = avi.MiviContext(np.float64, avi.MEANFIELD, d, M, 0, SEED)
    ctx2.set_problem(avi.LogRegProblem(X, y.astype(np.uint8), "lognormal_exp_bijector"))
    v2, g2 = ctx2.estimate_gradient(params, 0)
    assert abs(float(v2.item()) - ref["value"]) <= 1e-11 * abs(ref["value"])
    assert rel_err(g2.cpu().numpy(), ref["grad"]) < 1e-10
    ctx.close(); ctx2.close()


def _c3_data(n, rng):
    p = 511
    X = np.empty((n, p), dtype=np.float32)
    X[:, :510] = rng.normal(size=(n, 510)).astype(np.float32) / np.sqrt(510.0)   # SURVEY.md 8d synthetic inputs
    X[:, 510] = 1.0
    beta = rng.normal(size=p).astype(np.float32)
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(X @ beta)))).astype(np.uint8)
    return X, y


def test_c3_logreg_fullrank_oracle_parity_reduced_n():
    rng = np.random.default_rng(12)
    n, d, M = 20000, 512, 128
    X, y = _c3_data(n, rng)
    q = avi.FullRankGaussian(np.zeros(d, np.float32), 0.6 * np.eye(d, dtype=np.float32))   # docs/src/tutorials/basic.md:170
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
    ctx.set_problem(avi.LogRegProblem(X, y, "logsigma_normal", 1.0))
    assert ctx.logreg_kernels() == dict(mfma=True, logits_planes=True, xtr_planes=True)   # (what the size heuristic picks at C3's shape)
    _, eps = ctx.sample(params, 1)
    v, g = ctx.estimate_gradient(params, 1)
    ref = O.estimate_gradient(params.astype(np.float64), d, O.FULLRANK, O.LogRegTarget(X, y, "logsigma_normal", 1.0),
                              eps.cpu().numpy().astype(np.float64), 0)
    assert abs(float(v.item()) - ref["value"]) <= 1e-5 * abs(ref["value"])
    assert rel_err(g.cpu().numpy(), ref["grad"]) < C3_GRAD_RTOL
    ctx.close()


def test_c3_logreg_full_size_oracle_parity():
    """BASELINE configs[2] at its FULL size, n = 10^6 rows, D = 512, full-rank, n_mc = 128, against the fp64 oracle on identical eps (round 5's
    verdict: parity existed at n = 20 000 only).  The oracle's batched LogReg evaluation -- row-chunked f64 matrix products over the f32 data,
    equal to the per-column restatement to rounding (tests/test_oracle_pinning.py) -- finishes in seconds."""
    rng = np.random.default_rng(15)
    n, d, M = 1_000_000, 512, 128
    X, y = _c3_data(n, rng)
    q = avi.FullRankGaussian(np.zeros(d, np.float32), 0.6 * np.eye(d, dtype=np.float32))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
    ctx.set_problem(avi.LogRegProblem(X, y, "logsigma_normal", 1.0))
    assert ctx.logreg_kernels() == dict(mfma=True, logits_planes=True, xtr_planes=True)
    _, eps = ctx.sample(params, 4)
    v, g = ctx.estimate_gradient(params, 4)
    ref = O.estimate_gradient(params.astype(np.float64), d, O.FULLRANK, O.LogRegTarget(X, y, "logsigma_normal", 1.0, keep_storage=True),
                              eps.cpu().numpy().astype(np.float64), 0, batch_target=True)
    assert abs(float(v.item()) - ref["value"]) <= 1e-5 * abs(ref["value"])
    assert rel_err(g.cpu().numpy(), ref["grad"]) < C3_GRAD_RTOL
    ctx.close()


def test_c3_logreg_full_size_properties():
    """n = 10^6 rows, D = 512, full-rank, n_mc = 128 (BASELINE config 3): no CPU oracle at this size."""
    rng = np.random.default_rng(13)
    n_half, d, M = 500_000, 512, 128
    X, y = _c3_data(n_half, rng)
    q = avi.FullRankGaussian(np.zeros(d, np.float32), 0.6 * np.eye(d, dtype=np.float32))
    params, _ = avi.destructure(q)
    # (i) single copy, likeadj 1
    c1 = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
    c1.set_problem(avi.LogRegProblem(X, y, "logsigma_normal", 1.0))
    v1, g1 = c1.estimate_gradient(params, 2)
    v1, g1 = float(v1.item()), g1.cpu().numpy().astype(np.float64)
    c1.close()
    # (ii) rows duplicated (n = 10^6), likelihood scaled by n_data/n = 1/2: identical posterior => identical estimate
    X2, y2 = np.vstack([X, X]), np.concatenate([y, y])
    c2 = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
    c2.set_problem(avi.LogRegProblem(X2, y2, "logsigma_normal", 0.5))
    v2, g2 = c2.estimate_gradient(params, 2)
    assert np.isfinite(float(v2.item()))
    assert abs(float(v2.item()) - v1) <= 2e-5 * abs(v1)
    assert rel_err(g2.cpu().numpy(), g1) < 1e-4
    # structural zeros above the diagonal at full size
    gC = g2.cpu().numpy()[d:].reshape(d, d, order="F")
    assert np.all(np.triu(gC, 1) == 0.0)
    c2.close()


def _sharded_estimate(prob, params, d, M_total, R, idx, ent=0):
    """R shard contexts of M_total / R samples each (what R GPUs would run), partials summed in f64, finalized on shard 0."""
    plan = ShardPlan(M_total, R)
    total, ctxs = None, []
    for r in range(R):
        c = avi.MiviContext(np.float32, avi.FULLRANK, d, plan.count(r), ent, SEED, m_offset=plan.offset(r), m_total=M_total)
        c.set_problem(prob)
        part = c.estimate_partials(params, idx).double()
        total = part if total is None else total + part
        ctxs.append(c)
    v, g = ctxs[0].finalize(params, total.float())
    out = float(v.item()), g.cpu().numpy().astype(np.float64)
    for c in ctxs:
        c.close()
    return out


def test_c4_logreg_fullrank_sharded_1024_samples():
    """BASELINE config 4: D = 512 full-rank, n_mc = 1024 as 8 x 128 (partials route, RCCL's role played by a host sum)."""
    rng = np.random.default_rng(14)
    n, d, M_total, R = 20000, 512, 1024, 8
    X, y = _c3_data(n, rng)
    q = avi.FullRankGaussian(0.05 * rng.normal(size=d).astype(np.float32),
                             (0.6 * np.eye(d) + np.tril(rng.normal(size=(d, d)) * 0.01, -1)).astype(np.float32))
    params, _ = avi.destructure(q)
    prob = avi.LogRegProblem(X, y, "logsigma_normal", 1.0)
    v, g = _sharded_estimate(prob, params, d, M_total, R, 5)
    full = avi.MiviContext(np.float32, avi.FULLRANK, d, M_total, 0, SEED)
    full.set_problem(prob)
    _, eps = full.sample(params, 5)
    vf, gf = full.estimate_gradient(params, 5)
    ref = O.estimate_gradient(params.astype(np.float64), d, O.FULLRANK, O.LogRegTarget(X, y, "logsigma_normal", 1.0),
                              eps.cpu().numpy().astype(np.float64), 0)
    for vv, gg in ((v, g), (float(vf.item()), gf.cpu().numpy())):
        assert abs(vv - ref["value"]) <= 1e-5 * abs(ref["value"])
        assert rel_err(gg, ref["grad"]) < C3_GRAD_RTOL
    # the shard sum is the single estimate (same eps stream by construction), far inside the oracle tolerance
    assert rel_err(g, gf.cpu().numpy()) < 5e-6
    assert np.all(np.triu(g[d:].reshape(d, d, order="F"), 1) == 0.0)
    full.close()


def test_c4_sharded_full_size_duplication_property():
    """n = 10^6 rows on the sharded route (2 of the 8 shards' worth of contexts keep the run short: 4 x 256 samples):
    duplicating every row with likeadj = 1/2 leaves the estimate unchanged."""
    rng = np.random.default_rng(15)
    n_half, d, M_total, R = 500_000, 512, 1024, 4
    X, y = _c3_data(n_half, rng)
    q = avi.FullRankGaussian(np.zeros(d, np.float32), 0.6 * np.eye(d, dtype=np.float32))
    params, _ = avi.destructure(q)
    v1, g1 = _sharded_estimate(avi.LogRegProblem(X, y, "logsigma_normal", 1.0), params, d, M_total, R, 6)
    X2, y2 = np.vstack([X, X]), np.concatenate([y, y])
    v2, g2 = _sharded_estimate(avi.LogRegProblem(X2, y2, "logsigma_normal", 0.5), params, d, M_total, R, 6)
    assert np.isfinite(v2)
    assert abs(v2 - v1) <= 2e-5 * abs(v1)
    assert rel_err(g2, g1) < 1e-4


def test_c5_funnel_stl_full_size_and_sharding():
    d, M_total, R = 2048, 512, 8                  # BASELINE config 5: 64 samples per GPU on 8 GPUs
    q = avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32))
    params, _ = avi.destructure(q)
    prob = avi.FunnelProblem(d, 1.5)
    ent = avi.StickingTheLandingEntropy.code
    plan = ShardPlan(M_total, R)
    total = None
    ctxs = []
    for r in range(R):
        c = avi.MiviContext(np.float32, avi.MEANFIELD, d, plan.count(r), ent, SEED, m_offset=plan.offset(r), m_total=M_total)
        c.set_problem(prob)
        part = c.estimate_partials(params, 4).double()
        total = part if total is None else total + part
        ctxs.append(c)
    v, g = ctxs[0].finalize(params, total.float())
    full = avi.MiviContext(np.float32, avi.MEANFIELD, d, M_total, ent, SEED)
    full.set_problem(prob)
    _, eps = full.sample(params, 4)
    vf, gf = full.estimate_gradient(params, 4)
    ref = O.estimate_gradient(params.astype(np.float64), d, O.MEANFIELD, O.FunnelStackedTarget(d, 1.5),
                              eps.cpu().numpy().astype(np.float64), ent)
    for vv, gg in ((v, g), (vf, gf)):
        assert abs(float(vv.item()) - ref["value"]) <= 1e-5 * abs(ref["value"])
        assert rel_err(gg.cpu().numpy(), ref["grad"]) < 5e-5
    for c in ctxs:
        c.close()
    full.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("d,M,ent", [(2048, 64, 3), (2048, 64, 0), (7, 5, 3), (300, 200, 4), (64, 1024, 2)])
def test_c5_launch_free_batch_equals_single_calls(d, M, ent, dtype):
    """mivi_estimate_gradient_n on the fused funnel target (config 5's shard) runs all estimates inside one launch + one finishing
    launch (k_mf_funnel_loop / _value); value and gradient of the LAST estimate must be those of a single call (bitwise outside the
    funnel's row 0), for one wave per workgroup (n_mc <= 64) and four, and the result must still be the oracle's."""
    from oracle import oracle as O
    q = avi.MeanFieldGaussian((0.1 * np.arange(d) / d).astype(dtype), np.full(d, 0.8, dtype))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(dtype, avi.MEANFIELD, d, M, ent, SEED)
    ctx.set_problem(avi.FunnelProblem(d, 1.5))
    p = ctx.to_device(params)
    v, g = ctx.empty(1), ctx.empty(ctx.params_len).fill_(float("nan"))
    n = 9
    ctx.estimate_gradient_n(p, 30, n, v, g)
    ctx.synchronize()
    v1, g1 = ctx.estimate_gradient(p, 30 + n - 1)
    # every row but the funnel's row 0 is the same arithmetic in both kernels (bitwise); row 0 (d/dmu_0, d/dsigma_0) and the value collect
    # the other rows' sums through the loop's finishing kernel in another order than the single call's: an ulp or two (1 batch in ~10)
    ga, gb = g.cpu().numpy(), g1.cpu().numpy()
    rest = np.ones(ga.shape[0], bool)
    rest[[0, d]] = False
    assert np.array_equal(ga[rest], gb[rest])
    ulp = 4 * (np.finfo(dtype).eps)
    assert np.all(np.abs(ga[~rest] - gb[~rest]) <= ulp * np.abs(gb[~rest]))
    assert abs(float(v.item()) - float(v1.item())) <= ulp * abs(float(v1.item()))
    _, eps = ctx.sample(p, 30 + n - 1)
    ref = O.estimate_gradient(params.astype(np.float64), d, avi.MEANFIELD, O.FunnelStackedTarget(d, 1.5), eps.cpu().numpy().astype(np.float64), ent)
    vt, gt = (1e-5, 2e-5) if dtype == np.float32 else (1e-12, 1e-11)
    assert abs(float(v.item()) - ref["value"]) <= vt * max(1.0, abs(ref["value"]))
    assert np.linalg.norm(g.cpu().numpy() - ref["grad"]) <= gt * max(1.0, np.linalg.norm(ref["grad"]))
    ctx.close()


@pytest.mark.parametrize("n,p,M,family,variant", [(1000, 127, 128, avi.FULLRANK, "logsigma_normal"), (3001, 200, 256, avi.FULLRANK, "lognormal_exp_bijector"),
                                                   (777, 300, 128, avi.MEANFIELD, "logsigma_normal"), (4096, 40, 384, avi.FULLRANK, "logsigma_normal"),
                                                   (130, 1000, 128, avi.MEANFIELD, "logsigma_normal")])
def test_logreg_operand_planes_at_odd_shapes(n, p, M, family, variant):
    """The operand-plane route of the two data contractions (k_lr_logits_planes / k_lr_xtr_planes: data sets of 10^5 elements and more, n_mc a
    multiple of 128) at shapes the BASELINE configs do not visit: ragged last row tile, one / several feature groups with a partial last one,
    two and three sample groups, few rows with many features -- value and gradient against the fp64 oracle on the device's own eps."""
    rng = np.random.default_rng(n + p)
    d = p + 1
    X = (rng.normal(size=(n, p)) / np.sqrt(p)).astype(np.float32)
    X[:, p - 1] = 1.0
    y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    mu = (0.1 * rng.normal(size=d)).astype(np.float32)
    if family == avi.FULLRANK:
        C = (0.5 * np.eye(d) + np.tril(rng.normal(size=(d, d)) * (0.05 / np.sqrt(d)), -1)).astype(np.float32)
        q = avi.FullRankGaussian(mu, C)
    else:
        q = avi.MeanFieldGaussian(mu, rng.uniform(0.3, 0.7, size=d).astype(np.float32))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, family, d, M, 0, SEED)
    ctx.set_problem(avi.LogRegProblem(X, y, variant, 1.3))
    ctx.set_logreg_route(1)   # the matrix-core family of kernels whatever the size heuristic says
    assert ctx.logreg_kernels() == dict(mfma=True, logits_planes=True, xtr_planes=True)
    _, eps = ctx.sample(params, 4)
    v, g = ctx.estimate_gradient(params, 4)
    ref = O.estimate_gradient(params.astype(np.float64), d, family, O.LogRegTarget(X, y, variant, 1.3), eps.cpu().numpy().astype(np.float64), 0)
    assert abs(float(v.item()) - ref["value"]) <= 1e-5 * abs(ref["value"]), (float(v.item()), ref["value"])
    assert rel_err(g.cpu().numpy(), ref["grad"]) < C3_GRAD_RTOL
    # value only (no residual planes are written) agrees with the estimate's value
    vo = ctx.estimate_objective(params, 4, n_samples=0, entropy=0)
    assert abs(float(vo.item()) - float(v.item())) <= 2e-6 * abs(float(v.item()))
    ctx.close()


def _plane_shapes(k, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(k):
        p = int(rng.choice([17, 31, 33, 63, 64, 65, 95, 97, 129, 160, 255, 300, 511, 700]))
        n = int(max(100_000 // p + 1, rng.integers(130, 6000)))
        M = int(rng.choice([128, 256]))
        out.append((n, p, M))
    return out


@pytest.mark.parametrize("n,p,M", _plane_shapes(16, 20260930))
def test_logreg_operand_planes_random_shapes(n, p, M):
    """Random (rows, features, samples) over the operand-plane route -- feature counts on both sides of every 32- and 128-wide boundary (a
    p = 63 data set once read past the end of X's second-orientation planes) -- against the fp64 oracle."""
    rng = np.random.default_rng(n * 7 + p)
    d = p + 1
    X = (rng.normal(size=(n, p)) / np.sqrt(p)).astype(np.float32)
    y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    q = avi.MeanFieldGaussian((0.1 * rng.normal(size=d)).astype(np.float32), rng.uniform(0.3, 0.7, size=d).astype(np.float32))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.MEANFIELD, d, M, 0, SEED)
    ctx.set_problem(avi.LogRegProblem(X, y, "logsigma_normal", 0.9))
    ctx.set_logreg_route(1)
    assert ctx.logreg_kernels() == dict(mfma=True, logits_planes=True, xtr_planes=True)
    _, eps = ctx.sample(params, 2)
    v, g = ctx.estimate_gradient(params, 2)
    ref = O.estimate_gradient(params.astype(np.float64), d, avi.MEANFIELD, O.LogRegTarget(X, y, "logsigma_normal", 0.9), eps.cpu().numpy().astype(np.float64), 0)
    assert abs(float(v.item()) - ref["value"]) <= 1e-5 * abs(ref["value"]), (float(v.item()), ref["value"])
    assert rel_err(g.cpu().numpy(), ref["grad"]) < C3_GRAD_RTOL
    ctx.close()


def _intile_shapes(k, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(k):
        p = int(rng.choice([3, 7, 15, 16, 31, 33, 63, 65, 100, 127, 129, 255]))
        n = int(min(99_000 // p, rng.integers(17, 3000)))
        M = int(rng.choice([16, 32, 48, 64, 100, 128, 160, 256]))
        out.append((n, p, M))
    return out


@pytest.mark.parametrize("n,p,M", _intile_shapes(20, 20260931))
def test_logreg_matrix_core_kernels_without_planes_random_shapes(n, p, M):
    """The matrix-core kernels that split X in the tile (k_lr_logits_f16x2[_part] / k_lr_xtr_f16x2: data sets below 10^5 elements, or sample
    counts that are not multiples of 128) over random shapes, pinned with set_logreg_route(1), against the fp64 oracle."""
    rng = np.random.default_rng(n * 11 + p + M)
    d = p + 1
    X = (rng.normal(size=(n, p)) / np.sqrt(p)).astype(np.float32)
    y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    q, q_o = make_family_fr(rng, d)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
    ctx.set_problem(avi.LogRegProblem(X, y, "lognormal_exp_bijector", 1.0))
    ctx.set_logreg_route(1)
    k = ctx.logreg_kernels()
    assert k["mfma"] and not k["logits_planes"]
    _, eps = ctx.sample(params, 6)
    v, g = ctx.estimate_gradient(params, 6)
    ref = O.estimate_gradient(params.astype(np.float64), d, avi.FULLRANK, O.LogRegTarget(X, y, "lognormal_exp_bijector", 1.0), eps.cpu().numpy().astype(np.float64), 0)
    assert abs(float(v.item()) - ref["value"]) <= 2e-5 * max(abs(ref["value"]), 1.0), (float(v.item()), ref["value"])
    assert np.linalg.norm(g.cpu().numpy() - ref["grad"]) <= 4e-5 * max(np.linalg.norm(ref["grad"]), 1.0)
    ctx.close()


def make_family_fr(rng, d):
    mu = (0.1 * rng.normal(size=d)).astype(np.float32)
    C = (0.5 * np.eye(d) + np.tril(rng.normal(size=(d, d)) * (0.05 / np.sqrt(d)), -1)).astype(np.float32)
    return avi.FullRankGaussian(mu, C), None
