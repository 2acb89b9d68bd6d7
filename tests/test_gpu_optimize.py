"""The callers of the hot path on the GPU: `optimize` / `step` mirror (src/optimize.jl:42-94,
src/algorithms/common.jl:40-120) and the device-resident loop mivi_optimize_steps.  Restates the reference's
integration tests (test/algorithms/klminrepgraddescent.jl, test/general/optimize.jl)."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED, make_family

pytestmark = pytest.mark.gpu


def normal_meanfield(dtype=np.float64, d=5):
    """test/models/normal.jl:56-75"""
    mu, sig = np.full(d, 5.0, dtype), np.full(d, 0.3, dtype)
    return avi.DiagNormalProblem(mu, sig), mu, sig


def normal_fullrank(dtype=np.float64, d=5):
    """test/models/normal.jl:36-54"""
    mu = np.full(d, 5.0, dtype)
    L = (0.3 * np.eye(d)).astype(dtype)
    return avi.DenseNormalProblem(mu, L), mu, L


@pytest.mark.parametrize("n_samples", [1, 10])
def test_basic_runs(n_samples):
    """klminrepgraddescent.jl:9-13"""
    prob, _, _ = normal_meanfield()
    q0 = avi.MeanFieldGaussian(np.zeros(5), np.ones(5))
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=n_samples, operator=avi.ClipScale())
    q, info, state = avi.optimize(avi.PhiloxRNG(1), alg, 1, prob, q0)
    assert len(info) == 1 and np.isfinite(info[0]["elbo"]) and state["iteration"] == 1


def test_callback_ordering():
    """klminrepgraddescent.jl:15-21"""
    prob, _, _ = normal_meanfield()
    q0 = avi.MeanFieldGaussian(np.zeros(5), np.ones(5))
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), operator=avi.ClipScale())
    _, info, _ = avi.optimize(avi.PhiloxRNG(1), alg, 10, prob, q0, callback=lambda iteration, **kw: {"iteration_check": iteration})
    assert [i["iteration_check"] for i in info] == list(range(1, 11))


@pytest.mark.parametrize("family", ["meanfield", "fullrank"])
def test_determinism_bitwise(family):
    """klminrepgraddescent.jl:40-57: same seed => identical q_out."""
    if family == "meanfield":
        prob, _, _ = normal_meanfield()
        q0 = avi.MeanFieldGaussian(np.zeros(5), np.ones(5))
    else:
        prob, _, _ = normal_fullrank()
        q0 = avi.FullRankGaussian(np.zeros(5), np.eye(5))
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), operator=avi.ClipScale())
    outs = []
    for _ in range(2):
        q, _, _ = avi.optimize(avi.PhiloxRNG(SEED), alg, 10, prob, q0)
        outs.append(q)
    assert np.array_equal(outs[0].location, outs[1].location) and np.array_equal(outs[0].scale, outs[1].scale)


def test_warm_start_equals_single_run():
    """test/general/optimize.jl:30-40: optimize(T1) then optimize(T2; state) == optimize(T1+T2), bitwise."""
    prob, _, _ = normal_meanfield()
    q0 = avi.MeanFieldGaussian(np.zeros(5), np.ones(5))
    mk = lambda: avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=4, optimizer=avi.Adam(1e-2), operator=avi.ClipScale(),
                                         averager=avi.NoAveraging())
    q_ref, _, _ = avi.optimize(avi.PhiloxRNG(SEED), mk(), 20, prob, q0)
    rng = avi.PhiloxRNG(SEED)
    alg = mk()
    _, _, st = avi.optimize(rng, alg, 10, prob, q0)
    q2, _, _ = avi.optimize(rng, alg, 10, prob, q0, state=st)
    assert np.array_equal(q_ref.location, q2.location) and np.array_equal(q_ref.scale, q2.scale)


@pytest.mark.parametrize("realtype", [np.float32, np.float64])
def test_type_stability(realtype):
    """klminrepgraddescent.jl:90-103"""
    prob, _, _ = normal_meanfield(realtype)
    q0 = avi.MeanFieldGaussian(np.zeros(5, realtype), np.ones(5, realtype))
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=10, operator=avi.ClipScale())
    q, info, _ = avi.optimize(avi.PhiloxRNG(2), alg, 1, prob, q0)
    assert q.location.dtype == realtype and q.scale.dtype == realtype


@pytest.mark.parametrize("entropy", [avi.ClosedFormEntropy(), avi.StickingTheLandingEntropy()], ids=["CFE", "STL"])
@pytest.mark.parametrize("family", ["meanfield", "fullrank"])
def test_convergence_host_loop(entropy, family):
    """klminrepgraddescent.jl:105-121: T = 1000, Descent(1e-3): distance to the optimum at least halves."""
    if family == "meanfield":
        prob, mu, scale = normal_meanfield()
        q0 = avi.MeanFieldGaussian(np.zeros(5), np.ones(5))
    else:
        prob, mu, scale = normal_fullrank()
        q0 = avi.FullRankGaussian(np.zeros(5), np.eye(5))
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), entropy=entropy, optimizer=avi.Descent(1e-3), operator=avi.ClipScale())
    q, _, _ = avi.optimize(avi.PhiloxRNG(3), alg, 1000, prob, q0)
    d0 = np.sum((q0.location - mu) ** 2) + np.sum((q0.scale - scale) ** 2)
    d1 = np.sum((q.location - mu) ** 2) + np.sum((q.scale - scale) ** 2)
    assert d1 <= d0 / 2


@pytest.mark.parametrize("shape", [(64, 48), (70, 35), (128, 128)], ids=["aligned", "ragged", "gen2"])
@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
@pytest.mark.parametrize("rule", [0, 1], ids=["descent", "adam"])
def test_device_resident_loop_matches_host_loop(family, rule, shape):
    """mivi_optimize_steps (mean-field: one launch-free kernel; full-rank: one hipGraph of n estimates with deferred
    value / prefetched eps and the optimiser step + ClipScale fused into the VJP epilogue) must reproduce, bitwise, the
    step-by-step sequence of separate launches.  (More than 32 samples per step: with fewer, the full-rank family on this target takes
    the row-separable loop -- test_fullrank_rows_loop, to rounding; its graph route at those shapes: test_gpu_ab_switches.py.)"""
    d, M = shape
    T = 12
    rng = np.random.default_rng(4)
    tm, ts = rng.normal(size=d).astype(np.float32), rng.uniform(0.5, 2, size=d).astype(np.float32)
    if family == avi.MEANFIELD:
        q0 = avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32))
    else:
        q0 = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
    p0, _ = avi.destructure(q0)
    eta = 1e-2
    ctx = avi.MiviContext(np.float32, family, d, M, 0, SEED)
    ctx.set_problem(avi.DiagNormalProblem(tm, ts))
    # host-driven sequence
    p = ctx.to_device(p0).clone()
    st = ctx.empty(2 * p.numel()).zero_()
    elbos = []
    for t in range(T):
        v, g = ctx.estimate_gradient(p, 100 + t)
        elbos.append(-float(v.item()))
        if rule == 0:
            ctx.descent_update(p, g, eta)
        else:
            ctx.adam_update(p, g, st, t + 1, eta)
        ctx.clip_scale(p, 1e-5)
    # device-resident loop
    p2 = ctx.to_device(p0).clone()
    st2 = ctx.empty(2 * p2.numel()).zero_()
    elbo = ctx.empty(T)
    ctx.optimize_steps(p2, st2 if rule == 1 else None, 100, 0, T, rule, eta, 1e-5, elbo)
    ctx.synchronize()
    assert np.array_equal(p.cpu().numpy(), p2.cpu().numpy())
    assert np.allclose(elbo.cpu().numpy(), np.array(elbos, dtype=np.float32), rtol=1e-6)
    # and the whole trajectory agrees with an independent restatement: oracle gradient (on the device's own eps) ->
    # numpy Descent / Adam (oracle.descent_step / adam_step, Optimisers.jl semantics) -> ClipScale, followed in f64
    x = p0.astype(np.float64)
    ost = (np.zeros_like(x), np.zeros_like(x))
    tgt = O.DiagNormalTarget(tm, ts)
    for t in range(3):
        _, eps = ctx.sample(x.astype(np.float32), 100 + t)
        ref = O.estimate_gradient(x.astype(np.float32).astype(np.float64), d, family, tgt, eps.cpu().numpy().astype(np.float64), 0)
        if rule == 0:
            x = O.descent_step(x, ref["grad"], eta)
        else:
            x, ost = O.adam_step(x, ref["grad"], ost, t + 1, eta)
        x = O.clip_scale(x, d, family, 1e-5)
    p3 = ctx.to_device(p0).clone()
    st3 = ctx.empty(2 * p3.numel()).zero_()
    ctx.optimize_steps(p3, st3 if rule == 1 else None, 100, 0, 3, rule, eta, 1e-5, ctx.empty(3))
    ctx.synchronize()
    got = p3.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - x)) <= 5e-6 * max(1.0, np.max(np.abs(x))), np.max(np.abs(got - x))
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape", [(10, 1), (10, 8), (32, 16), (5, 64), (17, 3), (32, 64)], ids=["reference-bench", "d10-m8", "d32-m16", "d5-m64", "ragged", "largest"])
@pytest.mark.parametrize("rule", [0, 1], ids=["descent", "adam"])
@pytest.mark.parametrize("ent", [0, 3, 2], ids=["CFE", "STL", "MC"])
def test_small_fullrank_loop(ent, rule, shape, dtype):
    """Small full-rank problems (d <= 32, n_mc <= 64; the reference's own benchmark grid, bench/benchmarks.jl:43-94: d = 10, one sample per
    step, full-rank, ClosedFormEntropy / StickingTheLandingEntropy, Adam, ClipScale): mivi_optimize_steps runs the whole loop in ONE
    workgroup (k_fr_small_loop).  Its sums are sequential fused multiply-adds, not the tile kernels' MFMA chains: the trajectory equals the
    step-by-step sequence of single calls + update + ClipScale launches to rounding (stated here), and the oracle's gradient + numpy rules."""
    d, M = shape
    T = 12
    rng = np.random.default_rng(11)
    tm, ts = rng.normal(size=d).astype(dtype), rng.uniform(0.5, 2, size=d).astype(dtype)
    C0 = (np.eye(d) + 0.1 * np.tril(rng.normal(size=(d, d)), -1)).astype(dtype)
    q0 = avi.FullRankGaussian(np.zeros(d, dtype), C0)
    p0, _ = avi.destructure(q0)
    eta = 1e-2
    ctx = avi.MiviContext(dtype, avi.FULLRANK, d, M, ent, SEED)
    ctx.set_problem(avi.DiagNormalProblem(tm, ts))
    p = ctx.to_device(p0).clone()
    st = ctx.empty(2 * p.numel()).zero_()
    elbos = []
    for t in range(T):
        v, g = ctx.estimate_gradient(p, 70 + t)
        elbos.append(-float(v.item()))
        if rule == 0:
            ctx.descent_update(p, g, eta)
        else:
            ctx.adam_update(p, g, st, t + 1, eta)
        ctx.clip_scale(p, 1e-5)
    p2 = ctx.to_device(p0).clone()
    st2 = ctx.empty(2 * p2.numel()).zero_()
    elbo = ctx.empty(T)
    ctx.optimize_steps(p2, st2 if rule == 1 else None, 70, 0, 7, rule, eta, 1e-5, elbo[:7])
    ctx.optimize_steps(p2, st2 if rule == 1 else None, 77, 7, T - 7, rule, eta, 1e-5, elbo[7:])   # a second call continues (indices, Adam's t)
    ctx.synchronize()
    tol = 3e-5 if dtype == np.float32 else 1e-11
    a, b = p.cpu().numpy().astype(np.float64), p2.cpu().numpy().astype(np.float64)
    low = np.concatenate([np.ones(d, bool), np.tril(np.ones((d, d), bool)).T.reshape(-1)])      # [mu; vec C column-major]: entries on / below the diagonal
    assert np.max(np.abs(a[low] - b[low])) <= tol * max(1.0, np.max(np.abs(a[low]))), np.max(np.abs(a[low] - b[low]))
    assert np.array_equal(b[~low], p0.astype(np.float64)[~low])                                 # nothing above the diagonal is touched
    assert np.allclose(elbo.cpu().numpy().astype(np.float64), np.array(elbos), rtol=5e-5 if dtype == np.float32 else 1e-10, atol=1e-4 if dtype == np.float32 else 1e-10)
    if rule == 1:
        sa, sb = st.cpu().numpy().astype(np.float64), st2.cpu().numpy().astype(np.float64)
        low2 = np.concatenate([low, low])
        assert np.max(np.abs(sa[low2] - sb[low2])) <= tol * max(1.0, np.max(np.abs(sa[low2])))
    # independent restatement: oracle gradient on the device's own eps -> numpy rules -> ClipScale, in f64
    x = p0.astype(np.float64)
    ost = (np.zeros_like(x), np.zeros_like(x))
    tgt = O.DiagNormalTarget(tm, ts)
    for t in range(3):
        _, eps = ctx.sample(x.astype(dtype), 70 + t)
        ref = O.estimate_gradient(x.astype(dtype).astype(np.float64), d, avi.FULLRANK, tgt, eps.cpu().numpy().astype(np.float64), ent)
        if rule == 0:
            x = O.descent_step(x, ref["grad"], eta)
        else:
            x, ost = O.adam_step(x, ref["grad"], ost, t + 1, eta)
        x = O.clip_scale(x, d, avi.FULLRANK, 1e-5)
    p3 = ctx.to_device(p0).clone()
    st3 = ctx.empty(2 * p3.numel()).zero_()
    ctx.optimize_steps(p3, st3 if rule == 1 else None, 70, 0, 3, rule, eta, 1e-5, ctx.empty(3))
    ctx.synchronize()
    got = p3.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got[low] - x[low])) <= (2e-5 if dtype == np.float32 else 1e-11) * max(1.0, np.max(np.abs(x[low]))), np.max(np.abs(got[low] - x[low]))
    ctx.close()


@pytest.mark.parametrize("shape", [(64, 128), (128, 256), (320, 256), (512, 128), (1024, 256)], ids=["one-block-pair", "d128", "ragged-runs", "d512-m128", "north-star"])
@pytest.mark.parametrize("rule", [0, 1], ids=["descent", "adam"])
@pytest.mark.parametrize("ent", [0, 2], ids=["CFE", "MC"])
def test_fullrank_graph_loop_is_the_step_by_step_sequence(ent, rule, shape):
    """The north-star shape class (full-rank family, 128 / 256 samples per step, diagonal-Gaussian target): mivi_optimize_steps runs the hipGraph of
    launches with the optimiser step fused into the VJP epilogue.  Parameters and optimiser state must equal, BITWISE, the step-by-step sequence of
    single calls + update + ClipScale launches (two calls that continue each other); the ELBO record to rounding; nothing above the diagonal is
    touched.  (Round 4's persistent tile-owning kernel for this class was slower than this route and is kept under tools/experiments/.)"""
    d, M = shape
    T = 11
    rng = np.random.default_rng(17)
    tm, ts = rng.normal(size=d).astype(np.float32), rng.uniform(0.5, 2, size=d).astype(np.float32)
    C0 = (np.eye(d) + (0.3 / np.sqrt(d)) * np.tril(rng.normal(size=(d, d)), -1)).astype(np.float32)
    q0 = avi.FullRankGaussian((0.1 * rng.normal(size=d)).astype(np.float32), C0)
    p0, _ = avi.destructure(q0)
    eta = 1e-2
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
    ctx.set_problem(avi.DiagNormalProblem(tm, ts))
    p = ctx.to_device(p0).clone()
    st = ctx.empty(2 * p.numel()).zero_()
    elbos = []
    for t in range(T):
        v, g = ctx.estimate_gradient(p, 30 + t)
        elbos.append(-float(v.item()))
        if rule == 0:
            ctx.descent_update(p, g, eta)
        else:
            ctx.adam_update(p, g, st, t + 1, eta)
        ctx.clip_scale(p, 1e-5)
    p2 = ctx.to_device(p0).clone()
    st2 = ctx.empty(2 * p2.numel()).zero_()
    elbo = ctx.empty(T)
    ctx.optimize_steps(p2, st2 if rule == 1 else None, 30, 0, 6, rule, eta, 1e-5, elbo[:6])
    ctx.optimize_steps(p2, st2 if rule == 1 else None, 36, 6, T - 6, rule, eta, 1e-5, elbo[6:])   # a second call continues (indices, Adam's t)
    ctx.synchronize()
    assert np.array_equal(p.cpu().numpy(), p2.cpu().numpy())
    if rule == 1:
        assert np.array_equal(st.cpu().numpy(), st2.cpu().numpy())
    assert np.allclose(elbo.cpu().numpy().astype(np.float64), np.array(elbos), rtol=2e-6, atol=1e-4)
    low = np.concatenate([np.ones(d, bool), np.tril(np.ones((d, d), bool)).T.reshape(-1)])      # [mu; vec C column-major]: entries on / below the diagonal
    assert np.array_equal(p2.cpu().numpy()[~low], p0[~low])
    # independent restatement: oracle gradient on the device's own eps -> numpy rules -> ClipScale, in f64
    x = p0.astype(np.float64)
    ost = (np.zeros_like(x), np.zeros_like(x))
    tgt = O.DiagNormalTarget(tm, ts)
    for t in range(2):
        _, eps = ctx.sample(x.astype(np.float32), 30 + t)
        ref = O.estimate_gradient(x.astype(np.float32).astype(np.float64), d, avi.FULLRANK, tgt, eps.cpu().numpy().astype(np.float64), ent)
        if rule == 0:
            x = O.descent_step(x, ref["grad"], eta)
        else:
            x, ost = O.adam_step(x, ref["grad"], ost, t + 1, eta)
        x = O.clip_scale(x, d, avi.FULLRANK, 1e-5)
    p3 = ctx.to_device(p0).clone()
    st3 = ctx.empty(2 * p3.numel()).zero_()
    ctx.optimize_steps(p3, st3 if rule == 1 else None, 30, 0, 2, rule, eta, 1e-5, ctx.empty(2))
    ctx.synchronize()
    got = p3.cpu().numpy().astype(np.float64)
    tol = 2e-5 if rule == 0 else 1e-4
    assert np.max(np.abs(got[low] - x[low])) <= tol * max(1.0, np.max(np.abs(x[low]))), np.max(np.abs(got[low] - x[low]))
    ctx.close()


@pytest.mark.parametrize("shape", [(1024, 1), (256, 8), (1024, 16), (512, 32), (130, 2), (129, 4), (1126, 4), (1000, 12), (9, 3)],
                         ids=["one-sample", "d256-m8", "d1024-m16", "d512-m32", "even-ragged", "odd-middle-row", "widest", "m12", "tiny"])
@pytest.mark.parametrize("rule", [0, 1], ids=["descent", "adam"])
@pytest.mark.parametrize("ent", [0, 2], ids=["CFE", "MC"])
def test_fullrank_rows_loop(ent, rule, shape):
    """Full-rank family, diagonal-Gaussian target, few samples per step (n_mc <= 32; the reference's default is n_samples = 1,
    src/algorithms/klminrepgraddescent.jl): every row of (mu, C) needs only eps from the rest of the problem, so mivi_optimize_steps runs
    ONE kernel whose workgroups own row pairs for all steps (k_fr_rows_loop).  Its sums are sequential fused multiply-adds, not the tile
    kernels' MFMA chains: the trajectory equals the step-by-step sequence of single calls + update + ClipScale launches to rounding
    (stated here), and the oracle's gradient + numpy rules."""
    d, M = shape
    T = 9
    rng = np.random.default_rng(13)
    tm, ts = rng.normal(size=d).astype(np.float32), rng.uniform(0.5, 2, size=d).astype(np.float32)
    C0 = (np.eye(d) + (0.3 / np.sqrt(d)) * np.tril(rng.normal(size=(d, d)), -1)).astype(np.float32)
    q0 = avi.FullRankGaussian(np.zeros(d, np.float32), C0)
    p0, _ = avi.destructure(q0)
    eta = 1e-2
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
    ctx.set_problem(avi.DiagNormalProblem(tm, ts))
    p = ctx.to_device(p0).clone()
    st = ctx.empty(2 * p.numel()).zero_()
    elbos = []
    for t in range(T):
        v, g = ctx.estimate_gradient(p, 40 + t)
        elbos.append(-float(v.item()))
        if rule == 0:
            ctx.descent_update(p, g, eta)
        else:
            ctx.adam_update(p, g, st, t + 1, eta)
        ctx.clip_scale(p, 1e-5)
    p2 = ctx.to_device(p0).clone()
    st2 = ctx.empty(2 * p2.numel()).zero_()
    elbo = ctx.empty(T)
    ctx.optimize_steps(p2, st2 if rule == 1 else None, 40, 0, 5, rule, eta, 1e-5, elbo[:5])
    ctx.optimize_steps(p2, st2 if rule == 1 else None, 45, 5, T - 5, rule, eta, 1e-5, elbo[5:])   # a second call continues (indices, Adam's t)
    ctx.synchronize()
    a, b = p.cpu().numpy().astype(np.float64), p2.cpu().numpy().astype(np.float64)
    low = np.concatenate([np.ones(d, bool), np.tril(np.ones((d, d), bool)).T.reshape(-1)])      # [mu; vec C column-major]: entries on / below the diagonal
    tol = 3e-5 if rule == 0 else 1e-4   # (Adam's first steps divide by sqrt(v) ~ |g|: an entry whose gradient is rounding noise moves by +-eta either way)
    assert np.max(np.abs(a[low] - b[low])) <= tol * max(1.0, np.max(np.abs(a[low]))), np.max(np.abs(a[low] - b[low]))
    assert np.array_equal(b[~low], p0.astype(np.float64)[~low])                                 # nothing above the diagonal is touched
    assert np.allclose(elbo.cpu().numpy().astype(np.float64), np.array(elbos), rtol=5e-5, atol=1e-3)
    if rule == 1:
        sa, sb = st.cpu().numpy().astype(np.float64), st2.cpu().numpy().astype(np.float64)
        low2 = np.concatenate([low, low])
        assert np.max(np.abs(sa[low2] - sb[low2])) <= tol * max(1.0, np.max(np.abs(sa[low2])))
    # independent restatement: oracle gradient on the device's own eps -> numpy rules -> ClipScale, in f64
    x = p0.astype(np.float64)
    ost = (np.zeros_like(x), np.zeros_like(x))
    tgt = O.DiagNormalTarget(tm, ts)
    for t in range(3):
        _, eps = ctx.sample(x.astype(np.float32), 40 + t)
        ref = O.estimate_gradient(x.astype(np.float32).astype(np.float64), d, avi.FULLRANK, tgt, eps.cpu().numpy().astype(np.float64), ent)
        if rule == 0:
            x = O.descent_step(x, ref["grad"], eta)
        else:
            x, ost = O.adam_step(x, ref["grad"], ost, t + 1, eta)
        x = O.clip_scale(x, d, avi.FULLRANK, 1e-5)
    p3 = ctx.to_device(p0).clone()
    st3 = ctx.empty(2 * p3.numel()).zero_()
    ctx.optimize_steps(p3, st3 if rule == 1 else None, 40, 0, 3, rule, eta, 1e-5, ctx.empty(3))
    ctx.synchronize()
    got = p3.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got[low] - x[low])) <= tol * max(1.0, np.max(np.abs(x[low]))), np.max(np.abs(got[low] - x[low]))
    ctx.close()


@pytest.mark.parametrize("shape", [(64, 32), (2048, 64), (96, 200), (10, 5), (513, 256)], ids=["small", "c5", "four-waves", "tiny", "ragged"])
@pytest.mark.parametrize("rule", [0, 1], ids=["descent", "adam"])
@pytest.mark.parametrize("ent", [3, 0], ids=["STL", "CFE"])
def test_funnel_loop_matches_host_loop(rule, shape, ent):
    """The fused funnel target (BASELINE config 5: Neal's funnel + Stacked([log, identity]), mean-field): mivi_optimize_steps runs ONE
    kernel whose row-quad workgroups and row-0 workgroup exchange two scalars per workgroup and row 0's parameters every step
    (k_mf_funnel_sgd_loop).  It must reproduce, bitwise, the step-by-step sequence of single calls + update + ClipScale launches,
    including the ELBO record; and the first steps agree with the oracle's gradient + numpy update rules."""
    d, M = shape
    T = 9
    q0 = avi.MeanFieldGaussian((0.05 * np.arange(d) / d).astype(np.float32), np.full(d, 0.7, np.float32))
    p0, _ = avi.destructure(q0)
    eta = 5e-3 / max(1.0, d / 64.0)   # (row 0's gradient grows with d: a step that keeps Descent from diverging)
    ctx = avi.MiviContext(np.float32, avi.MEANFIELD, d, M, ent, SEED)
    ctx.set_problem(avi.FunnelProblem(d, 1.5))
    p = ctx.to_device(p0).clone()
    st = ctx.empty(2 * p.numel()).zero_()
    elbos = []
    for t in range(T):
        v, g = ctx.estimate_gradient(p, 40 + t)
        elbos.append(-float(v.item()))
        if rule == 0:
            ctx.descent_update(p, g, eta)
        else:
            ctx.adam_update(p, g, st, t + 1, eta)
        ctx.clip_scale(p, 1e-5)
    p2 = ctx.to_device(p0).clone()
    st2 = ctx.empty(2 * p2.numel()).zero_()
    elbo = ctx.empty(T)
    ctx.optimize_steps(p2, st2 if rule == 1 else None, 40, 0, T, rule, eta, 1e-5, elbo)
    ctx.synchronize()
    assert np.array_equal(p.cpu().numpy(), p2.cpu().numpy())
    if rule == 1:
        assert np.array_equal(st.cpu().numpy(), st2.cpu().numpy())
    assert np.allclose(elbo.cpu().numpy(), np.array(elbos, dtype=np.float32), rtol=1e-6)
    # a second call continues (t0 = T: Adam's bias correction, the estimate indices)
    for t in range(T, T + 3):
        v, g = ctx.estimate_gradient(p, 40 + t)
        if rule == 0:
            ctx.descent_update(p, g, eta)
        else:
            ctx.adam_update(p, g, st, t + 1, eta)
        ctx.clip_scale(p, 1e-5)
    ctx.optimize_steps(p2, st2 if rule == 1 else None, 40 + T, T, 3, rule, eta, 1e-5, ctx.empty(3))
    ctx.synchronize()
    assert np.array_equal(p.cpu().numpy(), p2.cpu().numpy())
    # independent restatement of the first steps
    x = p0.astype(np.float64)
    ost = (np.zeros_like(x), np.zeros_like(x))
    tgt = O.FunnelStackedTarget(d, 1.5)
    for t in range(2):
        _, eps = ctx.sample(x.astype(np.float32), 40 + t)
        ref = O.estimate_gradient(x.astype(np.float32).astype(np.float64), d, avi.MEANFIELD, tgt, eps.cpu().numpy().astype(np.float64), ent)
        if rule == 0:
            x = O.descent_step(x, ref["grad"], eta)
        else:
            x, ost = O.adam_step(x, ref["grad"], ost, t + 1, eta)
        x = O.clip_scale(x, d, avi.MEANFIELD, 1e-5)
    p3 = ctx.to_device(p0).clone()
    st3 = ctx.empty(2 * p3.numel()).zero_()
    ctx.optimize_steps(p3, st3 if rule == 1 else None, 40, 0, 2, rule, eta, 1e-5, ctx.empty(2))
    ctx.synchronize()
    got = p3.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - x)) <= 2e-5 * max(1.0, np.max(np.abs(x))), np.max(np.abs(got - x))
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
def test_update_and_projection_kernels_match_the_numpy_restatement(family, dtype):
    """mivi_descent_update / mivi_adam_update / mivi_clip_scale against oracle.descent_step / adam_step / clip_scale
    (Optimisers.jl Descent / Adam, src/optimization/clip_scale.jl:18-29) on a fixed gradient sequence: f32 to the ulp level
    (the restatement carries f32 arithmetic), f64 to 1e-13."""
    d = 37
    rng = np.random.default_rng(11)
    ctx = avi.MiviContext(dtype, family, d, 4, 0, SEED)
    plen = ctx.params_len
    x0 = rng.normal(size=plen).astype(dtype)
    if family == avi.FULLRANK:
        C = np.tril(x0[d:].reshape(d, d, order="F"))
        x0[d:] = C.reshape(-1, order="F")
    idx = d + np.arange(d) if family == avi.MEANFIELD else d + np.arange(d) * (d + 1)
    x0[idx] = rng.uniform(-0.5, 1.5, d).astype(dtype)        # some diagonals below the clip threshold (incl. negative)
    grads = [rng.normal(size=plen).astype(dtype) * dtype(10.0 ** rng.integers(-3, 2)) for _ in range(4)]
    if family == avi.FULLRANK:   # the estimator's gradient is exactly zero above the diagonal
        for g in grads:
            g[d:] = np.tril(g[d:].reshape(d, d, order="F")).reshape(-1, order="F")
    tol = 1e-13 if dtype == np.float64 else 4 * np.finfo(np.float32).eps
    # Descent
    x = x0.copy()
    p = ctx.to_device(x0).clone()
    for g in grads:
        ctx.descent_update(p, ctx.to_device(g), 0.0625)
        x = O.descent_step(x, g, 0.0625, dtype=dtype)
    assert np.max(np.abs(p.cpu().numpy().astype(np.float64) - x) / np.maximum(1.0, np.abs(x))) <= tol
    # Adam (bias correction in t), then ClipScale
    x, st = x0.copy(), (np.zeros(plen, dtype), np.zeros(plen, dtype))
    p = ctx.to_device(x0).clone()
    dst = ctx.empty(2 * plen).zero_()
    for t, g in enumerate(grads):
        ctx.adam_update(p, ctx.to_device(g), dst, t + 1, 1e-2)
        x, st = O.adam_step(x, g, st, t + 1, 1e-2, dtype=dtype)
    got = p.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - x) / np.maximum(1.0, np.abs(x))) <= 8 * tol
    assert np.max(np.abs(dst.cpu().numpy()[:plen].astype(np.float64) - st[0])) <= 8 * tol * max(1.0, np.max(np.abs(st[0])))
    ctx.clip_scale(p, 1e-5)
    ref = O.clip_scale(got, d, family, 1e-5)
    out = p.cpu().numpy().astype(np.float64)
    assert np.array_equal(out[idx], np.maximum(got[idx], dtype(1e-5)).astype(np.float64))
    assert np.max(np.abs(out - ref)) <= 1e-7        # (the oracle's threshold is the f64 1e-5, the kernel's the dtype's)
    ctx.close()


@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cocob_kernel_matches_the_numpy_restatement(family, dtype):
    """mivi_cocob_update against oracle.cocob_step (src/optimization/rules.jl:78-96) on a fixed gradient sequence; the entries
    above the diagonal of a full-rank scale never see a gradient and must stay put (the reference's expression is 0/0 there)."""
    d = 23
    rng = np.random.default_rng(12)
    ctx = avi.MiviContext(dtype, family, d, 4, 0, SEED)
    plen = ctx.params_len
    x0 = rng.normal(size=plen).astype(dtype)
    grads = [rng.normal(size=plen).astype(dtype) * dtype(10.0 ** rng.integers(-2, 2)) for _ in range(6)]
    if family == avi.FULLRANK:
        for g in grads:
            g[d:] = np.tril(g[d:].reshape(d, d, order="F")).reshape(-1, order="F")
    opt = avi.COCOB(100)
    p = ctx.to_device(x0).clone()
    st_dev = opt.setup(ctx, p)
    x, st = x0.copy(), O.cocob_init(x0)
    for t, g in enumerate(grads):
        opt.update(ctx, st_dev, p, ctx.to_device(g), t + 1)
        x, st = O.cocob_step(x, g, st, dtype(100))
    got = p.cpu().numpy()
    tol = 1e-12 if dtype == np.float64 else 64 * np.finfo(np.float32).eps
    assert np.all(np.isfinite(got))
    assert np.max(np.abs(got.astype(np.float64) - x) / np.maximum(1.0, np.abs(x))) <= tol
    sd = st_dev.cpu().numpy().astype(np.float64)
    for k in range(5):
        assert np.max(np.abs(sd[k * plen:(k + 1) * plen] - st[k]) / np.maximum(1.0, np.abs(st[k]))) <= tol
    if family == avi.FULLRANK:
        up = np.triu(np.ones((d, d), bool), 1).reshape(-1, order="F")
        assert np.array_equal(got[d:][up], x0[d:][up])
    ctx.close()


def test_cocob_through_optimize_reduces_the_objective():
    """KLMinRepGradDescent with COCOB (host-driven step loop: the rule is not in the device loop's table)."""
    d = 8
    tm, ts = np.full(d, 3.0, np.float32), np.full(d, 0.5, np.float32)
    q0 = avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32))
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=16, optimizer=avi.COCOB(), operator=avi.ClipScale())
    q, info, _ = avi.optimize(avi.PhiloxRNG(3), alg, 400, avi.DiagNormalProblem(tm, ts), q0)
    assert np.mean([i["elbo"] for i in info[-20:]]) > np.mean([i["elbo"] for i in info[:20]]) + 1.0
    assert np.linalg.norm(q.location - tm) < 0.5 * np.linalg.norm(tm)


def test_device_loop_converges_and_flags_divergence():
    d, M = 16, 16
    tm, ts = np.full(d, 5.0, np.float32), np.full(d, 0.3, np.float32)
    q0 = avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32))
    p0, _ = avi.destructure(q0)
    ctx = avi.MiviContext(np.float32, avi.MEANFIELD, d, M, 0, SEED)
    ctx.set_problem(avi.DiagNormalProblem(tm, ts))
    p = ctx.to_device(p0).clone()
    for blk in range(10):
        ctx.optimize_steps(p, None, 100 * blk, 100 * blk, 100, 0, 1e-3, 1e-5)
    out = p.cpu().numpy()
    d0 = np.sum((p0[:d] - tm) ** 2) + np.sum((p0[d:] - ts) ** 2)
    d1 = np.sum((out[:d] - tm) ** 2) + np.sum((out[d:] - ts) ** 2)
    assert d1 <= d0 / 2
    p = ctx.to_device(p0).clone()
    with pytest.raises(avi.MiviError) as e:
        ctx.optimize_steps(p, None, 0, 0, 5, 0, 1e30, 0.0)
    assert e.value.status in (2, 3)
    ctx.close()


def test_dog_dowg_and_averaging_match_oracle_formulas():
    """DoG / DoWG (src/optimization/rules.jl:17-64) and PolynomialAveraging (averaging.jl:36-53) kernels
    against a numpy restatement on a fixed gradient sequence."""
    d = 33
    rng = np.random.default_rng(6)
    ctx = avi.MiviContext(np.float64, avi.MEANFIELD, d, 4, 0, SEED)
    for kind in (0, 1):
        x0 = rng.normal(size=2 * d)
        grads = rng.normal(size=(5, 2 * d))
        x = ctx.to_device(x0).clone()
        st = ctx.dog_state()
        alpha = 1e-3
        ctx.dog_init(x, st, alpha)
        xr, v, r = x0.copy(), 0.0, alpha * (1 + np.linalg.norm(x0))
        for g in grads:
            ctx.dog_update(x, ctx.to_device(g), st, kind)
            r = max(np.linalg.norm(xr - x0), r)
            if kind == 1:
                v = v + r * r * np.sum(g * g)
                eta = r * r / np.sqrt(v)
            else:
                v = v + np.sum(g * g)
                eta = r / np.sqrt(v)
            xr = xr - eta * g
        assert np.allclose(x.cpu().numpy(), xr, rtol=1e-12, atol=1e-14)
    avg = avi.PolynomialAveraging(8)
    xs = rng.normal(size=(6, 2 * d))
    state = avg.init(ctx, ctx.to_device(xs[0]))
    ref, t = xs[0].copy(), 1
    for xrow in xs[1:]:
        state = avg.apply(ctx, state, ctx.to_device(xrow))
        w = 9.0 / (t + 8.0)
        ref = (1 - w) * ref + w * xrow
        t += 1
    assert np.allclose(avg.value(state).cpu().numpy(), ref, rtol=1e-12)
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK])
def test_proximal_operator_matches_oracle(family, dtype):
    """mivi_prox_scale_entropy vs the oracle for the three supported rules (Descent by value, DoG / DoWG from the
    device-resident optimiser state), incl. the reference's own known answer (d = 5, L = I, eta = 1e-2)."""
    d = 5
    q = (avi.MeanFieldGaussian(np.zeros(d, dtype), np.ones(d, dtype)) if family == avi.MEANFIELD
         else avi.FullRankGaussian(np.zeros(d, dtype), np.eye(d, dtype=dtype)))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(dtype, family, d, 4, 1, SEED)
    p = ctx.to_device(params).clone()
    ctx.prox_scale_entropy(p, 1e-2)
    ref = O.proximal_location_scale_entropy(params, d, family, 1e-2)
    tol = 1e-6 if dtype == np.float32 else 1e-14
    assert np.allclose(p.cpu().numpy(), ref, rtol=tol, atol=0)
    # DoG / DoWG: one optimiser step on the device, then prox with the step size implied by its (v, r)
    rng = np.random.default_rng(3)
    qq, _ = make_family(rng, 16, family, dtype)
    params, _ = avi.destructure(qq)
    grad = rng.normal(size=params.shape).astype(dtype)
    if family == avi.FULLRANK:   # gradients of a triangular scale have an exactly-zero upper triangle
        grad[16:] = np.tril(grad[16:].reshape(16, 16, order="F")).reshape(-1, order="F")
    c2 = avi.MiviContext(dtype, family, 16, 4, 1, SEED)
    for kind, name in ((0, "dog"), (1, "dowg")):
        p = c2.to_device(params).clone()
        st = c2.dog_state()
        c2.dog_init(p, st, 1e-2)
        c2.dog_update(p, c2.to_device(grad), st, kind)
        c2.prox_scale_entropy(p, 0.0, st, kind)
        x0 = params.astype(np.float64)
        p_ref, (_, v, r) = O.dog_step(x0, grad, (x0, 0.0, 1e-2 * (1.0 + np.linalg.norm(x0))), kind)
        ref = O.proximal_location_scale_entropy(p_ref, 16, family, O.stepsize_from_optimizer_state(name, v=v, r=r))
        assert np.allclose(p.cpu().numpy(), ref, rtol=(2e-6 if dtype == np.float32 else 1e-13), atol=1e-7 if dtype == np.float32 else 0)
    ctx.close(); c2.close()


def test_klminrepgradproxdescent_runs_deterministically_and_converges():
    """test/algorithms/klminrepgradproxdescent.jl: same-seed determinism (:40-57), estimate_objective at q = pi ~ 0
    (:36-37, atol 1e-3), and a convergence check in the spirit of :105-121 (distance to the optimum at least halves)."""
    d = 5
    rng = np.random.default_rng(1)
    mu_true = rng.normal(size=d)
    sig_true = rng.uniform(0.5, 1.5, size=d)
    prob = avi.DiagNormalProblem(mu_true, sig_true)
    q0 = avi.MeanFieldGaussian(np.zeros(d), np.ones(d))
    alg = avi.KLMinRepGradProxDescent(avi.AutoMIVI(), n_samples=10, optimizer=avi.DoG(1e-2))
    assert isinstance(alg.operator, avi.ProximalLocationScaleEntropy)
    assert isinstance(alg.objective.entropy, avi.ClosedFormEntropyZeroGradient)
    outs = []
    for _ in range(2):
        q, info, _ = avi.optimize(avi.PhiloxRNG(0x38bef07cf9cc549d), alg, 300, prob, q0)
        outs.append((q.location.copy(), np.asarray(q.scale).copy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    d0 = np.linalg.norm(mu_true) ** 2 + np.linalg.norm(1.0 - sig_true) ** 2
    d1 = np.linalg.norm(outs[0][0] - mu_true) ** 2 + np.linalg.norm(np.asarray(outs[0][1]).reshape(-1)[:d] - sig_true) ** 2 \
        if np.asarray(outs[0][1]).ndim == 1 else np.linalg.norm(outs[0][0] - mu_true) ** 2 + np.linalg.norm(np.diag(outs[0][1]) - sig_true) ** 2
    assert d1 <= d0 / 2, (d0, d1)
    q_true = avi.MeanFieldGaussian(mu_true, sig_true)
    assert abs(avi.estimate_objective(avi.PhiloxRNG(1), alg, q_true, prob, n_samples=10 ** 5)) < 1e-3
    with pytest.raises(TypeError):
        avi.KLMinRepGradProxDescent(avi.AutoMIVI(), optimizer=avi.Adam())
    with pytest.raises(TypeError):
        avi.KLMinRepGradProxDescent(avi.AutoMIVI(), entropy_zerograd=avi.ClosedFormEntropy())


def test_device_loop_with_the_logreg_target():
    """The built-in logistic regression is graph-capturable (its scratch is reserved before the capture), so the
    device-resident loop serves it too; it must reproduce the step-by-step sequence bitwise."""
    rng = np.random.default_rng(2)
    n, p, M, T = 500, 7, 16, 6
    d = p + 1
    X = (rng.normal(size=(n, p)) / np.sqrt(p)).astype(np.float32)
    y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    for family, route in ((avi.MEANFIELD, 1), (avi.MEANFIELD, 2), (avi.FULLRANK, 1), (avi.FULLRANK, 2)):
        q0 = (avi.MeanFieldGaussian(np.zeros(d, np.float32), np.full(d, 0.5, np.float32)) if family == avi.MEANFIELD
              else avi.FullRankGaussian(np.zeros(d, np.float32), 0.5 * np.eye(d, dtype=np.float32)))
        p0, _ = avi.destructure(q0)
        ctx = avi.MiviContext(np.float32, family, d, M, 0, SEED)
        ctx.set_problem(avi.LogRegProblem(X, y))
        ctx.set_logreg_route(route)      # matrix-core kernels / VALU kernels (the default picks by problem size)
        pa = ctx.to_device(p0).clone()
        st = ctx.empty(2 * pa.numel()).zero_()
        for t in range(T):
            v, g = ctx.estimate_gradient(pa, 50 + t)
            ctx.adam_update(pa, g, st, t + 1, 1e-2)
            ctx.clip_scale(pa, 1e-5)
        pb = ctx.to_device(p0).clone()
        st2 = ctx.empty(2 * pb.numel()).zero_()
        ctx.optimize_steps(pb, st2, 50, 0, T, 1, 1e-2, 1e-5, None)
        ctx.synchronize()
        assert np.array_equal(pa.cpu().numpy(), pb.cpu().numpy())
        ctx.close()


def test_device_loop_with_the_logreg_target_on_operand_planes():
    """The same with a data set that has operand planes (n p >= 1e5, n_mc = 128: k_lr_zplanes / k_lr_logits_planes / k_lr_xtr_planes inside
    the captured graph, their scratch reserved before the capture): bitwise the step-by-step sequence."""
    rng = np.random.default_rng(21)
    n, p, M, T = 2200, 63, 128, 7
    d = p + 1
    X = (rng.normal(size=(n, p)) / np.sqrt(p)).astype(np.float32)
    y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    for family in (avi.MEANFIELD, avi.FULLRANK):
        q0 = (avi.MeanFieldGaussian(np.zeros(d, np.float32), np.full(d, 0.5, np.float32)) if family == avi.MEANFIELD
              else avi.FullRankGaussian(np.zeros(d, np.float32), 0.5 * np.eye(d, dtype=np.float32)))
        p0, _ = avi.destructure(q0)
        ctx = avi.MiviContext(np.float32, family, d, M, 0, SEED)
        ctx.set_problem(avi.LogRegProblem(X, y))
        ctx.set_logreg_route(1)
        assert ctx.logreg_kernels()["xtr_planes"]
        pb = ctx.to_device(p0).clone()
        st2 = ctx.empty(2 * pb.numel()).zero_()
        ctx.optimize_steps(pb, st2, 50, 0, T, 1, 1e-2, 1e-5, None)      # (the loop FIRST: nothing of the route has run on this context yet)
        ctx.synchronize()
        pa = ctx.to_device(p0).clone()
        st = ctx.empty(2 * pa.numel()).zero_()
        for t in range(T):
            v, g = ctx.estimate_gradient(pa, 50 + t)
            ctx.adam_update(pa, g, st, t + 1, 1e-2)
            ctx.clip_scale(pa, 1e-5)
        ctx.synchronize()
        assert np.array_equal(pa.cpu().numpy(), pb.cpu().numpy())
        ctx.close()


@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
@pytest.mark.parametrize("combo", [
    ("descent", "clip", "poly"), ("adam", "identity", "none"), ("adam", "clip", "poly"), ("dog", "clip", "poly"),
    ("dowg", "identity", "poly"), ("dowg", "prox", "poly"), ("descent", "prox", "none"), ("dog", "prox", "none")])
def test_optimize_device_loop_equals_host_loop(family, combo):
    """`optimize` runs every iteration inside mivi_optimize_loop when there is no callback; the result (parameters,
    averaged output, per-iteration elbo) must be bitwise what the host-driven `step` loop produces, for every
    rule x operator x averager the reference's algorithms combine -- including a warm start (optimize.jl:58-62)."""
    rule, op, avg = combo
    d, T = 12, 37
    rng = np.random.default_rng(7)
    mu, sig = rng.normal(size=d), rng.uniform(0.5, 1.5, size=d)
    prob = avi.DiagNormalProblem(mu, sig) if family == avi.MEANFIELD else avi.DenseNormalProblem(mu, np.tril(rng.normal(size=(d, d)) * 0.1) + np.diag(sig))
    q0 = avi.MeanFieldGaussian(np.zeros(d), np.ones(d)) if family == avi.MEANFIELD else avi.FullRankGaussian(np.zeros(d), np.eye(d))
    opt = {"descent": avi.Descent(1e-2), "adam": avi.Adam(5e-2), "dog": avi.DoG(1e-2), "dowg": avi.DoWG(1e-2)}[rule]
    averager = avi.PolynomialAveraging() if avg == "poly" else avi.NoAveraging()
    if op == "prox":
        alg = avi.KLMinRepGradProxDescent(avi.AutoMIVI(), n_samples=8, optimizer=opt, averager=averager)
    else:
        alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=8, optimizer=opt, averager=averager,
                                      operator=avi.ClipScale() if op == "clip" else avi.IdentityOperator())
    outs = []
    import warnings
    for dev in (True, False):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q1, info1, st = avi.optimize(avi.PhiloxRNG(5), alg, T, prob, q0, device_loop=dev)
            rng2 = avi.PhiloxRNG(5, T)      # the estimate stream continues where the first call stopped
            q2, info2, st2 = avi.optimize(rng2, alg, 11, prob, None, state=st, device_loop=dev)
        outs.append((q2.location.copy(), np.asarray(q2.scale).copy(), st2["params"].cpu().numpy().copy(),
                     np.array([i["elbo"] for i in info1 + info2]), [i["iteration"] for i in info1]))
    a, b = outs
    if family == avi.MEANFIELD and rule in ("dog", "dowg"):
        # launch-free here too (k_mf_gen_loop): DoG / DoWG's two norms are summed over the workgroups' partials, not in k_dog_update's order --
        # equal to the rounding of those f64 sums
        assert np.allclose(a[2], b[2], rtol=1e-11, atol=1e-13) and np.allclose(a[0], b[0], rtol=1e-11, atol=1e-13) and np.allclose(a[1], b[1], rtol=1e-11, atol=1e-13)
    else:
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.allclose(a[3], b[3], rtol=1e-10 if rule in ("dog", "dowg") else 1e-12, atol=0) and a[4] == b[4] == list(range(1, T + 1))


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape", [(1024, 8), (2048, 256), (70, 19), (4096, 4)], ids=["d1024-m8", "d2048-m256", "ragged", "d4096-m4"])
@pytest.mark.parametrize("combo", [
    ("dowg", "clip", "poly"), ("dowg", "prox", "poly"), ("dog", "clip", "none"), ("descent", "prox", "poly"), ("adam", "clip", "poly")])
def test_meanfield_general_loop(combo, shape, dtype):
    """Mean-field family, diagonal-Gaussian target, the rules x operators x averagers beyond Descent / Adam + ClipScale -- DoWG +
    PolynomialAveraging + ClipScale / ProximalLocationScaleEntropy are the reference's DEFAULTS (src/algorithms/constructors.jl:44-157): the
    device loop is ONE launch-free kernel here too (k_mf_gen_loop; DoG / DoWG exchange two norm partials per workgroup and step).  Against the
    host-driven `step` loop (separate estimate / update / operator / averager launches): bit for bit for Descent / Adam, to the rounding of
    the two f64 norm sums for DoG / DoWG -- parameters, averaged output, elbo record; a warm start continues."""
    rule, op, avg = combo
    d, M = shape
    T = 23
    rng = np.random.default_rng(3)
    mu, sig = rng.normal(size=d).astype(dtype), rng.uniform(0.5, 1.5, size=d).astype(dtype)
    prob = avi.DiagNormalProblem(mu, sig)
    q0 = avi.MeanFieldGaussian(np.zeros(d, dtype), np.ones(d, dtype))
    opt = {"descent": avi.Descent(1e-2), "adam": avi.Adam(5e-2), "dog": avi.DoG(1e-2), "dowg": avi.DoWG(1e-2)}[rule]
    averager = avi.PolynomialAveraging() if avg == "poly" else avi.NoAveraging()
    if op == "prox":
        alg = avi.KLMinRepGradProxDescent(avi.AutoMIVI(), n_samples=M, optimizer=opt, averager=averager)
    else:
        alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=M, optimizer=opt, averager=averager, operator=avi.ClipScale())
    outs = []
    import warnings
    for dev in (True, False):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q1, info1, st = avi.optimize(avi.PhiloxRNG(9), alg, T, prob, q0, device_loop=dev)
            q2, info2, st2 = avi.optimize(avi.PhiloxRNG(9, T), alg, 9, prob, None, state=st, device_loop=dev)
        outs.append((q2.location.copy(), np.asarray(q2.scale).copy(), st2["params"].cpu().numpy().copy(), np.array([i["elbo"] for i in info1 + info2])))
    a, b = outs
    if rule in ("dog", "dowg"):
        tol = 2e-5 if dtype == np.float32 else 1e-11
        for x, y in zip(a[:3], b[:3]):
            assert np.max(np.abs(x.astype(np.float64) - y.astype(np.float64))) <= tol * max(1.0, np.max(np.abs(y))), np.max(np.abs(x - y))
    else:
        for x, y in zip(a[:3], b[:3]):
            assert np.array_equal(x, y)
    assert np.allclose(a[3], b[3], rtol=1e-5 if dtype == np.float32 else 1e-10)


@pytest.mark.parametrize("shape", [(1024, 1), (256, 8), (130, 2)], ids=["d1024-one-sample", "d256-m8", "ragged"])
@pytest.mark.parametrize("combo", [
    ("dowg", "clip", "poly"), ("dowg", "prox", "poly"), ("dog", "clip", "none"), ("descent", "prox", "poly"), ("adam", "clip", "poly")])
def test_fullrank_rows_general_loop(combo, shape):
    """Full-rank family with few samples per step (the reference's default is n_samples = 1), diagonal-Gaussian target, the reference's DEFAULT
    rule / averager (DoWG + PolynomialAveraging) and the other combinations beyond Descent / Adam + ClipScale: the row-owning workgroups of
    k_fr_rows_loop (DoG / DoWG: two norm partials per workgroup and step).  Against the host-driven `step` loop, to rounding (the loop's sums
    are sequential multiply-adds, not the tile kernels' MFMA chains): parameters, averaged output, elbo record; a warm start continues."""
    rule, op, avg = combo
    d, M = shape
    T = 14
    rng = np.random.default_rng(5)
    mu, sig = rng.normal(size=d).astype(np.float32), rng.uniform(0.5, 1.5, size=d).astype(np.float32)
    prob = avi.DiagNormalProblem(mu, sig)
    C0 = (np.eye(d) + (0.3 / np.sqrt(d)) * np.tril(rng.normal(size=(d, d)), -1)).astype(np.float32)
    q0 = avi.FullRankGaussian(np.zeros(d, np.float32), C0)
    opt = {"descent": avi.Descent(1e-2), "adam": avi.Adam(1e-2), "dog": avi.DoG(1e-2), "dowg": avi.DoWG(1e-2)}[rule]
    averager = avi.PolynomialAveraging() if avg == "poly" else avi.NoAveraging()
    if op == "prox":
        alg = avi.KLMinRepGradProxDescent(avi.AutoMIVI(), n_samples=M, optimizer=opt, averager=averager)
    else:
        alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=M, optimizer=opt, averager=averager, operator=avi.ClipScale())
    outs = []
    import warnings
    for dev in (True, False):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q1, info1, st = avi.optimize(avi.PhiloxRNG(11), alg, T, prob, q0, device_loop=dev)
            q2, info2, st2 = avi.optimize(avi.PhiloxRNG(11, T), alg, 6, prob, None, state=st, device_loop=dev)
        outs.append((q2.location.copy(), np.asarray(q2.scale).copy(), st2["params"].cpu().numpy().copy(), np.array([i["elbo"] for i in info1 + info2])))
    a, b = outs
    tol = 1e-4 if rule == "adam" else 3e-5
    for x, y in zip(a[:3], b[:3]):
        assert np.max(np.abs(x.astype(np.float64) - y.astype(np.float64))) <= tol * max(1.0, np.max(np.abs(y))), np.max(np.abs(x - y))
    assert np.allclose(a[3], b[3], rtol=5e-5, atol=1e-3)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape", [(10, 1), (32, 16), (17, 3)], ids=["reference-bench", "d32-m16", "ragged"])
@pytest.mark.parametrize("combo", [
    ("dowg", "clip", "poly"), ("dowg", "prox", "poly"), ("dog", "clip", "none"), ("descent", "prox", "poly"), ("adam", "clip", "poly")])
def test_small_fullrank_general_loop(combo, shape, dtype):
    """Small full-rank problems (one workgroup, k_fr_small_loop) with the reference's DEFAULT rule / averager (DoWG + PolynomialAveraging) and the
    other combinations beyond Descent / Adam + ClipScale: against the host-driven `step` loop, to rounding -- parameters, averaged output,
    elbo record; a warm start continues."""
    rule, op, avg = combo
    d, M = shape
    T = 14
    rng = np.random.default_rng(6)
    mu, sig = rng.normal(size=d).astype(dtype), rng.uniform(0.5, 1.5, size=d).astype(dtype)
    prob = avi.DiagNormalProblem(mu, sig)
    C0 = (np.eye(d) + 0.1 * np.tril(rng.normal(size=(d, d)), -1)).astype(dtype)
    q0 = avi.FullRankGaussian(np.zeros(d, dtype), C0)
    opt = {"descent": avi.Descent(1e-2), "adam": avi.Adam(1e-2), "dog": avi.DoG(1e-2), "dowg": avi.DoWG(1e-2)}[rule]
    averager = avi.PolynomialAveraging() if avg == "poly" else avi.NoAveraging()
    if op == "prox":
        alg = avi.KLMinRepGradProxDescent(avi.AutoMIVI(), n_samples=M, optimizer=opt, averager=averager)
    else:
        alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=M, optimizer=opt, averager=averager, operator=avi.ClipScale())
    outs = []
    import warnings
    for dev in (True, False):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q1, info1, st = avi.optimize(avi.PhiloxRNG(12), alg, T, prob, q0, device_loop=dev)
            q2, info2, st2 = avi.optimize(avi.PhiloxRNG(12, T), alg, 6, prob, None, state=st, device_loop=dev)
        outs.append((q2.location.copy(), np.asarray(q2.scale).copy(), st2["params"].cpu().numpy().copy(), np.array([i["elbo"] for i in info1 + info2])))
    a, b = outs
    tol = (1e-4 if rule == "adam" else 3e-5) if dtype == np.float32 else 1e-10
    for x, y in zip(a[:3], b[:3]):
        assert np.max(np.abs(x.astype(np.float64) - y.astype(np.float64))) <= tol * max(1.0, np.max(np.abs(y))), np.max(np.abs(x - y))
    assert np.allclose(a[3], b[3], rtol=5e-5 if dtype == np.float32 else 1e-9, atol=1e-4 if dtype == np.float32 else 1e-9)


@pytest.mark.parametrize("family,d,M", [(avi.MEANFIELD, 1024, 8), (avi.FULLRANK, 256, 4)], ids=["meanfield", "fullrank-rows"])
def test_dog_loops_report_divergence_not_a_lost_exchange(family, d, M):
    """In the launch-free DoG / DoWG loops a partial norm's slot holds NaN until its workgroup has stored it.  Diverged parameters make the
    norms themselves NaN: they must travel (as +Inf) and the call must end with the reference's `diverged` status (non-finite objective /
    non-positive scale, src/algorithms/common.jl:83-89), not with an expired device-side wait."""
    q0 = (avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32)) if family == avi.MEANFIELD
          else avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32)))
    p0, _ = avi.destructure(q0)
    ctx = avi.MiviContext(np.float32, family, d, M, 0, SEED)
    ctx.set_problem(avi.DiagNormalProblem(np.zeros(d, np.float32), np.ones(d, np.float32)))
    p = ctx.to_device(p0).clone()
    st = ctx.dog_state()
    ctx.dog_init(p, st, 1e-6)
    p[3] = float("nan")
    with pytest.raises(avi.MiviError) as ei:
        ctx.optimize_loop(p, 5, 0, 0, rule=3, op=1, averager=0, clip_epsilon=1e-5, opt_state=st)
        ctx.synchronize()
    assert ei.value.status in (2, 3) and "wait expired" not in str(ei.value)
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("case", [("fullrank", 208, 60, 1, "lognormal_exp_bijector"), ("meanfield", 208, 60, 1, "lognormal_exp_bijector"),
                                  ("fullrank", 100, 20, 8, "logsigma_normal"), ("meanfield", 333, 7, 3, "logsigma_normal"),
                                  ("meanfield", 3001, 16, 2, "lognormal_exp_bijector"), ("fullrank", 1500, 12, 4, "logsigma_normal")],
                         ids=["readme-sonar-fullrank-one-sample", "readme-sonar-meanfield-one-sample", "fullrank-m8", "ragged-meanfield",
                              "six-workgroups-meanfield", "five-workgroups-fullrank"])
@pytest.mark.parametrize("combo", [("dowg", "prox", "poly"), ("adam", "clip", "none"), ("descent", "clip", "poly"), ("dog", "clip", "none")])
def test_logreg_small_loop(combo, case, dtype):
    """Tiny hierarchical logistic regressions -- the reference README's own example (README.md:42-119: 208 rows, 60 features, theta = [beta; sigma]
    behind the exp bijector, one sample per step) and neighbours -- run the whole `optimize` loop inside ONE kernel (k_lr_small_loop: one
    workgroup, or several that split the rows of X and exchange their partial sums every step), for every rule x operator x averager.  Against the host-driven `step` loop (separate launches of the general
    route), to rounding: parameters, averaged output, elbo record; a warm start continues."""
    rule, op, avg = combo
    fam, n, p, M, variant = case
    d = p + 1
    T = 12
    rng = np.random.default_rng(21)
    X = rng.normal(size=(n, p)).astype(dtype)
    beta_true = rng.normal(size=p) * 0.5
    y = (rng.uniform(size=n) < 1.0 / (1.0 + np.exp(-X.astype(np.float64) @ beta_true))).astype(np.float32)
    prob = avi.LogRegProblem(X, y, variant=variant)
    q0 = (avi.MeanFieldGaussian(np.zeros(d, dtype), np.full(d, 0.5, dtype)) if fam == "meanfield"
          else avi.FullRankGaussian(np.zeros(d, dtype), (0.5 * np.eye(d)).astype(dtype)))
    opt = {"descent": avi.Descent(1e-4), "adam": avi.Adam(1e-2), "dog": avi.DoG(1e-2), "dowg": avi.DoWG(1e-2)}[rule]
    averager = avi.PolynomialAveraging() if avg == "poly" else avi.NoAveraging()
    if op == "prox":
        alg = avi.KLMinRepGradProxDescent(avi.AutoMIVI(), n_samples=M, optimizer=opt, averager=averager)
    else:
        alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=M, optimizer=opt, averager=averager, operator=avi.ClipScale())
    outs = []
    import warnings
    for dev in (True, False):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q1, info1, st = avi.optimize(avi.PhiloxRNG(13), alg, T, prob, q0, device_loop=dev)
            q2, info2, st2 = avi.optimize(avi.PhiloxRNG(13, T), alg, 5, prob, None, state=st, device_loop=dev)
        outs.append((q2.location.copy(), np.asarray(q2.scale).copy(), st2["params"].cpu().numpy().copy(), np.array([i["elbo"] for i in info1 + info2])))
    a, b = outs
    tol = 2e-4 if dtype == np.float32 else 1e-9
    for x, yv in zip(a[:3], b[:3]):
        assert np.max(np.abs(x.astype(np.float64) - yv.astype(np.float64))) <= tol * max(1.0, np.max(np.abs(yv))), np.max(np.abs(x - yv))
    assert np.allclose(a[3], b[3], rtol=2e-4 if dtype == np.float32 else 1e-9)


@pytest.mark.parametrize("family,d,M", [(avi.MEANFIELD, 64, 8), (avi.FULLRANK, 128, 4), (avi.FULLRANK, 10, 1)], ids=["meanfield", "fullrank-rows", "fullrank-small"])
@pytest.mark.parametrize("rule", [0, 3], ids=["descent", "dowg"])
def test_launch_free_loops_edge_calls(family, d, M, rule):
    """Edge calls of the launch-free loops through mivi_optimize_loop: a single step, no ELBO record requested, IdentityOperator (no ClipScale),
    a second call that continues the first -- the same parameters as one call of two steps (bitwise: the same kernel on the same stream of draws)."""
    q0 = (avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32)) if family == avi.MEANFIELD
          else avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32)))
    p0, _ = avi.destructure(q0)
    ctx = avi.MiviContext(np.float32, family, d, M, 0, SEED)
    ctx.set_problem(avi.DiagNormalProblem(np.full(d, 0.5, np.float32), np.ones(d, np.float32)))

    def run(splits):
        p = ctx.to_device(p0).clone()
        st = None
        if rule == 3:
            st = ctx.dog_state()
            ctx.dog_init(p, st, 1e-6)
        done = 0
        for n in splits:
            ctx.optimize_loop(p, n, 7 + done, done, rule=rule, op=0, averager=0, eta=1e-3, opt_state=st, elbo=None)
            done += n
        ctx.synchronize()
        return p.cpu().numpy()

    a, b = run([1, 1]), run([2])
    assert np.all(np.isfinite(a)) and np.array_equal(a, b)
    assert not np.array_equal(a, p0)
    ctx.close()


def test_optimize_falls_back_to_the_host_loop_for_plugin_targets_and_callbacks():
    class Plug:
        def __init__(self, mu):
            self.mu = mu

        def dimension(self):
            return self.mu.size

        def capabilities(self):
            return avi.LogDensityOrder(1)

        def logdensity_and_gradient(self, x):
            r = x - self.mu
            return -0.5 * float(r @ r), -r

    d = 4
    q0 = avi.MeanFieldGaussian(np.zeros(d), np.ones(d))
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=8, optimizer=avi.Adam(5e-2), operator=avi.ClipScale())
    q, info, _ = avi.optimize(avi.PhiloxRNG(1), alg, 60, Plug(np.arange(d, dtype=float)), q0)     # host callback target
    assert len(info) == 60 and np.linalg.norm(q.location - np.arange(d)) < 0.8
    seen = []
    avi.optimize(avi.PhiloxRNG(1), alg, 5, avi.DiagNormalProblem(np.zeros(d), np.ones(d)), q0,
                 callback=lambda **kw: seen.append(kw["iteration"]))
    assert seen == [1, 2, 3, 4, 5]


def test_cached_graph_survives_a_capacity_change():
    """A captured loop bakes work-buffer pointers and leading dimensions; an objective estimate with more samples than n_mc
    reallocates them.  The next loop call must not replay into freed memory (round-1 advisor finding): same result as a
    context that never grew."""
    d, M, T = 128, 128, 6
    rng = np.random.default_rng(21)
    tm, ts = rng.normal(size=d).astype(np.float32), rng.uniform(0.5, 2, size=d).astype(np.float32)
    q0 = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
    p0, _ = avi.destructure(q0)
    outs = []
    for grow in (False, True):
        ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
        ctx.set_problem(avi.DiagNormalProblem(tm, ts))
        p = ctx.to_device(p0).clone()
        st = ctx.empty(2 * p.numel()).zero_()
        ctx.optimize_steps(p, st, 0, 0, T, 1, 1e-2, 1e-5)
        if grow:
            ctx.estimate_objective(p, 99, n_samples=4 * M)       # M grows: work buffers are reallocated
        ctx.optimize_steps(p, st, T, T, T, 1, 1e-2, 1e-5)
        ctx.synchronize()
        outs.append(p.cpu().numpy().copy())
        ctx.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
def test_clipscale_with_nonpositive_epsilon_is_the_same_on_both_loops(family):
    """ClipScale(epsilon <= 0) still clamps negative diagonals to epsilon (clip_scale.jl:18-29): the device-resident loop must
    not read `epsilon <= 0` as `no operator` (round-1 advisor finding)."""
    d, M = 64, 32
    rng = np.random.default_rng(22)
    tm, ts = rng.normal(size=d).astype(np.float32), rng.uniform(0.5, 2, size=d).astype(np.float32)
    scale = -0.5 * np.ones(d, np.float32)            # negative diagonals: only a ClipScale can repair them
    q0 = avi.MeanFieldGaussian(np.zeros(d, np.float32), scale) if family == avi.MEANFIELD else \
        avi.FullRankGaussian(np.zeros(d, np.float32), np.diag(scale).astype(np.float32))
    res = []
    for dev in (True, False):
        alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), optimizer=avi.Descent(1e-3), n_samples=M, operator=avi.ClipScale(0.0),
                                      averager=avi.NoAveraging())
        try:
            out, info, st = avi.optimize(avi.PhiloxRNG(SEED), alg, 3, avi.DiagNormalProblem(tm, ts), q0, device_loop=dev)
            res.append(("ok", st["params"].cpu().numpy().copy()))
        except RuntimeError as e:
            res.append(("diverged", str(e)))
    assert res[0][0] == res[1][0]
    if res[0][0] == "ok":
        assert np.array_equal(res[0][1], res[1][1])


def test_divergence_inside_a_device_chunk_leaves_the_host_loop_state():
    """A run that diverges at step k: the device loop must raise with the steps before k applied and rng / iteration
    advanced exactly like the host-driven loop (the reference throws at the offending step, common.jl:83-89)."""
    d, M = 16, 16
    tm, ts = np.full(d, 5.0, np.float32), np.full(d, 0.3, np.float32)
    q0 = avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32))
    states = []
    for dev in (True, False):
        alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), optimizer=avi.Descent(3.0e1), n_samples=M, operator=avi.IdentityOperator(),
                                      averager=avi.NoAveraging())
        rng = avi.PhiloxRNG(SEED)
        st = avi.init(rng, alg, q0, avi.DiagNormalProblem(tm, ts))
        with pytest.raises(RuntimeError):
            for _ in range(4):                       # chunks of the public API; the failure is somewhere inside
                _, _, st = avi.optimize(rng, alg, 50, state=st, device_loop=dev)
        states.append((st["iteration"], rng.counter, st["params"].cpu().numpy().copy()))
    # the failing call raises before returning its state: compare what the caller still holds + the rng position
    assert states[0][0] == states[1][0]
    assert states[0][1] == states[1][1]
    assert np.array_equal(states[0][2], states[1][2], equal_nan=True)
