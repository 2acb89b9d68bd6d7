"""The launch-free optimisation loops for every rule x operator x averager beyond Descent / Adam + ClipScale (k_mf_gen_loop, k_fr_rows_loop,
k_fr_small_loop, k_lr_small_loop) against an INDEPENDENT f64 trajectory: oracle gradient on the device's own eps -> numpy rule
(oracle.dog_step / descent_step / adam_step: src/optimization/rules.jl:17-64 and Optimisers.jl) -> operator (oracle.clip_scale /
proximal_location_scale_entropy: clip_scale.jl:18-29, proximal_location_scale_entropy.jl:26-61) -> PolynomialAveraging
(averaging.jl:36-53), in the order of `step` (src/algorithms/common.jl:69-104).  Round 4 compared these loops with the HIP host loop only.
Also here: the loops' robustness contract -- a lost grid-wide exchange restores the state and re-runs on the graph of launches; a scale
diagonal below zero comes back positive through the proximal operator; `info.iteration` on a warm start."""
import os
import subprocess
import sys

import numpy as np
import pytest

import advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED, make_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RULES = {"descent": 0, "adam": 1, "dog": 2, "dowg": 3, "cocob": 4}
OPS = {"none": 0, "clip": 1, "prox": 2}
COMBOS = [("dowg", "clip", "poly"), ("dowg", "prox", "poly"), ("dog", "clip", "none"), ("dog", "prox", "poly"), ("descent", "prox", "poly"),
          ("adam", "clip", "poly")]


def oracle_trajectory(ctx, p0, d, family, tgt, ent, rule, op, avg, T, idx0, eta, alpha, clip_eps, avg_eta=8.0):
    """T iterations of `step` in f64 on the device's eps stream; returns (params, running average or None, elbo list)."""
    x = p0.astype(np.float64)
    dstate = (x.copy(), 0.0, alpha * (1.0 + float(np.linalg.norm(x))))      # DoG / DoWG init (rules.jl:22-24, 53-55)
    ast = (np.zeros_like(x), np.zeros_like(x))
    cst = O.cocob_init(x)
    xbar, elbos = None, []
    for t in range(T):
        _, eps = ctx.sample(x.astype(p0.dtype), idx0 + t)
        ref = O.estimate_gradient(x.astype(p0.dtype).astype(np.float64), d, family, tgt, eps.cpu().numpy().astype(np.float64), ent)
        elbos.append(-ref["value"])
        g = ref["grad"]
        if rule == "descent":
            x = O.descent_step(x, g, eta)
        elif rule == "adam":
            x, ast = O.adam_step(x, g, ast, t + 1, eta)
        elif rule == "cocob":
            x, cst = O.cocob_step(x, g, cst, eta)          # (alpha travels in `eta`: include/mivi.h)
        else:
            x, dstate = O.dog_step(x, g, dstate, 1 if rule == "dowg" else 0)
        if op == "clip":
            x = O.clip_scale(x, d, family, clip_eps)
        elif op == "prox":
            gamma = O.stepsize_from_optimizer_state(rule, eta=eta, v=dstate[1], r=dstate[2])
            x = O.proximal_location_scale_entropy(x, d, family, gamma)
        if avg == "poly":
            w = (avg_eta + 1.0) / ((t + 1) + avg_eta)
            xbar = x.copy() if xbar is None else (1.0 - w) * xbar + w * x
    return x, xbar, np.array(elbos)


def device_loop(ctx, p0, rule, op, avg, T, idx0, eta, alpha, clip_eps, avg_eta=8.0):
    p = ctx.to_device(p0).clone()
    st = None
    if rule == "adam":
        st = ctx.empty(2 * p.numel()).zero_()
    elif rule in ("dog", "dowg"):
        st = ctx.dog_state()
        ctx.dog_init(p, st, alpha)
    elif rule == "cocob":                                   # (L, G, R, theta, x1) = (0, 0, 0, 0, params): rules.jl:84-86
        st = ctx.empty(5 * p.numel()).zero_()
        st[4 * p.numel():] = p
    ap = p.clone() if avg == "poly" else None
    elbo = ctx.empty(T)
    ctx.optimize_loop(p, T, idx0, 0, rule=RULES[rule], op=OPS[op], averager=1 if avg == "poly" else 0, eta=eta, clip_epsilon=clip_eps,
                      avg_eta=avg_eta, opt_state=st, avg_params=ap, elbo=elbo)
    ctx.synchronize()
    return p.cpu().numpy().astype(np.float64), (ap.cpu().numpy().astype(np.float64) if ap is not None else None), elbo.cpu().numpy().astype(np.float64)


def _check(ctx, p0, d, family, tgt, ent, combo, T=3, tol=5e-6, eta=1e-2, alpha=1e-2):
    rule, op, avg = combo
    if op == "prox" and rule == "adam":
        pytest.skip("ProximalLocationScaleEntropy has no step size for Adam (proximal_location_scale_entropy.jl:26-42)")
    x, xbar, el = oracle_trajectory(ctx, p0, d, family, tgt, ent, rule, op, avg, T, 70, eta, alpha, 1e-5)
    got, gbar, gel = device_loop(ctx, p0, rule, op, avg, T, 70, eta, alpha, 1e-5)
    low = np.ones(p0.size, bool) if family == avi.MEANFIELD else np.concatenate([np.ones(d, bool), np.tril(np.ones((d, d), bool)).T.reshape(-1)])
    assert np.max(np.abs(got[low] - x[low])) <= tol * max(1.0, np.max(np.abs(x))), (combo, np.max(np.abs(got[low] - x[low])))
    if xbar is not None:
        assert np.max(np.abs(gbar[low] - xbar[low])) <= tol * max(1.0, np.max(np.abs(xbar))), combo
    assert np.allclose(gel, el, rtol=2e-5, atol=1e-4), (combo, gel, el)


@pytest.mark.parametrize("combo", COMBOS, ids=["-".join(c) for c in COMBOS])
@pytest.mark.parametrize("shape", [(1024, 8), (70, 19)], ids=["d1024-m8", "ragged"])
def test_meanfield_general_loop_follows_the_oracle(combo, shape):
    d, M = shape
    rng = np.random.default_rng(11)
    prob, tgt = make_problem(rng, "diag", d, np.float32)
    q0 = avi.MeanFieldGaussian((0.2 * rng.normal(size=d)).astype(np.float32), rng.uniform(0.6, 1.4, size=d).astype(np.float32))
    p0, _ = avi.destructure(q0)
    ctx = avi.MiviContext(np.float32, avi.MEANFIELD, d, M, 0, SEED)
    ctx.set_problem(prob)
    _check(ctx, p0, d, avi.MEANFIELD, tgt, 0, combo)
    ctx.close()


@pytest.mark.parametrize("combo", COMBOS, ids=["-".join(c) for c in COMBOS])
@pytest.mark.parametrize("shape", [(256, 8), (1024, 1), (10, 4)], ids=["rows-d256-m8", "rows-d1024-m1", "small-d10-m4"])
def test_fullrank_general_loops_follow_the_oracle(combo, shape):
    """(256, 8), (1024, 1): the row-owning workgroups of k_fr_rows_loop; (10, 4): one workgroup, k_fr_small_loop."""
    d, M = shape
    rng = np.random.default_rng(12)
    prob, tgt = make_problem(rng, "diag", d, np.float32)
    C0 = (np.eye(d) + (0.3 / np.sqrt(d)) * np.tril(rng.normal(size=(d, d)), -1)).astype(np.float32)
    q0 = avi.FullRankGaussian((0.2 * rng.normal(size=d)).astype(np.float32), C0)
    p0, _ = avi.destructure(q0)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
    ctx.set_problem(prob)
    _check(ctx, p0, d, avi.FULLRANK, tgt, 0, combo)
    ctx.close()


@pytest.mark.parametrize("combo", [("cocob", "clip", "poly"), ("cocob", "none", "none")], ids=["cocob-clip-poly", "cocob-none-none"])
@pytest.mark.parametrize("cfg", [(avi.MEANFIELD, 1024, 8, np.float32), (avi.MEANFIELD, 70, 19, np.float64), (avi.FULLRANK, 256, 8, np.float32),
                                 (avi.FULLRANK, 10, 4, np.float64), (avi.FULLRANK, 128, 128, np.float32)],
                         ids=["mf-d1024-m8-f32", "mf-ragged-f64", "fr-d256-m8-f32", "fr-d10-m4-f64", "fr-d128-m128-f32"])
def test_cocob_in_the_device_loop_follows_the_oracle(combo, cfg):
    """Round 6: COCOB (src/optimization/rules.jl:66-96) as rule 4 of mivi_optimize_loop -- a hipGraph of chained estimates with the update
    kernel, no host round trip per step -- against oracle.cocob_step (pinned on the reference's own rule test, tests/test_oracle_pinning.py)
    for three steps, with ClipScale + PolynomialAveraging and bare.  alpha = 100 (the reference's default); the proximal operator is refused."""
    family, d, M, dtype = cfg
    rng = np.random.default_rng(14)
    prob, tgt = make_problem(rng, "diag", d, dtype)
    if family == avi.MEANFIELD:
        q0 = avi.MeanFieldGaussian((0.2 * rng.normal(size=d)).astype(dtype), rng.uniform(0.6, 1.4, size=d).astype(dtype))
    else:
        q0 = avi.FullRankGaussian((0.2 * rng.normal(size=d)).astype(dtype), (np.eye(d) + (0.3 / np.sqrt(d)) * np.tril(rng.normal(size=(d, d)), -1)).astype(dtype))
    p0, _ = avi.destructure(q0)
    ctx = avi.MiviContext(dtype, family, d, M, 0, SEED)
    ctx.set_problem(prob)
    _check(ctx, p0, d, family, tgt, 0, combo, tol=5e-6 if dtype == np.float32 else 1e-11, eta=100.0)
    with pytest.raises(Exception, match="COCOB"):
        device_loop(ctx, p0, "cocob", "prox", "none", 2, 70, 100.0, 1e-2, 1e-5)
    ctx.close()


def test_cocob_optimize_runs_on_the_device_and_matches_the_host_loop():
    """`optimize` with COCOB: the device-resident loop (mivi_optimize_loop, rule 4) and the host-driven `step` loop give the same parameters
    and ELBO record (the same kernels in the same order), including a warm start from the returned state."""
    d = 40
    rng = np.random.default_rng(3)
    prob, _ = make_problem(rng, "diag", d, np.float64)
    q0 = avi.MeanFieldGaussian(np.zeros(d), np.ones(d))
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=8, optimizer=avi.COCOB(), operator=avi.ClipScale(), averager=avi.PolynomialAveraging())
    qa, ia, sa = avi.optimize(avi.PhiloxRNG(SEED), alg, 12, prob, q0)
    qb, ib, sb = avi.optimize(avi.PhiloxRNG(SEED), alg, 12, prob, q0, device_loop=False)
    assert np.array_equal(qa.location, qb.location) and np.array_equal(np.asarray(qa.scale), np.asarray(qb.scale))
    assert np.array_equal([float(i["elbo"]) for i in ia], [float(i["elbo"]) for i in ib])
    rng2a, rng2b = avi.PhiloxRNG(SEED, 12), avi.PhiloxRNG(SEED, 12)
    qa2, _, _ = avi.optimize(rng2a, alg, 5, prob, q0, state=sa)
    qb2, _, _ = avi.optimize(rng2b, alg, 5, prob, q0, state=sb, device_loop=False)
    assert np.array_equal(qa2.location, qb2.location) and np.array_equal(np.asarray(qa2.scale), np.asarray(qb2.scale))


@pytest.mark.parametrize("combo", [("dowg", "clip", "poly"), ("dog", "prox", "poly"), ("descent", "prox", "poly")], ids=["dowg-clip-poly", "dog-prox-poly", "descent-prox-poly"])
@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
def test_small_logreg_loop_follows_the_oracle(combo, family):
    """k_lr_small_loop (the reference README's example class: a small hierarchical logistic regression, one sample per step)."""
    d, M = 9, 2
    rng = np.random.default_rng(13)
    prob, tgt = make_problem(rng, "logreg1", d, np.float32)
    q0 = (avi.MeanFieldGaussian(np.zeros(d, np.float32), np.full(d, 0.6, np.float32)) if family == avi.MEANFIELD
          else avi.FullRankGaussian(np.zeros(d, np.float32), (0.6 * np.eye(d)).astype(np.float32)))
    p0, _ = avi.destructure(q0)
    ctx = avi.MiviContext(np.float32, family, d, M, 0, SEED)
    ctx.set_problem(prob)
    _check(ctx, p0, d, family, tgt, 0, combo, tol=2e-5, eta=1e-3)
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_prox_keeps_a_negative_diagonal_positive(dtype):
    """mivi_prox_scale_entropy on entries below zero with a step size far below eps c^2: the exact value gamma / |c| (the reference's literal
    Float32 expression returns 0 there: tests/test_oracle_pinning.py).  The oracle keeps the reference's LITERAL expression in f64, which
    itself cancels for c < 0 (absolute error about eps64 |c|): the device is held to the exact value -- the cancellation-free form in f64 --
    at rounding, and to the oracle within the oracle's own cancellation error."""
    d = 8
    c = np.array([-3.0, -1e-3, -1.0, 0.5, 2.0, 1e-4, -40.0, 1.0])
    ctx = avi.MiviContext(dtype, avi.MEANFIELD, d, 4, 1, SEED)
    for gamma in (1e-2, 1e-9, 1e-12):
        params = np.concatenate([np.zeros(d), c]).astype(dtype)
        p = ctx.to_device(params).clone()
        ctx.prox_scale_entropy(p, gamma)
        got = p.cpu().numpy()[d:].astype(np.float64)
        c64 = params[d:].astype(np.float64)
        rt = np.sqrt(c64 * c64 + 4 * gamma)
        exact = np.where(c64 < 0, 2 * gamma / (rt - c64), c64 + (rt - c64) / 2.0)
        ref = O.proximal_location_scale_entropy(params.astype(np.float64), d, O.MEANFIELD, gamma)[d:]
        rtol = 2e-6 if dtype == np.float32 else 1e-13
        assert np.all(got > 0.0), (gamma, got)
        assert np.allclose(got, exact, rtol=rtol, atol=0.0), (gamma, got, exact)
        assert np.all(np.abs(got - ref) <= rtol * np.abs(ref) + 8 * np.finfo(np.float64).eps * np.abs(c64)), (gamma, got, ref)
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_dowg_prox_averaging_500_steps_reach_one_verdict(dtype):
    """The shape round 4's log left unexplained ("status 3: scale diagonal is not positive" at full-rank d = 1024, n_mc = 8, DoWG + Prox +
    PolynomialAveraging): un-clipped DoWG steps push diagonal entries below zero, where the literal proximal expression cancels to 0 in
    Float32.  With the stable form the device loop, the host-driven loop and (f64) the oracle trajectory all complete; the ELBO records agree."""
    d, M, T = 1024, 8, 500
    prob = avi.DiagNormalProblem(np.full(d, 5.0, dtype), np.ones(d, dtype))
    q0 = avi.FullRankGaussian(np.zeros(d, dtype), np.eye(d, dtype=dtype))
    alg = avi.KLMinRepGradProxDescent(avi.AutoMIVI(), n_samples=M, optimizer=avi.DoWG(), averager=avi.PolynomialAveraging())
    runs = []
    for dev in (True, False):
        q, info, st = avi.optimize(avi.PhiloxRNG(5), alg, T if dev else 60, prob, q0, device_loop=dev)
        runs.append(np.array([i["elbo"] for i in info]))
        assert np.all(np.isfinite(runs[-1])) and np.all(np.diag(np.asarray(q.scale)) > 0)
    assert np.allclose(runs[0][:60], runs[1], rtol=2e-4 if dtype == np.float32 else 1e-9)
    if dtype == np.float64:   # ... and the independent restatement (30 steps: the oracle's d = 1024 gradient is an O(d^2 n_mc) numpy pass per step)
        ctx = avi.MiviContext(dtype, avi.FULLRANK, d, M, 0, SEED)
        ctx.set_problem(prob)
        p0, _ = avi.destructure(q0)
        tgt = O.DiagNormalTarget(np.full(d, 5.0), np.ones(d))
        x, xbar, el = oracle_trajectory(ctx, p0, d, avi.FULLRANK, tgt, 0, "dowg", "prox", "poly", 30, 0, 0.0, avi.DoWG().alpha, 0.0)
        got, gbar, gel = device_loop(ctx, p0, "dowg", "prox", "poly", 30, 0, 0.0, avi.DoWG().alpha, 0.0)
        assert np.allclose(gel, el, rtol=1e-9) and np.max(np.abs(got - x)) <= 1e-9 * max(1.0, np.max(np.abs(x)))
        ctx.close()


def test_info_iteration_is_the_loop_index_of_this_call_on_both_routes():
    """src/optimize.jl:64-68: info = merge(info', (iteration = t,)) with t = 1 .. max_iter of THIS call -- also on a warm start, where the state's
    own counter carries on (common.jl:75).  The device loop and the host-driven loop must return the same `info` lists."""
    d, M = 64, 8
    prob = avi.DiagNormalProblem(np.full(d, 2.0, np.float32), np.ones(d, np.float32))
    q0 = avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32))
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=M, optimizer=avi.Adam(1e-2), averager=avi.NoAveraging(), operator=avi.ClipScale())
    infos = []
    for dev in (True, False):
        _, i1, st = avi.optimize(avi.PhiloxRNG(3), alg, 7, prob, q0, device_loop=dev)
        _, i2, st2 = avi.optimize(avi.PhiloxRNG(3, 7), alg, 5, prob, None, state=st, device_loop=dev)
        assert st2["iteration"] == 12
        infos.append((i1, i2))
    for a, b in zip(infos[0], infos[1]):
        assert [x["iteration"] for x in a] == [x["iteration"] for x in b] == list(range(1, len(a) + 1))
        assert np.array_equal(np.array([x["elbo"] for x in a]), np.array([x["elbo"] for x in b]))


LOST = """
import numpy as np, advancedvi_jl_amd as avi
from tests.test_gpu_loop_oracle import device_loop, oracle_trajectory
from tests.helpers import SEED, make_problem
from oracle import oracle as O
d, M = 1024, 8
rng = np.random.default_rng(11)
prob, tgt = make_problem(rng, "diag", d, np.float32)
q0 = avi.MeanFieldGaussian((0.2 * rng.normal(size=d)).astype(np.float32), rng.uniform(0.6, 1.4, size=d).astype(np.float32))
p0, _ = avi.destructure(q0)
ctx = avi.MiviContext(np.float32, avi.MEANFIELD, d, M, 0, SEED)
ctx.set_problem(prob)
assert ctx.exchange_lost() is False
got, gbar, gel = device_loop(ctx, p0, "dowg", "clip", "poly", 5, 70, 0.0, 1e-2, 1e-5)     # first call: the launch-free loop, its exchange declared lost
assert ctx.exchange_lost() is True                                                           # ... the state was restored, the steps re-run on the graph of launches
x, xbar, el = oracle_trajectory(ctx, p0, d, avi.MEANFIELD, tgt, 0, "dowg", "clip", "poly", 5, 70, 0.0, 1e-2, 1e-5)
assert np.max(np.abs(got - x)) <= 5e-6 * max(1.0, np.max(np.abs(x))) and np.max(np.abs(gbar - xbar)) <= 5e-6 * max(1.0, np.max(np.abs(xbar)))
assert np.allclose(gel, el, rtol=2e-5, atol=1e-4)
got2, _, _ = device_loop(ctx, p0, "dowg", "clip", "poly", 5, 70, 0.0, 1e-2, 1e-5)            # later calls stay on the graph route: same result, bit for bit
assert np.array_equal(got, got2)
print("ok")
"""


def test_a_lost_exchange_restores_the_state_and_falls_back_to_the_graph_route():
    """the DoG / DoWG loops exchange two norm partials grid-wide every step; if the workgroups do not run side by side the
    spin expires (status bit 8).  MIVI_FORCE_EXCHANGE_LOST=1 (a test hook: the wrapper treats the first exchanging call as lost) must leave
    the caller's parameters / optimiser state / average exactly as given, re-run the same steps on the hipGraph of launches, and pin the
    context to that route."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("MIVI_")}
    env["MIVI_FORCE_EXCHANGE_LOST"] = "1"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", LOST], env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
