"""RepGradELBO objective: host-side mirror of src/algorithms/repgradelbo.jl + src/algorithms/entropy.jl
+ src/algorithms/abstractobjective.jl (AdvancedVI.jl v0.7.0) over libmivi.

Same names, argument meaning and error behaviour as the reference for the hot path:
    RepGradELBO(n_samples; entropy)                      repgradelbo.jl:21-24, 72-74
    init(rng, obj, adtype, q, prob, params, restructure) repgradelbo.jl:41-70
    estimate_gradient_(rng, obj, adtype, out, state, params, restructure) -> (out, state, info)
                                                         repgradelbo.jl:151-177  (Julia's `estimate_gradient!`)
    estimate_objective(rng, obj, q, prob; n_samples)     repgradelbo.jl:112-122
    set_objective_state_problem(state, prob)             repgradelbo.jl:31-39
The AD backend argument is the drop-in seam: `AutoMIVI()` selects the analytic-VJP HIP
path that replaces AD of `estimate_repgradelbo_ad_forward` (repgradelbo.jl:142-149)."""
from __future__ import annotations

import logging

import numpy as np

from . import problems as P
from .context import MiviContext
from .families import MvLocationScale, destructure


_log = logging.getLogger("advancedvi_jl_amd")


# --- entropy estimators (src/algorithms/entropy.jl) -----------------------------------------------
class AbstractEntropyEstimator:
    code = -1

    def __repr__(self):
        return f"{type(self).__name__}()"


class ClosedFormEntropy(AbstractEntropyEstimator):              # entropy.jl:25-29
    code = 0


class ClosedFormEntropyZeroGradient(AbstractEntropyEstimator):  # entropy.jl:11-15
    code = 1


class MonteCarloEntropy(AbstractEntropyEstimator):              # entropy.jl:40-46
    code = 2


class StickingTheLandingEntropy(AbstractEntropyEstimator):      # entropy.jl:57-65
    code = 3


class StickingTheLandingEntropyZeroGradient(AbstractEntropyEstimator):  # entropy.jl:78-90
    code = 4


class AutoMIVI:
    """ADTypes.AbstractADType subtype selecting libmivi's closed-form VJP instead of an AD backend for the ESTIMATOR.
    `target_ad`: the backend that differentiates an ORDER-0 target's `logdensity` on the host (the reference differentiates through it
    with `adtype` itself, repgradelbo.jl:50-57; here the estimator needs no AD, so only the target's own gradient is left to provide):
    "forwarddiff" (dual numbers, `forwarddiff.py`) or None = reject order-0 targets."""

    def __init__(self, device: int = 0, target_ad="forwarddiff"):
        self.device = device
        self.target_ad = target_ad


class PhiloxRNG:
    """The `rng` threaded through the reference API.  State = (seed, next estimate index): one
    estimate consumes one index, replacing the one `rand(rng, Normal, d, M)` draw per estimate
    (src/families/location_scale.jl:76,86).  Copyable => warm starts are reproducible."""

    def __init__(self, seed: int = 0x38BEF07CF9CC549D, counter: int = 0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.counter = int(counter)

    def next_index(self) -> int:
        i = self.counter
        self.counter += 1
        return i

    def copy(self):
        return PhiloxRNG(self.seed, self.counter)


_default_rng = PhiloxRNG(0x9E3779B97F4A7C15)


def default_rng():
    return _default_rng


class DiffResult:
    """DiffResults.MutableDiffResult(value, (gradient,)): the caller-owned output buffer
    (src/algorithms/common.jl:51), device resident."""

    def __init__(self, value, gradient):
        self.value_t = value        # torch tensor [1]
        self.gradient_t = gradient  # torch tensor [params_len]

    def value(self):
        return self.value_t.item()

    def gradient(self):
        return self.gradient_t


class RepGradELBO:
    """RepGradELBO(n_samples; entropy=ClosedFormEntropy()): repgradelbo.jl:21-24,72-74."""

    def __init__(self, n_samples: int, entropy: AbstractEntropyEstimator = None):
        if not isinstance(n_samples, (int, np.integer)) or n_samples < 1:
            raise ValueError("n_samples must be a positive Int")
        self.n_samples = int(n_samples)
        self.entropy = entropy if entropy is not None else ClosedFormEntropy()
        if not isinstance(self.entropy, AbstractEntropyEstimator):
            raise TypeError("entropy must be an AbstractEntropyEstimator")

    def __repr__(self):  # Base.show, repgradelbo.jl:76-82
        return f"RepGradELBO(entropy={self.entropy}, n_samples={self.n_samples})"


class RepGradELBOState:
    """RepGradELBOState(problem, obj_ad_prep): the prepared estimator (repgradelbo.jl:26-29);
    `obj_ad_prep` is the libmivi context instead of an AD preparation."""

    def __init__(self, problem, ctx: MiviContext):
        self.problem = problem
        self.obj_ad_prep = ctx


def _order0(prob) -> bool:
    """capability < LogDensityOrder{1}() (repgradelbo.jl:50-51), looking through a TransformedProblem like README.md:115-119."""
    inner = prob.prob if isinstance(prob, P.TransformedProblem) else prob
    return not isinstance(inner, P.BUILTIN) and P.capabilities(inner) < P.LogDensityOrder(1)


def _ad_problem(adtype, prob, announce=True):
    """The capability dispatch of `init` / `set_objective_state_problem` (repgradelbo.jl:31-39, 50-62).  Order >= 1: the problem's own
    `logdensity_and_gradient` is used (the MixedADLogDensityProblem route).  Order 0: the reference differentiates through `logdensity`
    with the AD backend; AutoMIVI's estimator has none, so the TARGET is wrapped in ADgradient(adtype.target_ad, prob) -- what
    README.md:168-174 does by hand -- and the reference's @info is emitted."""
    if not _order0(prob):
        return prob
    kind = getattr(adtype, "target_ad", None)
    cap = P.capabilities(prob.prob if isinstance(prob, P.TransformedProblem) else prob)
    if kind is None:
        raise TypeError(
            f"The capability of the supplied LogDensityProblem {cap} is less than LogDensityOrder{{1}}(): "
            "AutoMIVI(target_ad=None) has no AD backend to differentiate `logdensity`; supply `logdensity_and_gradient` "
            "or construct AutoMIVI(target_ad=\"forwarddiff\").")
    if announce:
        _log.info("The capability of the supplied `LogDensityProblem` %s is less than %s. `AdvancedVI` will attempt to directly "
                  "differentiate through `LogDensityProblems.logdensity`. If this is not intended, please supply a log-density problem "
                  "with capability at least %s", cap, P.LogDensityOrder(1), P.LogDensityOrder(1))
    if isinstance(prob, P.TransformedProblem):   # the bijector stays on the device; AD only sees the constrained-scale target
        return P.TransformedProblem(P.ADgradient(kind, prob.prob), prob.bijector)
    return P.ADgradient(kind, prob)


def _make_ctx(rng, obj, adtype, q, prob, n_mc=None, entropy=None):
    if not isinstance(q, MvLocationScale):
        raise TypeError("libmivi implements the RepGradELBO path for MvLocationScale families only")
    device = getattr(adtype, "device", 0)
    ctx = MiviContext(q.eltype, q.family, len(q), n_mc or obj.n_samples,
                      (entropy or obj.entropy).code, rng.seed, device=device)
    ctx.set_problem(prob)
    return ctx


def init(rng, obj: RepGradELBO, adtype, q, prob, params, restructure) -> RepGradELBOState:
    """AdvancedVI.init(rng, obj::RepGradELBO, adtype, q, prob, params, restructure): repgradelbo.jl:41-70.
    The capability dispatch of :50-62: order >= 1 problems are used through `logdensity_and_gradient`; an order-0 problem (only
    `logdensity`: the README model, README.md:64-66, and the benchmark target, bench/benchmarks.jl:39-41) is differentiated on the host
    by `adtype.target_ad` with the reference's @info -- see `_ad_problem`."""
    if not isinstance(adtype, AutoMIVI):
        raise TypeError("adtype must be AutoMIVI() for the libmivi path")
    ad_prob = _ad_problem(adtype, prob)
    st = RepGradELBOState(ad_prob, _make_ctx(rng, obj, adtype, q, ad_prob))
    st.adtype = adtype
    return st


def set_objective_state_problem(state: RepGradELBOState, prob) -> RepGradELBOState:
    """repgradelbo.jl:31-39 (the same capability dispatch as `init`, without the @info)."""
    adtype = getattr(state, "adtype", None)
    ad_prob = _ad_problem(adtype, prob, announce=False) if adtype is not None else prob
    state.obj_ad_prep.set_problem(ad_prob)
    st = RepGradELBOState(ad_prob, state.obj_ad_prep)
    st.adtype = adtype
    return st


def estimate_gradient_(rng, obj: RepGradELBO, adtype, out: DiffResult, state: RepGradELBOState, params, restructure,
                       *args):
    """`estimate_gradient!`: repgradelbo.jl:151-177.  Writes -elbo and its gradient into `out`
    (both device resident), returns (out, state, info) with info = {"elbo": -value}.  `info["elbo"]`
    is a 0-dim device tensor so the call stays asynchronous; `float()` it to synchronise."""
    ctx = state.obj_ad_prep
    ctx.estimate_gradient(params, rng.next_index(), out.value_t, out.gradient_t)
    info = {"elbo": -out.value_t[0]}
    return out, state, info


def estimate_objective(rng, obj: RepGradELBO, q, prob, n_samples: int = None, adtype=None, _ctx_cache={}):
    """estimate_objective(rng, obj::RepGradELBO, q, prob; n_samples): repgradelbo.jl:112-118 (q_stop := q).
    Returns the NEGATIVE elbo as a Python float."""
    if isinstance(rng, RepGradELBO):  # estimate_objective(obj, q, prob): default rng, repgradelbo.jl:120-122
        rng, obj, q, prob = default_rng(), rng, obj, q
    n = int(n_samples) if n_samples is not None else obj.n_samples
    ctx = _make_ctx(rng, obj, adtype or AutoMIVI(), q, prob, n_mc=min(n, 16384))
    try:
        params, _ = destructure(q)
        v = ctx.estimate_objective(params, rng.next_index(), n_samples=n, entropy=obj.entropy.code)
        return float(v.item())
    finally:
        ctx.close()


def rand(rng, q: MvLocationScale, num_samples: int, device: int = 0):
    """rand(rng, q, num_samples): d x num_samples samples (location_scale.jl:71-87), returned as a
    device tensor (column m = sample m)."""
    ctx = MiviContext(q.eltype, q.family, len(q), num_samples, 0, rng.seed, device=device)
    try:
        params, _ = destructure(q)
        return ctx.sample(params, rng.next_index(), want_eps=False).clone()
    finally:
        ctx.close()


def gaussian_expectation_gradient_and_hessian_(rng, q: MvLocationScale, n_samples: int, grad_buf, hess_buf, prob,
                                               adtype=None, _ctx=None):
    """`gaussian_expectation_gradient_and_hessian!(rng, q, n_samples, grad_buf, hess_buf, prob)`:
    src/algorithms/gauss_expected_grad_hess.jl:20-60, the Stein / Price-identity branch (what a target with
    `logdensity_and_gradient` takes, :32-60).  `grad_buf` (d) / `hess_buf` (d*d, column-major) are device tensors
    that are overwritten, or None to allocate.  Returns (logpi_avg: float, grad (d), hess (d, d)) -- `hess` is not
    symmetrised, like the reference's.  A target whose `capabilities` exceed LogDensityOrder(1) -- a plugin with
    `logdensity_gradient_and_hessian`, or a built-in Gaussian problem constructed with order=2 -- takes the reference's second-order
    branch (:61-83): the sample average of the Hessians, no Stein identity (mivi_gauss_expected_grad_hess2)."""
    if not isinstance(q, MvLocationScale) or q.family != 1:
        raise TypeError("gaussian_expectation_gradient_and_hessian_ expects a Gaussian with a triangular scale "
                        "(gauss_expected_grad_hess.jl:22)")
    ctx = _ctx
    if ctx is None:
        ctx = MiviContext(q.eltype, q.family, len(q), min(int(n_samples), 16384), 0, rng.seed,
                          device=getattr(adtype, "device", 0))
        ctx.set_problem(prob)
    try:
        params, _ = destructure(q)
        from .problems import LogDensityOrder, capabilities
        second = LogDensityOrder(1) < capabilities(prob)          # (the reference's branch test, gauss_expected_grad_hess.jl:31-32)
        logpi, g, H = ctx.gauss_expected_grad_hess(params, rng.next_index(), int(n_samples), grad_buf, hess_buf, second_order=second)
        return float(logpi.item()), g, H
    finally:
        if _ctx is None:
            ctx.close()
