"""Callers of the hot path: KLMinRepGradDescent (= ADVI), init / step / output and `optimize`.
Host-side mirror of src/algorithms/constructors.jl:44-79, src/algorithms/common.jl:29-120 and
src/optimize.jl:42-94 (AdvancedVI.jl v0.7.0); parameters stay resident in HBM between steps and every
update is a libmivi kernel."""
from __future__ import annotations

import warnings

import numpy as np

from . import objectives as O
from . import subsampling as S
from .families import MvLocationScale, destructure


# --- operators (src/optimization/clip_scale.jl, src/AdvancedVI.jl:173-204) ------------------------
class IdentityOperator:
    def apply(self, ctx, params, optimizer=None, opt_st=None):
        return params


class ClipScale:
    """ClipScale(eps = 1e-5): src/optimization/clip_scale.jl:8-29."""

    def __init__(self, epsilon=1e-5):
        self.epsilon = epsilon

    def apply(self, ctx, params, optimizer=None, opt_st=None):
        ctx.clip_scale(params, self.epsilon)
        return params


class ProximalLocationScaleEntropy:
    """Proximal operator of the entropy of a location-scale family, applied after the optimiser step with that step's
    size: src/optimization/proximal_location_scale_entropy.jl:17-61.  Supports Descent, DoG, DoWG (:26-42); the DoG/DoWG
    step size is read from the optimiser state on the device."""

    def apply(self, ctx, params, optimizer=None, opt_st=None):
        if isinstance(optimizer, DoG):            # DoWG is a subclass
            ctx.prox_scale_entropy(params, 0.0, opt_st, optimizer.kind)
        elif isinstance(optimizer, Descent):
            ctx.prox_scale_entropy(params, optimizer.eta)
        else:
            raise TypeError(f"`ProximalLocationScaleEntropy` does not support optimization rule {type(optimizer).__name__}.")
        return params


# --- optimisation rules (Optimisers.jl rules used by the reference's tests/bench) -----------------
class Descent:
    def __init__(self, eta=0.1):
        self.eta = eta

    def setup(self, ctx, params):
        return None

    def update(self, ctx, state, params, grad, t):
        ctx.descent_update(params, grad, self.eta)
        return state


class Adam:
    def __init__(self, eta=1e-3, beta=(0.9, 0.999), epsilon=1e-8):
        self.eta, self.beta, self.epsilon = eta, beta, epsilon

    def setup(self, ctx, params):
        import torch
        return torch.zeros(2 * params.numel(), dtype=params.dtype, device=params.device)

    def update(self, ctx, state, params, grad, t):
        ctx.adam_update(params, grad, state, t, self.eta, self.beta[0], self.beta[1], self.epsilon)
        return state


class DoG:
    """src/optimization/rules.jl:48-64."""
    kind = 0

    def __init__(self, alpha=1e-6):
        self.alpha = alpha

    def setup(self, ctx, params):
        st = ctx.dog_state()
        ctx.dog_init(params, st, self.alpha)
        return st

    def update(self, ctx, state, params, grad, t):
        ctx.dog_update(params, grad, state, self.kind)
        return state


class DoWG(DoG):
    """src/optimization/rules.jl:17-34."""
    kind = 1


class COCOB:
    """src/optimization/rules.jl:78-96 (COCOB-Backprop); state (L, G, R, theta, x1) = (0, 0, 0, 0, params) (:84-86).
    Device-resident in `optimize` (mivi_optimize_loop rule 4: a hipGraph of chained estimates + the update kernel, no host round trip per
    step); `update` below is the host-driven `step` loop's launch."""

    def __init__(self, alpha=100):
        self.alpha = alpha

    def setup(self, ctx, params):
        import torch
        n = params.numel()
        st = torch.zeros(5 * n, dtype=params.dtype, device=params.device)
        st[4 * n:] = params
        return st

    def update(self, ctx, state, params, grad, t):
        ctx.cocob_update(params, grad, state, self.alpha)
        return state


# --- averagers (src/optimization/averaging.jl) -----------------------------------------------------
class NoAveraging:
    def init(self, ctx, params):
        return params

    def apply(self, ctx, state, params):
        return params

    def value(self, state):
        return state


class PolynomialAveraging:
    """x_bar_t = (1 - w_t) x_bar_{t-1} + w_t x_t, w_t = (eta+1)/(t+eta): averaging.jl:36-53."""

    def __init__(self, eta=8):
        self.eta = eta

    def init(self, ctx, params):
        return (params.clone(), 1)

    def apply(self, ctx, state, params):
        x_bar, t = state
        w = (self.eta + 1) / (t + self.eta)
        ctx.axpby(x_bar, w, params, 1.0 - w)
        return (x_bar, t + 1)

    def value(self, state):
        return state[0]


class KLMinRepGradDescent:
    """KLMinRepGradDescent(adtype; entropy, optimizer, n_samples, averager, operator): constructors.jl:44-79."""

    def __init__(self, adtype, entropy=None, optimizer=None, n_samples: int = 1, averager=None, operator=None,
                 subsampling=None):
        entropy = entropy if entropy is not None else O.ClosedFormEntropy()
        if not isinstance(entropy, (O.ClosedFormEntropy, O.StickingTheLandingEntropy, O.MonteCarloEntropy)):
            raise TypeError("entropy must be ClosedFormEntropy, StickingTheLandingEntropy or MonteCarloEntropy")
        self.objective = O.RepGradELBO(n_samples, entropy=entropy)
        if subsampling is not None:   # constructors.jl:69-73
            self.objective = S.SubsampledObjective(self.objective, subsampling)
        self.adtype = adtype
        self.optimizer = optimizer if optimizer is not None else DoWG()
        self.averager = averager if averager is not None else PolynomialAveraging()
        self.operator = operator if operator is not None else IdentityOperator()


ADVI = KLMinRepGradDescent


class KLMinRepGradProxDescent(KLMinRepGradDescent):
    """KLMinRepGradProxDescent(adtype; entropy_zerograd, optimizer, n_samples, averager): constructors.jl:122-157.
    The entropy is handled by the proximal operator, so the gradient estimator must ignore it: the entropy estimator is
    one of the two *ZeroGradient kinds; optimizer one of Descent / DoG / DoWG."""

    def __init__(self, adtype, entropy_zerograd=None, optimizer=None, n_samples: int = 1, averager=None, subsampling=None):
        entropy = entropy_zerograd if entropy_zerograd is not None else O.ClosedFormEntropyZeroGradient()
        if not isinstance(entropy, (O.ClosedFormEntropyZeroGradient, O.StickingTheLandingEntropyZeroGradient)):
            raise TypeError("entropy_zerograd must be ClosedFormEntropyZeroGradient or StickingTheLandingEntropyZeroGradient")
        optimizer = optimizer if optimizer is not None else DoWG()
        if not isinstance(optimizer, (Descent, DoG)):
            raise TypeError("optimizer must be Descent, DoG or DoWG")
        self.objective = O.RepGradELBO(n_samples, entropy=entropy)
        if subsampling is not None:   # constructors.jl:145-149
            self.objective = S.SubsampledObjective(self.objective, subsampling)
        self.adtype = adtype
        self.optimizer = optimizer
        self.averager = averager if averager is not None else PolynomialAveraging()
        self.operator = ProximalLocationScaleEntropy()


def _obj_init(rng, obj, *a):
    return (S.init if isinstance(obj, S.SubsampledObjective) else O.init)(rng, obj, *a)


def _obj_estimate_gradient(rng, obj, *a):
    return (S.estimate_gradient_ if isinstance(obj, S.SubsampledObjective) else O.estimate_gradient_)(rng, obj, *a)


def _ctx_of(obj_st):
    return obj_st.obj_st.obj_ad_prep if isinstance(obj_st, S.SubsampledObjectiveState) else obj_st.obj_ad_prep


def estimate_objective(rng, alg, q, prob, n_samples=None, entropy=None):
    """estimate_objective([rng,] alg, q, prob; n_samples, entropy=MonteCarloEntropy()): common.jl:29-38."""
    if isinstance(rng, KLMinRepGradDescent):
        rng, alg, q, prob = O.default_rng(), rng, alg, q
    n = n_samples if n_samples is not None else alg.objective.n_samples
    ent = entropy if entropy is not None else O.MonteCarloEntropy()
    if isinstance(alg.objective, S.SubsampledObjective):
        sub = S.SubsampledObjective(O.RepGradELBO(n, entropy=ent), alg.objective.subsampling)
        return S.estimate_objective(rng, sub, q, prob, adtype=alg.adtype)
    return O.estimate_objective(rng, O.RepGradELBO(n, entropy=ent), q, prob, adtype=alg.adtype)


def init(rng, alg: KLMinRepGradDescent, q_init, prob):
    """init(rng, alg::ParamSpaceSGD, q_init, prob): common.jl:40-61."""
    if isinstance(q_init, MvLocationScale) and isinstance(alg.operator, IdentityOperator):
        warnings.warn(
            "IdentityOperator is used with a variational family <:MvLocationScale. Optimization can easily fail under "
            "this combination due to singular scale matrices. Consider using the operator `ClipScale` in the algorithm "
            "instead.")
    params_h, re = destructure(q_init)
    obj_st = _obj_init(rng, alg.objective, alg.adtype, q_init, prob, params_h, re)
    ctx = _ctx_of(obj_st)
    params = ctx.to_device(params_h).clone()
    opt_st = alg.optimizer.setup(ctx, params)
    avg_st = alg.averager.init(ctx, params)
    grad_buf = O.DiffResult(ctx.empty(1), ctx.empty(ctx.params_len))
    return dict(prob=prob, q=q_init, params=params, restructure=re, iteration=0, grad_buf=grad_buf, opt_st=opt_st,
                obj_st=obj_st, avg_st=avg_st)


def output(alg, state):
    """output(alg, state) = re(value(averager, avg_st)): common.jl:63-67."""
    return state["restructure"](alg.averager.value(state["avg_st"]).cpu().numpy())


def step(rng, alg, state, callback, *objargs):
    """step(rng, alg::ParamSpaceSGD, state, callback): common.jl:69-120."""
    state = dict(state)
    state["iteration"] += 1
    t = state["iteration"]
    ctx = _ctx_of(state["obj_st"])
    params, re = state["params"], state["restructure"]
    grad_buf, obj_st, info = _obj_estimate_gradient(rng, alg.objective, alg.adtype, state["grad_buf"], state["obj_st"],
                                                  params, re, *objargs)
    state["obj_st"] = obj_st
    value = grad_buf.value()          # host sync, like the reference's eager isfinite check
    if not np.isfinite(value):        # common.jl:83-89
        raise RuntimeError(f"The objective value is {value}. This indicates that the optimization run diverged.")
    grad = grad_buf.gradient()
    state["opt_st"] = alg.optimizer.update(ctx, state["opt_st"], params, grad, t)   # Optimisers.update!
    params = alg.operator.apply(ctx, params, alg.optimizer, state["opt_st"])   # apply(operator, typeof(q), opt_st, params, re)
    state["avg_st"] = alg.averager.apply(ctx, state["avg_st"], params)
    state["params"] = params
    state["q"] = None  # materialised lazily by `output` / callbacks (params are device resident)
    info = {**{k: v for k, v in info.items() if k != "elbo"}, "elbo": -value}   # subsampling adds (epoch, step)
    if callback is not None:
        extra = callback(rng=rng, iteration=t, restructure=re, params=params,
                         averaged_params=alg.averager.value(state["avg_st"]), gradient=grad, state=state)
        if extra is not None:
            info = {**extra, **info}
    return state, False, info


_RULES = {Descent: 0, Adam: 1, DoG: 2, DoWG: 3, COCOB: 4}
_OPS = {IdentityOperator: 0, ClipScale: 1, ProximalLocationScaleEntropy: 2}
_AVGS = {NoAveraging: 0, PolynomialAveraging: 1}
DEVICE_LOOP_CHUNK = 256   # iterations per mivi_optimize_loop call (bounds the work done past a divergence)


def _device_loop_codes(alg, callback, objargs):
    """(rule, op, averager) when a whole `step` can run inside mivi_optimize_loop, else None: no callback (it needs the
    parameters on the host every iteration), a plain RepGradELBO, and rule / operator / averager of the exact reference
    types (subclasses may override behaviour)."""
    if callback is not None or objargs or not isinstance(alg.objective, O.RepGradELBO):
        return None
    r, o, a = _RULES.get(type(alg.optimizer)), _OPS.get(type(alg.operator)), _AVGS.get(type(alg.averager))
    if r is None or o is None or a is None or (o == 2 and r in (1, 4)):
        return None
    return r, o, a


def _optimize_on_device(rng, alg, max_iter, state, codes, show_progress):
    """The loop of `optimize` with every iteration on the device (the same parameters as the host-driven loop: bitwise for Descent / Adam on the
    mean-field loops and the graph route, to rounding for the launch-free loops with their own summation order -- DoG / DoWG, the row-owning
    full-rank loops, the small logistic regression).
    Returns None when the target cannot be captured (host-callback targets): the caller then takes the host loop."""
    from ._lib import MiviError
    rule, op, avg = codes
    ctx = _ctx_of(state["obj_st"])
    params = state["params"]
    opt, info_total = alg.optimizer, []
    done = 0
    import torch

    def _tensors(x):   # the device tensors inside an optimiser / averager state (tensor, tuple of tensors and scalars, None)
        if torch.is_tensor(x):
            return [x]
        if isinstance(x, (tuple, list)):
            return [t for e in x for t in _tensors(e)]
        return []

    while done < max_iter:
        n = min(DEVICE_LOOP_CHUNK, max_iter - done)
        elbo = ctx.empty(n)
        avg_params = state["avg_st"][0] if avg == 1 else None
        # what a divergence inside the chunk must not destroy: the reference throws AT the offending step, before
        # Optimisers.update! (common.jl:83-94), leaving every earlier step applied
        live = [params] + _tensors(state["opt_st"]) + _tensors(state["avg_st"])
        snap = [t.clone() for t in live]
        try:
            ctx.optimize_loop(params, n, rng.counter, state["iteration"], rule=rule, op=op, averager=avg,
                              eta=getattr(opt, "alpha", 0.0) if rule == 4 else getattr(opt, "eta", 0.0), beta=getattr(opt, "beta", (0.9, 0.999)),
                              adam_eps=getattr(opt, "epsilon", 1e-8), clip_epsilon=getattr(alg.operator, "epsilon", 0.0),
                              avg_eta=getattr(alg.averager, "eta", 8), opt_state=state["opt_st"], avg_params=avg_params,
                              elbo=elbo)
        except MiviError as e:
            if e.status == 6 and done == 0:      # MIVI_ERR_UNSUPPORTED: not a device-resident target
                return None
            if e.status in (2, 3):               # common.jl:83-89 (a non-positive scale makes the objective NaN)
                # the chunk ran past the bad step: restore its entry state and replay it on the host-driven loop, which raises
                # at the offending iteration with the steps before it applied and rng / iteration advanced consistently
                for t_live, t_snap in zip(live, snap):
                    t_live.copy_(t_snap)
                try:
                    ctx.synchronize()            # drop the sticky device flag of the failed chunk
                except MiviError:
                    pass
                try:
                    for k in range(n):
                        new_state, _, info = step(rng, alg, state, None)
                        state.update(new_state)
                        info_total.append({**info, "iteration": done + k + 1})   # (the loop index of THIS call: optimize.jl:64-68)
                except MiviError as e2:          # the host replay met the same device flag: the documented exception type
                    if e2.status in (2, 3):
                        raise RuntimeError("The objective value is not finite. This indicates that the optimization run "
                                           "diverged.") from e2
                    raise
                # the host-driven replay of the chunk completed with finite objectives: the device flag was not reproducible.
                # (mivi_optimize_loop clears every status word it later reads, so a stale flag of an earlier call cannot cause
                # this any more; if it happens, it is worth knowing.)  For Descent / Adam the replayed steps ARE the chunk (bitwise the
                # same arithmetic); for the launch-free loops that agree with the host loop to rounding only (DoG / DoWG's norm sums, the
                # row-owning full-rank loops, the small logistic regression) this chunk then comes from host arithmetic and the run carries
                # on from it -- a flag that does not reproduce is therefore warned about, never silently dropped.
                warnings.warn(f"device optimisation chunk reported status {e.status} ({e}) but its host-driven replay completed with "
                              "finite objectives; continuing from the replayed steps", RuntimeWarning)
                done += n
                continue
            raise
        for _ in range(n):
            rng.next_index()
        vals = elbo.cpu().numpy()
        # info = merge(info', (iteration = t,)) with t the loop index of THIS `optimize` call (src/optimize.jl:64-68) -- not the state's
        # cumulative counter, which a warm start carries on (common.jl:75): both routes report the same `info`
        info_total += [{"elbo": float(vals[i]), "iteration": done + i + 1} for i in range(n)]
        state["iteration"] = state["iteration"] + n
        state["avg_st"] = (state["avg_st"][0], state["avg_st"][1] + n) if avg == 1 else params
        done += n
        if show_progress:
            print(f"\rOptimizing {done}/{max_iter} elbo={info_total[-1]['elbo']:.6g}", end="" if done < max_iter else "\n")
    state["q"] = None
    return info_total


def optimize(rng, algorithm, max_iter: int, prob=None, q_init=None, *objargs, show_progress=False, state=None,
             callback=None, device_loop=True):
    """optimize([rng,] algorithm, max_iter, prob, q_init; show_progress, state, callback): src/optimize.jl:42-94.
    Returns (output, info, state).  Without a callback and with a device-resident target every iteration runs inside
    mivi_optimize_loop (`device_loop=False` forces the host-driven `step` loop; both give the same result -- bitwise or to rounding, see
    _optimize_on_device)."""
    if isinstance(rng, KLMinRepGradDescent):   # default-rng overload, optimize.jl:83-94
        extra = (q_init,) if q_init is not None else ()
        rng, algorithm, max_iter, prob, q_init = O.default_rng(), rng, algorithm, max_iter, prob
        objargs = extra + objargs
    info_total = []
    if state is None:
        state = init(rng, algorithm, q_init, prob)
    codes = _device_loop_codes(algorithm, callback, objargs) if device_loop else None
    if codes is not None and max_iter > 0:
        state = dict(state)
        info_dev = _optimize_on_device(rng, algorithm, max_iter, state, codes, show_progress)
        if info_dev is not None:
            return output(algorithm, state), info_dev, state
    for t in range(1, max_iter + 1):
        state, terminate, info = step(rng, algorithm, state, callback, *objargs)
        info = {**info, "iteration": t}
        if terminate:
            break
        if show_progress:
            print(f"\rOptimizing {t}/{max_iter} elbo={info['elbo']:.6g}", end="" if t < max_iter else "\n")
        info_total.append(info)
    return output(algorithm, state), info_total, state
