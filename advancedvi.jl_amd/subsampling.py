"""Data subsampling around the RepGradELBO hot path.

    ReshufflingBatchSubsampling    src/reshuffling.jl:13-60
    SubsampledObjective            src/algorithms/subsampledobjective.jl:11-90

The objective state keeps ONE libmivi context; a step only re-points its target at the minibatch
(`set_objective_state_problem`, repgradelbo.jl:31-39) -- for the built-in logistic regression that is a row gather on
the device (mivi_logreg_select_rows), for plugin targets it is whatever the model's `subsample` returns.
"""
from dataclasses import dataclass

import numpy as np

from . import objectives as O
from .families import destructure
from .problems import subsample


class ReshufflingBatchSubsampling:
    """Random reshuffling: every epoch shuffles the data set, splits it into batches of `batchsize` (the last one may be
    shorter) and walks through them once.  src/reshuffling.jl:13-33."""

    def __init__(self, dataset, batchsize: int):
        self.dataset = np.asarray(dataset)
        self.batchsize = int(batchsize)
        if self.batchsize < 1 or self.dataset.ndim != 1 or self.dataset.size == 0:
            raise ValueError("dataset must be a non-empty vector and batchsize >= 1")

    def __len__(self):
        return -(-self.dataset.size // self.batchsize)   # ceil(n / batchsize), reshuffling.jl:23-25


@dataclass
class ReshufflingBatchSubsamplingState:
    epoch: int
    batches: list      # remaining (sub_step, batch) pairs of the current epoch, in order


def _host_generator(rng):
    """The shuffles consume the same `rng` object as the estimates (draw order is part of the contract):
    each shuffle takes one index off the Philox counter and seeds a host generator with (seed, index)."""
    return np.random.Generator(np.random.Philox(key=rng.seed, counter=[rng.next_index(), 0, 0, 0x5ABF1E]))


def reshuffle_batches(rng, sub: ReshufflingBatchSubsampling):
    """reshuffling.jl:27-32"""
    shuffled = _host_generator(rng).permutation(sub.dataset)
    return [(i + 1, shuffled[k:k + sub.batchsize]) for i, k in enumerate(range(0, shuffled.size, sub.batchsize))]


def init_subsampling(rng, sub: ReshufflingBatchSubsampling):
    """init(rng, sub): reshuffling.jl:34-36"""
    return ReshufflingBatchSubsamplingState(1, reshuffle_batches(rng, sub))


def step_subsampling(rng, sub: ReshufflingBatchSubsampling, state: ReshufflingBatchSubsamplingState,
                     drop_trailing_batch_if_too_small: bool = False):
    """step(rng, sub, state, drop_trailing_batch_if_too_small): reshuffling.jl:38-60.  Returns (batch, state, info).
    When the batch just taken was the last of its epoch the next epoch is shuffled immediately, and -- gradient
    estimation only -- a trailing batch shorter than `batchsize` is replaced by the first batch of that next epoch."""
    epoch, batches = state.epoch, list(state.batches)
    sub_step, batch = batches.pop(0)
    if not batches:
        batches = reshuffle_batches(rng, sub)
        if drop_trailing_batch_if_too_small and len(batch) < sub.batchsize:
            sub_step, batch = batches.pop(0)
        epoch += 1
    return batch, ReshufflingBatchSubsamplingState(epoch, batches), {"epoch": epoch, "step": sub_step}


class SubsampledObjective:
    """SubsampledObjective(objective, subsampling): subsampledobjective.jl:11-15"""

    def __init__(self, objective, subsampling):
        self.objective = objective
        self.subsampling = subsampling

    @property
    def n_samples(self):
        return self.objective.n_samples

    @property
    def entropy(self):
        return self.objective.entropy


@dataclass
class SubsampledObjectiveState:
    prob: object
    sub_st: ReshufflingBatchSubsamplingState
    obj_st: object


def init(rng, subobj: SubsampledObjective, adtype, q_init, prob, params, restructure):
    """subsampledobjective.jl:23-45.  The inner objective is prepared on a minibatch-conditioned problem; the
    subsampling state is NOT advanced by that peek (the reference discards the peeked state too)."""
    sub_st = init_subsampling(rng, subobj.subsampling)
    batch, _, _ = step_subsampling(rng, subobj.subsampling, sub_st, True)
    prob_sub = subsample(prob, batch)
    q_sub = subsample(q_init, batch)
    params_sub, re_sub = destructure(q_sub)
    obj_st = O.init(rng, subobj.objective, adtype, q_sub, prob_sub, params_sub, re_sub)
    return SubsampledObjectiveState(prob, sub_st, obj_st)


def estimate_objective(rng, subobj: SubsampledObjective, q, prob, n_samples: int = None, adtype=None):
    """Average of the inner objective over one pass through the batches: subsampledobjective.jl:47-58.  One context
    serves all batches (the data set is uploaded once, each batch is a row selection)."""
    if isinstance(rng, SubsampledObjective):   # default-rng overload, subsampledobjective.jl:60-62
        rng, subobj, q, prob = O.default_rng(), rng, subobj, q
    sub, obj = subobj.subsampling, subobj.objective
    n = int(n_samples) if n_samples is not None else obj.n_samples
    sub_st = init_subsampling(rng, sub)
    params, _ = destructure(q)
    ctx = None
    total = 0.0
    try:
        for _ in range(len(sub)):
            batch, sub_st, _ = step_subsampling(rng, sub, sub_st)
            prob_sub = subsample(prob, batch)
            if ctx is None:
                ctx = O._make_ctx(rng, obj, adtype or O.AutoMIVI(), q, prob_sub, n_mc=min(n, 16384))
            else:
                ctx.set_problem(prob_sub)
            total += float(ctx.estimate_objective(params, rng.next_index(), n_samples=n, entropy=obj.entropy.code).item())
    finally:
        if ctx is not None:
            ctx.close()
    return total / len(sub)


def estimate_gradient_(rng, subobj: SubsampledObjective, adtype, out, state: SubsampledObjectiveState, params,
                       restructure, *args):
    """`estimate_gradient!`: subsampledobjective.jl:64-90 -- take the next batch, condition the problem on it, run the
    inner estimator.  `info` merges the subsampling info (epoch, step) with the inner one (elbo)."""
    batch, sub_st, sub_inf = step_subsampling(rng, subobj.subsampling, state.sub_st, True)
    prob_sub = subsample(state.prob, batch)
    obj_st = O.set_objective_state_problem(state.obj_st, prob_sub)
    out, obj_st, obj_inf = O.estimate_gradient_(rng, subobj.objective, adtype, out, obj_st, params, restructure, *args)
    return out, SubsampledObjectiveState(state.prob, sub_st, obj_st), {**sub_inf, **obj_inf}
