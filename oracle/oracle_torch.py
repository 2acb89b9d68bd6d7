"""
CPU ORACLE, AD leg (TEST INFRASTRUCTURE ONLY -- never imported by the product path).

The reference has no hand-written gradient: `estimate_gradient!` differentiates the forward
function `estimate_repgradelbo_ad_forward` (src/algorithms/repgradelbo.jl:142-149) with an AD
backend (src/AdvancedVI.jl:57-67).  This module restates that *forward* in torch float64 and
lets torch's reverse-mode autograd play the role of the reference's AD backend, so the
closed-form VJP in `oracle.py` (and through it the HIP kernels) is checked against
"AD of the forward" -- the reference's actual definition of the gradient -- rather than against
a derivation of ours.  `q_stop` is a detached copy, exactly like `restructure(params)` held
outside the AD path at repgradelbo.jl:162.

Targets are torch re-statements of the log-densities in oracle.py (same formulas, same
reference citations); they are differentiated *through*, i.e. the LogDensityOrder{0} route of
repgradelbo.jl:50-53.  The order>=1 route (MixedADLogDensityProblem rrule,
src/mixedad_logdensity.jl:23-34) gives the same cotangent by construction: dy * grad(logpi).
"""
from __future__ import annotations

import math

import torch

from . import oracle as O

LOG2PI = math.log(2.0 * math.pi)


def _restructure(params, d, family):
    mu = params[:d]
    if family == O.MEANFIELD:
        return mu, params[d:]
    C = params[d:].reshape(d, d).T  # column-major vec -> matrix
    return mu, torch.tril(C)         # LowerTriangular projection


def _rand(mu, scale, eps):
    if scale.ndim == 1:
        return scale[:, None] * eps + mu[:, None]       # location_scale.jl:80-87
    return scale @ eps + mu[:, None]                    # location_scale.jl:71-77


def _entropy(mu, scale):
    d = mu.shape[0]
    diag = scale if scale.ndim == 1 else torch.diagonal(scale)
    return d * 0.5 * (1.0 + LOG2PI) + torch.sum(torch.log(diag))   # location_scale.jl:52-57


def _logpdf_cols(mu, scale, Z):
    R = Z - mu[:, None]
    if scale.ndim == 1:
        Zs = R / scale[:, None]
        logdet = torch.sum(torch.log(scale))
    else:
        Zs = torch.linalg.solve_triangular(scale, R, upper=False)
        logdet = torch.sum(torch.log(torch.diagonal(scale)))
    return torch.sum(-0.5 * Zs * Zs - 0.5 * LOG2PI, dim=0) - logdet  # location_scale.jl:59-63


def _estimate_entropy(kind, Z, q, q_stop):
    if kind == O.ENT_CLOSED_FORM:
        return _entropy(*q)
    if kind == O.ENT_CLOSED_FORM_ZERO_GRAD:
        return _entropy(*q_stop)
    if kind == O.ENT_MONTE_CARLO:
        return torch.mean(-_logpdf_cols(*q, Z))
    if kind == O.ENT_STL:
        return torch.mean(-_logpdf_cols(*q_stop, Z))
    if kind == O.ENT_STL_ZERO_GRAD:
        return torch.mean(-_logpdf_cols(*q_stop, Z)) - _entropy(*q) + _entropy(*q_stop)
    raise ValueError(kind)


def target_logdensity_cols(prob, Z):
    """Column-batched torch restatement of prob.logdensity for the oracle targets."""
    t = lambda a: torch.as_tensor(a, dtype=torch.float64)
    if isinstance(prob, O.DiagNormalTarget):
        r = (Z - t(prob.mean)[:, None]) / t(prob.std)[:, None]
        return -0.5 * torch.sum(r * r, dim=0) - torch.sum(torch.log(t(prob.std))) - 0.5 * Z.shape[0] * LOG2PI
    if isinstance(prob, O.DenseNormalTarget):
        r = Z - t(prob.mean)[:, None]
        return -0.5 * torch.sum(r * (t(prob.prec) @ r), dim=0) - 0.5 * prob.logdet_cov - 0.5 * Z.shape[0] * LOG2PI
    if isinstance(prob, O.LogRegTarget):
        p = prob.X.shape[1]
        beta, s = Z[:p], Z[p]
        sigma = torch.exp(s)
        logit = t(prob.X) @ beta
        y = t(prob.y)[:, None]
        loglike = torch.sum(y * logit - torch.nn.functional.softplus(logit), dim=0)
        logprior_beta = -0.5 * p * LOG2PI - p * s - 0.5 * torch.sum(beta * beta, dim=0) / sigma ** 2
        if prob.variant == "logsigma_normal":
            logprior_sigma = -0.5 * math.log(2.0 * math.pi * 9.0) - sigma ** 2 / 18.0
            jac = 0.0
        else:
            logprior_sigma = -s - math.log(3.0) - 0.5 * LOG2PI - s * s / 18.0
            jac = s
        return prob.likeadj * loglike + logprior_beta + logprior_sigma + jac
    if isinstance(prob, O.FunnelStackedTarget):
        e1, x = Z[0], Z[1:]
        n = prob.d - 1
        sv = prob.sigma_v
        log_ln = -e1 - math.log(sv) - 0.5 * LOG2PI - e1 * e1 / (2.0 * sv * sv)
        log_x = -n * e1 - 0.5 * n * LOG2PI - 0.5 * torch.sum(x * x, dim=0) * torch.exp(-2.0 * e1)
        return log_ln + log_x + e1
    raise TypeError(type(prob))


def forward(params, d, family, prob, eps, ent_kind):
    """estimate_repgradelbo_ad_forward, src/algorithms/repgradelbo.jl:142-149."""
    q = _restructure(params, d, family)
    q_stop = tuple(x.detach() for x in _restructure(params.detach(), d, family))
    Z = _rand(*q, eps)                                         # reparam_with_entropy :104-110
    ent = _estimate_entropy(ent_kind, Z, q, q_stop)
    energy = torch.mean(target_logdensity_cols(prob, Z))       # estimate_energy_with_samples :84-86
    return -(energy + ent)


def value_and_gradient(params, d, family, prob, eps, ent_kind):
    """What `_value_and_gradient!` (src/AdvancedVI.jl:57-67) writes into the DiffResult."""
    p = torch.tensor(params, dtype=torch.float64, requires_grad=True)
    e = torch.as_tensor(eps, dtype=torch.float64)
    val = forward(p, d, family, prob, e, ent_kind)
    (g,) = torch.autograd.grad(val, p)
    return float(val.detach()), g.numpy()
