"""
CPU ORACLE (TEST INFRASTRUCTURE ONLY -- never imported by the product path).

A float64 NumPy restatement of the AdvancedVI.jl v0.7.0 RepGradELBO hot path.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; `advancedvi.jl_amd/` must not (and fails loudly without its
HIP library instead of falling back to anything here).

Pinning status: the reference is pure Julia, Julia is not installed in the build
container, and the reference ships NO golden numeric vectors.
The oracle is therefore pinned against the reference's own *known-answer* tests
(tests/test_oracle_pinning.py restates each one, with the reference file:line):
  * STL gradient == 0 at q = pi            test/algorithms/klminrepgraddescent.jl:66-87
  * estimate_objective(q = pi) ~ 0         test/algorithms/klminrepgraddescent.jl:36-37
  * entropy(q) == entropy(MvNormal)        test/families/location_scale.jl:44-47
  * logpdf(q, z) == logpdf(MvNormal, z)    test/families/location_scale.jl:38-42
  * sample mean / var / cov                test/families/location_scale.jl:68-97
  * mean-field destructure length 2d       test/families/location_scale.jl:146-155
  * rrule seam returns plugin gradient     test/general/mixedad_logdensity.jl:37-61
and its analytic gradient is cross-checked against reverse-mode AD (torch CPU
autograd standing in for the reference's AD backends) of the *forward* function
restated line by line in `oracle_torch.py`.  Parity against real Julia output is
"unpinned" (no Julia toolchain); see DESIGN.md.

All `file:line` citations are relative to /root/reference.

Randomness: the reference draws `eps = rand(rng, Normal{T}(0,1), d, M)` column-major
(src/families/location_scale.jl:76,86).  Julia's RNG stream cannot be reproduced
here, so every function below takes `eps` (d x M, one sample per column,
src/utils.jl:6) as an explicit input: "identical RNG streams" is defined at the
eps level.  `philox4x32_10` / `philox_normal_block` restate the counter-based
generator the HIP kernels use so eps itself can be checked.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

LOG2PI = math.log(2.0 * math.pi)

# entropy estimator ids == include/mivi.h mivi_entropy_t
ENT_CLOSED_FORM = 0          # ClosedFormEntropy                      src/algorithms/entropy.jl:25-29
ENT_CLOSED_FORM_ZERO_GRAD = 1  # ClosedFormEntropyZeroGradient        src/algorithms/entropy.jl:11-15
ENT_MONTE_CARLO = 2          # MonteCarloEntropy                      src/algorithms/entropy.jl:40-46
ENT_STL = 3                  # StickingTheLandingEntropy              src/algorithms/entropy.jl:57-65
ENT_STL_ZERO_GRAD = 4        # StickingTheLandingEntropyZeroGradient  src/algorithms/entropy.jl:78-90

MEANFIELD = 0
FULLRANK = 1


# --------------------------------------------------------------------------------------
# Variational family: MvLocationScale with a standard normal base distribution
# --------------------------------------------------------------------------------------
@dataclass
class MvLocationScale:
    """src/families/location_scale.jl:15-19.  `scale` is a length-d vector (Diagonal,
    mean-field) or a d x d lower-triangular matrix (full-rank)."""

    location: np.ndarray
    scale: np.ndarray

    @property
    def is_meanfield(self) -> bool:
        return self.scale.ndim == 1

    @property
    def d(self) -> int:
        return self.location.shape[0]


def destructure(q: MvLocationScale) -> np.ndarray:
    """Flat trainable parameters.
    mean-field: [location; diag(scale)]          src/families/location_scale.jl:39-43
    full-rank : [location; vec(scale)] column-major with the strict upper triangle zero
                (default Optimisers.destructure over `@functor (location, scale)`,
                src/families/location_scale.jl:21; layout fixed explicitly in mivi.h)."""
    if q.is_meanfield:
        return np.concatenate([q.location, q.scale])
    return np.concatenate([q.location, np.tril(q.scale).reshape(-1, order="F")])


def restructure(params: np.ndarray, d: int, family: int) -> MvLocationScale:
    """RestructureMeanField, src/families/location_scale.jl:32-37; full-rank re-projects
    onto LowerTriangular (entries above the diagonal are ignored)."""
    params = np.asarray(params, dtype=np.float64)
    if family == MEANFIELD:
        assert params.shape[0] == 2 * d
        return MvLocationScale(params[:d].copy(), params[d:].copy())
    assert params.shape[0] == d + d * d
    C = params[d:].reshape(d, d, order="F")
    return MvLocationScale(params[:d].copy(), np.tril(C))


def rand_batch(q: MvLocationScale, eps: np.ndarray) -> np.ndarray:
    """`rand(rng, q, M)`: samples = scale * eps .+ location, d x M, one sample per column.
    dense: src/families/location_scale.jl:71-77; Diagonal: :80-87."""
    if q.is_meanfield:
        return q.scale[:, None] * eps + q.location[:, None]
    return q.scale @ eps + q.location[:, None]


def entropy_closed_form(q: MvLocationScale) -> float:
    """StatsBase.entropy(q) = d * entropy(Normal(0,1)) + logdet(scale)
    src/families/location_scale.jl:52-57."""
    diag = q.scale if q.is_meanfield else np.diag(q.scale)
    return q.d * 0.5 * (1.0 + LOG2PI) + float(np.sum(np.log(diag)))


def _solve_scale(q: MvLocationScale, r: np.ndarray) -> np.ndarray:
    if q.is_meanfield:
        return r / (q.scale if r.ndim == 1 else q.scale[:, None])
    # forward substitution with the lower-triangular scale ("scale \\ (z - location)")
    return np.linalg.solve(np.tril(q.scale), r)


def logpdf(q: MvLocationScale, z: np.ndarray) -> float:
    """Distributions.logpdf(q, z) = sum(logpdf.(Normal(0,1), scale \\ (z - location))) - logdet(scale)
    src/families/location_scale.jl:59-63."""
    z_std = _solve_scale(q, z - q.location)
    diag = q.scale if q.is_meanfield else np.diag(q.scale)
    return float(np.sum(-0.5 * z_std * z_std - 0.5 * LOG2PI) - np.sum(np.log(diag)))


def estimate_entropy(kind: int, samples: np.ndarray, q: MvLocationScale, q_stop: MvLocationScale) -> float:
    """The five `estimate_entropy` methods, src/algorithms/entropy.jl:13-15, 27-29, 42-46,
    59-65, 80-90 (for MonteCarloEntropy the more specific method at :42 wins dispatch)."""
    M = samples.shape[1]
    if kind == ENT_CLOSED_FORM:
        return entropy_closed_form(q)
    if kind == ENT_CLOSED_FORM_ZERO_GRAD:
        return entropy_closed_form(q_stop)
    if kind == ENT_MONTE_CARLO:
        return float(np.mean([-logpdf(q, samples[:, m]) for m in range(M)]))
    if kind == ENT_STL:
        return float(np.mean([-logpdf(q_stop, samples[:, m]) for m in range(M)]))
    if kind == ENT_STL_ZERO_GRAD:
        ent_stl = float(np.mean([-logpdf(q_stop, samples[:, m]) for m in range(M)]))
        return ent_stl - entropy_closed_form(q) + entropy_closed_form(q_stop)
    raise ValueError(kind)


# --------------------------------------------------------------------------------------
# Target log-densities (the LogDensityProblems plugin side)
# --------------------------------------------------------------------------------------
class DiagNormalTarget:
    """MvNormal(mean, Diagonal(std.^2)); the reference's mean-field test target
    (test/models/normal.jl:56-75) and bench target (bench/benchmarks.jl:43-47)."""

    def __init__(self, mean, std):
        self.mean = np.asarray(mean, dtype=np.float64)
        self.std = np.asarray(std, dtype=np.float64)

    def dimension(self):
        return self.mean.shape[0]

    def logdensity(self, z):
        r = (z - self.mean) / self.std
        return float(-0.5 * np.sum(r * r) - np.sum(np.log(self.std)) - 0.5 * self.dimension() * LOG2PI)

    def logdensity_and_gradient(self, z):
        return self.logdensity(z), -(z - self.mean) / (self.std ** 2)

    def logdensity_gradient_and_hessian(self, z):
        return self.logdensity(z), -(z - self.mean) / (self.std ** 2), -np.diag(1.0 / self.std ** 2)


class DenseNormalTarget:
    """MvNormal(mean, L L'), `TestNormal` with a dense covariance: test/models/normal.jl:2-11, 36-54."""

    def __init__(self, mean, L):
        self.mean = np.asarray(mean, dtype=np.float64)
        self.L = np.tril(np.asarray(L, dtype=np.float64))
        self.cov = self.L @ self.L.T
        self.prec = np.linalg.inv(self.cov)
        self.logdet_cov = 2.0 * float(np.sum(np.log(np.diag(self.L))))

    def dimension(self):
        return self.mean.shape[0]

    def logdensity(self, z):
        r = z - self.mean
        return float(-0.5 * r @ self.prec @ r - 0.5 * self.logdet_cov - 0.5 * self.dimension() * LOG2PI)

    def logdensity_and_gradient(self, z):
        return self.logdensity(z), -self.prec @ (z - self.mean)

    def logdensity_gradient_and_hessian(self, z):
        return self.logdensity(z), -self.prec @ (z - self.mean), -self.prec


def _log1pexp(x):
    return np.logaddexp(0.0, x)


class LogRegTarget:
    """Hierarchical logistic regression, theta = [beta (p); s] with
        beta ~ MvNormal(0, sigma^2 I),  y ~ BernoulliLogit(X beta)
    variant "logsigma_normal" (docs/src/tutorials/subsampling.md:26-38):
        s = log sigma, sigma = exp(s), logprior_sigma = logpdf(Normal(0,3), sigma), no Jacobian,
        likelihood scaled by likeadj = n_data / n.
    variant "lognormal_exp_bijector" (README.md:42-66 wrapped by the TransformedLogDensityProblem
        of README.md:91-106 with Bijectors.Stacked([identity, log-bijector])):
        sigma = exp(s), logprior_sigma = logpdf(LogNormal(0,3), sigma), + logabsdetjac = s.
    """

    def __init__(self, X, y, variant="logsigma_normal", likeadj=1.0, keep_storage=False):
        # keep_storage: X stays in the dtype it was given in (float32 data of 10^6 rows: 2 GB instead of 4); only the batched
        # evaluation, which promotes every row chunk to f64, may then be used
        self.X = np.asarray(X) if keep_storage else np.asarray(X, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)
        self.variant = variant
        self.likeadj = float(likeadj)

    def dimension(self):
        return self.X.shape[1] + 1

    def logdensity_and_gradient(self, z):
        if self.X.dtype != np.float64:
            raise TypeError("LogRegTarget(keep_storage=True): use logdensity_and_gradient_batch")
        p = self.X.shape[1]
        beta, s = z[:p], z[p]
        sigma = math.exp(s)
        logit = self.X @ beta
        loglike = float(np.sum(self.y * logit - _log1pexp(logit)))
        resid = self.y - 1.0 / (1.0 + np.exp(-logit))
        bb = float(beta @ beta)
        logprior_beta = -0.5 * p * LOG2PI - p * s - 0.5 * bb / sigma ** 2
        g = np.empty(p + 1)
        g[:p] = self.likeadj * (self.X.T @ resid) - beta / sigma ** 2
        g_s = -p + bb / sigma ** 2
        if self.variant == "logsigma_normal":
            logprior_sigma = -0.5 * math.log(2.0 * math.pi * 9.0) - sigma ** 2 / 18.0
            g_s += -(sigma ** 2) / 9.0
            jac = 0.0
        elif self.variant == "lognormal_exp_bijector":
            logprior_sigma = -s - math.log(3.0) - 0.5 * LOG2PI - s * s / 18.0
            g_s += -1.0 - s / 9.0
            jac = s
            g_s += 1.0
        else:
            raise ValueError(self.variant)
        g[p] = g_s
        return self.likeadj * loglike + logprior_beta + logprior_sigma + jac, g

    def logdensity(self, z):
        return self.logdensity_and_gradient(z)[0]

    def logdensity_gradient_and_hessian(self, z):
        """LogDensityProblems.logdensity_gradient_and_hessian of the same model (what a LogDensityOrder{2} problem provides to
        gaussian_expectation_gradient_and_hessian!, src/algorithms/gauss_expected_grad_hess.jl:61-83): the gradient's derivative, written out;
        pinned by finite differences of `logdensity_and_gradient` (tests/test_oracle_pinning.py)."""
        val, g = self.logdensity_and_gradient(z)
        p = self.X.shape[1]
        beta, s = z[:p], z[p]
        is2 = math.exp(-2.0 * s)
        pi = 1.0 / (1.0 + np.exp(-(self.X @ beta)))
        H = np.zeros((p + 1, p + 1))
        H[:p, :p] = -self.likeadj * (self.X.T * (pi * (1.0 - pi))) @ self.X - is2 * np.eye(p)
        H[:p, p] = H[p, :p] = 2.0 * beta * is2
        H[p, p] = -2.0 * float(beta @ beta) * is2 + (-2.0 * math.exp(2.0 * s) / 9.0 if self.variant == "logsigma_normal" else -1.0 / 9.0)
        return val, g, H

    def logdensity_and_gradient_batch(self, Z, row_chunk=65536):
        """The same function for all columns of Z (d x M) at once -- `logdensity_and_gradient` column by column, with the two data products
        as matrix products over row chunks (X may be float32 storage: every chunk is promoted to f64 before it is used).  What makes the
        n = 10^6 configuration (BASELINE configs[2]) checkable in seconds; equal to the per-column loop to rounding (tests/test_oracle_c.py)."""
        Z = np.asarray(Z, dtype=np.float64)
        p, M = self.X.shape[1], Z.shape[1]
        B, s = Z[:p], Z[p]
        sigma = np.exp(s)
        loglike = np.zeros(M)
        xtr = np.zeros((p, M))
        for lo in range(0, self.X.shape[0], row_chunk):
            Xc = np.asarray(self.X[lo:lo + row_chunk], dtype=np.float64)
            yc = self.y[lo:lo + row_chunk, None]
            logit = Xc @ B
            loglike += np.sum(yc * logit - _log1pexp(logit), axis=0)
            xtr += Xc.T @ (yc - 1.0 / (1.0 + np.exp(-logit)))
        bb = np.sum(B * B, axis=0)
        logprior_beta = -0.5 * p * LOG2PI - p * s - 0.5 * bb / sigma ** 2
        G = np.empty((p + 1, M))
        G[:p] = self.likeadj * xtr - B / sigma ** 2
        g_s = -p + bb / sigma ** 2
        if self.variant == "logsigma_normal":
            logprior_sigma = -0.5 * math.log(2.0 * math.pi * 9.0) - sigma ** 2 / 18.0
            g_s = g_s - (sigma ** 2) / 9.0
            jac = 0.0
        elif self.variant == "lognormal_exp_bijector":
            logprior_sigma = -s - math.log(3.0) - 0.5 * LOG2PI - s * s / 18.0
            g_s = g_s - 1.0 - s / 9.0 + 1.0
            jac = s
        else:
            raise ValueError(self.variant)
        G[p] = g_s
        return self.likeadj * loglike + logprior_beta + logprior_sigma + jac, G

    def subsample(self, idx):
        """AdvancedVI.subsample for the tutorial's LogReg (docs/src/tutorials/subsampling.md:99-102): the rows `idx`,
        likelihood rescaled by n_data / n (:37)."""
        idx = np.asarray(idx, dtype=np.int64)
        return LogRegTarget(self.X[idx], self.y[idx], self.variant, self.likeadj * self.X.shape[0] / idx.size)


class FunnelStackedTarget:
    """Neal's funnel on its constrained scale (defined in SURVEY.md section 8d -- the reference has
    no funnel): s ~ LogNormal(0, sigma_v), x_i ~ Normal(0, s) (std-dev s), theta = [s; x],
    unconstrained via binv = inverse(Stacked([log-bijector, identity], [1:1, 2:d])) exactly as
    README.md:76-82,102-106 does: s = exp(eta_1), logabsdetjac = eta_1."""

    def __init__(self, d, sigma_v=1.5):
        self.d = d
        self.sigma_v = float(sigma_v)

    def dimension(self):
        return self.d

    def logdensity_and_gradient(self, eta):
        e1 = eta[0]
        x = eta[1:]
        n = self.d - 1
        sv2 = self.sigma_v ** 2
        log_lognormal = -e1 - math.log(self.sigma_v) - 0.5 * LOG2PI - e1 * e1 / (2.0 * sv2)
        inv_s2 = math.exp(-2.0 * e1)
        sx2 = float(x @ x)
        log_x = -n * e1 - 0.5 * n * LOG2PI - 0.5 * sx2 * inv_s2
        val = log_lognormal + log_x + e1
        g = np.empty(self.d)
        g[0] = (-1.0 - e1 / sv2) + (-n + sx2 * inv_s2) + 1.0
        g[1:] = -x * inv_s2
        return val, g

    def logdensity(self, eta):
        return self.logdensity_and_gradient(eta)[0]

    def logdensity_gradient_and_hessian(self, eta):
        """The arrow-shaped Hessian of the unconstrained funnel (finite-difference pinned in tests/test_oracle_pinning.py)."""
        val, g = self.logdensity_and_gradient(eta)
        e2, x = math.exp(-2.0 * eta[0]), eta[1:]
        H = np.diag(np.concatenate([[-1.0 / self.sigma_v ** 2 - 2.0 * e2 * float(x @ x)], np.full(self.d - 1, -e2)]))
        H[0, 1:] = H[1:, 0] = 2.0 * e2 * x
        return val, g, H


class FunnelConstrainedTarget:
    """The same funnel on its CONSTRAINED scale, theta = [s; x], s > 0, no bijector:
    log p = log LogNormal(s; 0, sigma_v) + sum_i log Normal(x_i; 0, s).  StackedBijectorTarget(this, exp on [0, 1)) is
    FunnelStackedTarget (pinned in tests/test_oracle_pinning.py)."""

    def __init__(self, d, sigma_v=1.5):
        self.d = d
        self.sigma_v = float(sigma_v)

    def dimension(self):
        return self.d

    def logdensity_and_gradient(self, theta):
        s = theta[0]
        x = theta[1:]
        n = self.d - 1
        sv2 = self.sigma_v ** 2
        ls = math.log(s)
        sx2 = float(x @ x)
        val = (-ls - math.log(self.sigma_v) - 0.5 * LOG2PI - ls * ls / (2.0 * sv2)) + (-n * ls - 0.5 * n * LOG2PI - 0.5 * sx2 / (s * s))
        g = np.empty(self.d)
        g[0] = (-1.0 - ls / sv2 - n + sx2 / (s * s)) / s
        g[1:] = -x / (s * s)
        return val, g

    def logdensity(self, theta):
        return self.logdensity_and_gradient(theta)[0]

    def logdensity_gradient_and_hessian(self, theta):
        val, g = self.logdensity_and_gradient(theta)
        s, x = theta[0], theta[1:]
        sx2 = float(x @ x)
        H = np.diag(np.concatenate([[self.d / s ** 2 - (1.0 - math.log(s)) / (self.sigma_v ** 2 * s ** 2) - 3.0 * sx2 / s ** 4], np.full(self.d - 1, -1.0 / s ** 2)]))
        H[0, 1:] = H[1:, 0] = 2.0 * x / s ** 3
        return val, g, H


class StackedBijectorTarget:
    """TransformedLogDensityProblem(prob, binv) with binv = inverse(Bijectors.Stacked(...)) over index blocks
    (README.md:76-82, 91-119; docs/src/tutorials/constrained.md:154-196):
        logdensity(eta) = logdensity(prob, binv(eta)) + logabsdetjac(binv, eta)
    blocks: list of (begin, end, kind), 0-based half-open, kind "identity" or "exp" (binv = exp: x = exp(eta),
    logabsdetjac = eta).  The gradient follows by the chain rule: J' g + 1 on exp coordinates."""

    def __init__(self, inner, blocks):
        self.inner = inner
        self.mask = np.zeros(inner.dimension(), dtype=bool)
        for lo, hi, kind in blocks:
            if kind == "exp":
                self.mask[lo:hi] = True
            elif kind != "identity":
                raise ValueError(kind)

    def dimension(self):
        return self.inner.dimension()

    def logdensity_and_gradient(self, eta):
        eta = np.asarray(eta, dtype=np.float64)
        x = np.where(self.mask, np.exp(eta), eta)
        v, g = self.inner.logdensity_and_gradient(x)
        return v + float(np.sum(eta[self.mask])), np.where(self.mask, x * g + 1.0, g)

    def logdensity(self, eta):
        eta = np.asarray(eta, dtype=np.float64)
        x = np.where(self.mask, np.exp(eta), eta)
        return self.inner.logdensity(x) + float(np.sum(eta[self.mask]))


# --------------------------------------------------------------------------------------
# RepGradELBO
# --------------------------------------------------------------------------------------
def estimate_energy_with_samples(prob, samples: np.ndarray) -> float:
    """mean(logdensity(prob, z_m) for z_m in eachcol(samples)), src/algorithms/repgradelbo.jl:84-86."""
    return float(np.mean([prob.logdensity(samples[:, m]) for m in range(samples.shape[1])]))


def reparam_with_entropy(q, q_stop, eps, ent_kind):
    """src/algorithms/repgradelbo.jl:104-110."""
    samples = rand_batch(q, eps)
    return samples, estimate_entropy(ent_kind, samples, q, q_stop)


def estimate_objective(q: MvLocationScale, prob, eps: np.ndarray, ent_kind: int = ENT_MONTE_CARLO) -> float:
    """estimate_objective(rng, obj::RepGradELBO, q, prob; n_samples): q_stop := q, returns the
    NEGATIVE elbo.  src/algorithms/repgradelbo.jl:112-118; the algorithm-level wrapper defaults to
    MonteCarloEntropy (src/algorithms/common.jl:29-38)."""
    samples, ent = reparam_with_entropy(q, q, eps, ent_kind)
    return -(estimate_energy_with_samples(prob, samples) + ent)


def estimate_repgradelbo_forward(params, d, family, prob, eps, ent_kind, q_stop=None) -> float:
    """estimate_repgradelbo_ad_forward(params, aux) = -(energy + entropy),
    src/algorithms/repgradelbo.jl:142-149 (q_stop = restructure(params), :162)."""
    q = restructure(params, d, family)
    if q_stop is None:
        q_stop = restructure(params, d, family)
    samples, ent = reparam_with_entropy(q, q_stop, eps, ent_kind)
    return -(estimate_energy_with_samples(prob, samples) + ent)


def gaussian_expectation_gradient_and_hessian(q: MvLocationScale, prob, u: np.ndarray):
    r"""gaussian_expectation_gradient_and_hessian!, first-order (Stein / Price) branch:
    src/algorithms/gauss_expected_grad_hess.jl:32-60.  `u` (d x n) are the standard-normal draws,
    z = C u + m; per sample the loop accumulates logpi/n, grad/n and u * (grad/n)'; finally
    hess = C' \ hess.  Returns (logpi_avg, grad (d), hess (d x d))."""
    if q.is_meanfield:
        raise TypeError("the reference method takes a triangular scale (gauss_expected_grad_hess.jl:22)")
    d, n = u.shape
    C = np.tril(q.scale)
    z = C @ u + q.location[:, None]
    logpi_avg = 0.0
    grad = np.zeros(d)
    hess = np.zeros((d, d))
    for b in range(n):
        lp, g = prob.logdensity_and_gradient(z[:, b])
        logpi_avg += lp / n
        g = np.asarray(g, dtype=np.float64) / n
        grad += g
        hess += np.outer(u[:, b], g)
    hess = np.linalg.solve(C.T, hess)
    return float(logpi_avg), grad, hess


def gaussian_expectation_gradient_and_hessian_order2(q: MvLocationScale, prob, u: np.ndarray):
    """gaussian_expectation_gradient_and_hessian!, second-order branch (targets with `logdensity_gradient_and_hessian`):
    src/algorithms/gauss_expected_grad_hess.jl:61-83.  z = rand(rng, q, n) = C u + m for the standard-normal draws `u` (d x n);
    per sample the loop accumulates logpi / n, grad / n and hess / n -- the naive sample averages, no Stein identity."""
    if q.is_meanfield:
        raise TypeError("the reference method takes a triangular scale (gauss_expected_grad_hess.jl:22)")
    d, n = u.shape
    z = np.tril(q.scale) @ u + q.location[:, None]
    logpi_avg = 0.0
    grad = np.zeros(d)
    hess = np.zeros((d, d))
    for b in range(n):
        lp, g, h = prob.logdensity_gradient_and_hessian(z[:, b])
        logpi_avg += lp / n
        grad += np.asarray(g, dtype=np.float64) / n
        hess += np.asarray(h, dtype=np.float64) / n
    return float(logpi_avg), grad, hess


def c_inv_t_eps(q: MvLocationScale, eps: np.ndarray) -> np.ndarray:
    """C^{-T} eps == -grad_z log q_stop(z) at z = mu + C eps  (SURVEY.md section 3.4)."""
    if q.is_meanfield:
        return eps / q.scale[:, None]
    return np.linalg.solve(np.tril(q.scale).T, eps)


def estimate_gradient(params, d, family, prob, eps, ent_kind, batch_target=False):
    """What `estimate_gradient!` (src/algorithms/repgradelbo.jl:151-177) leaves in `out`:
    value = -elbo and gradient = d(value)/d(params), written out in closed form
    (checked against AD of the forward in tests/test_oracle_pinning.py).

    Returns dict(value, grad, elbo, Z, ell, G, W, entropy, partials) where `partials` is the
    un-normalised shard-additive buffer of include/mivi.h:
        mean-field: [sum_m W_im (d); sum_m W_im eps_im (d); sum_m ell_m; sum_m 0.5|eps_m|^2]
        full-rank : [sum_m W_im (d); vec(tril(sum_m W_im eps_jm)) (d*d); sum ell; sum 0.5|eps|^2]
    """
    q = restructure(params, d, family)
    M = eps.shape[1]
    Z = rand_batch(q, eps)
    ell = np.empty(M)
    G = np.empty((d, M))
    if batch_target and hasattr(prob, "logdensity_and_gradient_batch"):   # (the same arithmetic with the data products blocked: large data sets)
        ell, G = prob.logdensity_and_gradient_batch(Z)
    else:
        for m in range(M):
            ell[m], G[:, m] = prob.logdensity_and_gradient(Z[:, m])
    diag = q.scale if q.is_meanfield else np.diag(q.scale)
    ent_cf = entropy_closed_form(q)
    half_eps2 = 0.5 * np.sum(eps * eps, axis=0)
    ent_mc = float(np.mean(half_eps2)) + 0.5 * d * LOG2PI + float(np.sum(np.log(diag)))
    if ent_kind in (ENT_CLOSED_FORM, ENT_CLOSED_FORM_ZERO_GRAD):
        ent = ent_cf
    else:
        ent = ent_mc
    stl = ent_kind in (ENT_STL, ENT_STL_ZERO_GRAD)
    W = G + (c_inv_t_eps(q, eps) if stl else 0.0)
    direct = {ENT_CLOSED_FORM: 1.0, ENT_CLOSED_FORM_ZERO_GRAD: 0.0, ENT_MONTE_CARLO: 1.0,
              ENT_STL: 0.0, ENT_STL_ZERO_GRAD: -1.0}[ent_kind]
    g_mu = -W.sum(axis=1) / M
    if q.is_meanfield:
        P_scale = (W * eps).sum(axis=1)
        g_scale = -P_scale / M - direct / diag
        grad = np.concatenate([g_mu, g_scale])
        partials = np.concatenate([W.sum(axis=1), P_scale, [ell.sum()], [half_eps2.sum()]])
    else:
        P_scale = np.tril(W @ eps.T)
        gC = -P_scale / M - direct * np.diag(1.0 / diag)
        grad = np.concatenate([g_mu, gC.reshape(-1, order="F")])
        packed = np.concatenate([P_scale[j:, j] for j in range(d)])       # lower triangle, column by column
        partials = np.concatenate([W.sum(axis=1), packed, [ell.sum()], [half_eps2.sum()]])
    value = -(float(np.mean(ell)) + ent)
    return dict(value=value, grad=grad, elbo=-value, Z=Z, ell=ell, G=G, W=W, entropy=ent, partials=partials)


def finalize_partials(partials, params, d, family, ent_kind, M_total):
    """Host restatement of the finalize step applied after the all-reduce of shard partials
    (SURVEY.md section 8e): scale by -1/M_total, add the parameter-only entropy terms once."""
    q = restructure(params, d, family)
    diag = q.scale if q.is_meanfield else np.diag(q.scale)
    direct = {ENT_CLOSED_FORM: 1.0, ENT_CLOSED_FORM_ZERO_GRAD: 0.0, ENT_MONTE_CARLO: 1.0,
              ENT_STL: 0.0, ENT_STL_ZERO_GRAD: -1.0}[ent_kind]
    sum_ell, sum_half_eps2 = partials[-2], partials[-1]
    logdet = float(np.sum(np.log(diag)))
    if ent_kind in (ENT_CLOSED_FORM, ENT_CLOSED_FORM_ZERO_GRAD):
        ent = d * 0.5 * (1.0 + LOG2PI) + logdet
    else:
        ent = sum_half_eps2 / M_total + 0.5 * d * LOG2PI + logdet
    g_mu = -partials[:d] / M_total
    if family == MEANFIELD:
        g_scale = -partials[d:2 * d] / M_total - direct / diag
        grad = np.concatenate([g_mu, g_scale])
    else:
        P = np.zeros((d, d))
        off = d
        for j in range(d):                                                # unpack column j (rows j..d-1)
            P[j:, j] = partials[off:off + d - j]
            off += d - j
        gC = -P / M_total - direct * np.diag(1.0 / diag)
        grad = np.concatenate([g_mu, gC.reshape(-1, order="F")])
    return -(sum_ell / M_total + ent), grad


def engine_partials(partials, d):
    """The batch engine's internal layout of a full-rank partial vector (csrc/kernels_fullrank_batch.hip k_fb_vjp<PART>; d a multiple of 128):
    [sum_m W (d) | the lower triangle of sum_m W (x) eps as its 128 x 128 tiles, tile (rb, cb <= rb) at d + (rb (rb + 1) / 2 + cb) 128^2,
    column-major inside, zeros above the diagonal of the diagonal tiles | sum ell | sum |eps|^2 / 2 | pad to a multiple of four], from the
    C ABI's column-packed vector `partials` (estimate_gradient(...)["partials"])."""
    T = d // 128
    P = np.zeros((d, d))
    off = d
    for j in range(d):
        P[j:, j] = partials[off:off + d - j]
        off += d - j
    n = (d + T * (T + 1) // 2 * 16384 + 2 + 3) // 4 * 4
    out = np.zeros(n)
    out[:d] = partials[:d]
    for rb in range(T):
        for cb in range(rb + 1):
            t = d + (rb * (rb + 1) // 2 + cb) * 16384
            out[t:t + 16384] = P[128 * rb:128 * rb + 128, 128 * cb:128 * cb + 128].reshape(-1, order="F")
    so = d + T * (T + 1) // 2 * 16384
    out[so], out[so + 1] = partials[-2], partials[-1]
    return out


def finalize_engine_partials(ep, params, d, ent_kind, M_total):
    """k_fb_finalize_parts restated: the (summed) engine-layout partial vector -> (value, dense gradient with exact zeros above the diagonal)."""
    T = d // 128
    P = np.zeros((d, d))
    for rb in range(T):
        for cb in range(rb + 1):
            t = d + (rb * (rb + 1) // 2 + cb) * 16384
            P[128 * rb:128 * rb + 128, 128 * cb:128 * cb + 128] = ep[t:t + 16384].reshape(128, 128, order="F")
    packed = np.concatenate([np.tril(P)[j:, j] for j in range(d)])
    so = d + T * (T + 1) // 2 * 16384
    return finalize_partials(np.concatenate([ep[:d], packed, ep[so:so + 2]]), params, d, FULLRANK, ent_kind, M_total)


def finalize_slice(slice_sum, g0, params, d, family, ent_kind, M_total, L):
    """Host restatement of mivi_finalize_slice: the packed final values of partial-vector elements [g0, g0 + n) given their
    sums over the ranks.  Gradient entries: -(1/M) sum - direct * [diagonal] / C_ii; element L-2 becomes the
    objective value, L-1 the status bits; padding beyond L is zero."""
    slice_sum = np.asarray(slice_sum, dtype=np.float64)
    n = slice_sum.shape[0]
    direct = {ENT_CLOSED_FORM: 1.0, ENT_CLOSED_FORM_ZERO_GRAD: 0.0, ENT_MONTE_CARLO: 1.0, ENT_STL: 0.0,
              ENT_STL_ZERO_GRAD: -1.0}[ent_kind]
    out = np.zeros(n)
    if family == MEANFIELD:
        diag_of = {d + i: params[d + i] for i in range(d)}
    else:
        diag_of = {}
        C = params[d:].reshape(d, d, order="F")
        for j in range(d):
            diag_of[d + j * d - (j * (j - 1)) // 2] = C[j, j]
    for t in range(n):
        g = g0 + t
        if g >= L - 2:
            continue
        v = -slice_sum[t] / M_total
        if g in diag_of:
            v -= direct / diag_of[g]
        out[t] = v
    if g0 <= L - 2 and g0 + n >= L:
        so = L - 2 - g0
        diag = params[d:] if family == MEANFIELD else np.diag(params[d:].reshape(d, d, order="F"))
        s_ld = float(np.sum(np.log(diag)))
        ent = (0.5 * d * (1.0 + LOG2PI) if ent_kind in (ENT_CLOSED_FORM, ENT_CLOSED_FORM_ZERO_GRAD)
               else slice_sum[so + 1] / M_total + 0.5 * d * LOG2PI) + s_ld
        out[so] = -(slice_sum[so] / M_total + ent)
        out[so + 1] = 0.0
    return out


def unpack_final(packed, d, family):
    """Host restatement of mivi_unpack_final: packed final vector -> (value, gradient in the parameter layout)."""
    L = (2 * d if family == MEANFIELD else d + d * (d + 1) // 2) + 2
    if family == MEANFIELD:
        return float(packed[L - 2]), np.array(packed[:2 * d], dtype=np.float64)
    gC = np.zeros((d, d))
    for j in range(d):
        o = d + j * d - (j * (j - 1)) // 2
        gC[j:, j] = packed[o:o + d - j]
    return float(packed[L - 2]), np.concatenate([packed[:d], gC.reshape(-1, order="F")])


def p2p_geometry(L, world):
    """Restatement of the peer-to-peer exchange geometry (csrc/api_dist.hip p2p_geometry, exported as mivi_p2p_geometry):
    slice length n (a multiple of 4, the two scalars L-2 / L-1 in ONE slice), chunk length cn, chunk count G, value-owner rank."""
    n = ((L + world - 1) // world + 3) & ~3
    while (L - 1) % n == 0:
        n += 4
    vs = (L - 2) // n
    G = min(255, max(1, (n + 511) // 512))
    cn = ((n + G - 1) // G + 3) & ~3
    return n, cn, G, vs


def p2p_exchange(partials_by_rank, params, d, family, ent_kind, M_total, epochs=1):
    """Host restatement of k_p2p_exchange (csrc/kernels_p2p.hip), phase by phase and chunk by chunk, with every rank's staging / final
    areas as explicit arrays double-buffered by epoch parity: push (rank r stores chunk g of slice s into stage[s][parity][r]),
    reduce (owner s sums the contributions in rank order, finalises, stores the final chunk into fin[every rank][parity]), unpack.
    `partials_by_rank`: list over epochs of lists over ranks (or one list over ranks, reused).  Returns per epoch a list of
    (value, grad) per rank."""
    R = len(partials_by_rank[0]) if isinstance(partials_by_rank[0], (list, tuple)) else len(partials_by_rank)
    per_epoch = partials_by_rank if isinstance(partials_by_rank[0], (list, tuple)) else [partials_by_rank] * epochs
    L = (2 * d if family == MEANFIELD else d + d * (d + 1) // 2) + 2
    n, cn, G, vs = p2p_geometry(L, R)
    Lp = n * R
    stage = [np.full((2, R, n), np.nan) for _ in range(R)]     # NaN: an element nobody pushed must never be consumed
    fin = [np.full((2, Lp), np.nan) for _ in range(R)]
    direct = {ENT_CLOSED_FORM: 1.0, ENT_CLOSED_FORM_ZERO_GRAD: 0.0, ENT_MONTE_CARLO: 1.0, ENT_STL: 0.0,
              ENT_STL_ZERO_GRAD: -1.0}[ent_kind]
    if family == MEANFIELD:
        diag_at = {d + i: params[d + i] for i in range(d)}
        diag = np.asarray(params[d:], dtype=np.float64)
    else:
        C = np.asarray(params[d:], dtype=np.float64).reshape(d, d, order="F")
        diag_at = {d + j * d - (j * (j - 1)) // 2: C[j, j] for j in range(d)}
        diag = np.diag(C)
    out = []
    for ep, parts in enumerate(per_epoch, start=1):
        p = ep & 1
        padded = [np.concatenate([np.asarray(x, dtype=np.float64), np.zeros(Lp - L)]) for x in parts]
        for r in range(R):                                       # phase 1
            for g in range(G):
                c0 = g * cn
                clen = max(0, min(cn, n - c0))
                for s in range(R):
                    stage[s][p, r, c0:c0 + clen] = padded[r][s * n + c0:s * n + c0 + clen]
        for s in range(R):                                       # phase 2
            for g in range(G):
                c0 = g * cn
                clen = max(0, min(cn, n - c0))
                acc = np.zeros(clen)
                for src in range(R):                             # rank order
                    acc += stage[s][p, src, c0:c0 + clen]
                o = -acc / M_total
                for k in range(clen):
                    gi = s * n + c0 + k
                    if gi >= L - 2:
                        o[k] = np.nan                            # scalars: the value workgroup; padding: never read
                    elif gi in diag_at:
                        o[k] -= direct / diag_at[gi]
                for r in range(R):
                    keep = np.array([s * n + c0 + k < L - 2 for k in range(clen)], dtype=bool)
                    fin[r][p, s * n + c0:s * n + c0 + clen][keep] = o[keep]
        o0 = L - 2 - vs * n                                      # the value workgroup of rank vs
        sum_ell = sum(stage[vs][p, src, o0] for src in range(R))
        s_he = sum(stage[vs][p, src, o0 + 1] for src in range(R))
        s_ld = float(np.sum(np.log(diag)))
        ent = (0.5 * d * (1.0 + LOG2PI) if ent_kind in (ENT_CLOSED_FORM, ENT_CLOSED_FORM_ZERO_GRAD)
               else s_he / M_total + 0.5 * d * LOG2PI) + s_ld
        for r in range(R):
            fin[r][p, L - 2] = -(sum_ell / M_total + ent)
            fin[r][p, L - 1] = 0.0
        out.append([unpack_final(fin[r][p], d, family) for r in range(R)])   # phase 3
    return out


# --------------------------------------------------------------------------------------
# Host-side operators next to the hot path (section 8f)
# --------------------------------------------------------------------------------------
def clip_scale(params, d, family, epsilon=1e-5):
    """ClipScale: scale[diagind] = max(scale[diagind], eps); src/optimization/clip_scale.jl:18-29."""
    out = np.array(params, dtype=np.float64, copy=True)
    if family == MEANFIELD:
        out[d:] = np.maximum(out[d:], epsilon)
    else:
        idx = d + np.arange(d) * (d + 1)
        out[idx] = np.maximum(out[idx], epsilon)
        # restructure -> destructure re-projects onto LowerTriangular
        C = np.tril(out[d:].reshape(d, d, order="F"))
        out[d:] = C.reshape(-1, order="F")
    return out


def proximal_location_scale_entropy(params, d, family, stepsize):
    """ProximalLocationScaleEntropy: diag(scale) <- c + (sqrt(c^2 + 4 gamma) - c) / 2, the minimiser of
    -log|det L'| + ||L' - L||^2 / (2 gamma); src/optimization/proximal_location_scale_entropy.jl:44-61."""
    out = np.array(params, dtype=np.float64, copy=True)
    idx = d + np.arange(d) if family == MEANFIELD else d + np.arange(d) * (d + 1)
    c = out[idx]
    # the reference's literal expression, in f64 (no cancellation problem at any magnitude a test uses).  The DEVICE evaluates, for c < 0,
    # 2 gamma / (sqrt(c^2 + 4 gamma) - c): equal in exact arithmetic, but positive in Float32 where the literal form cancels to 0 -- a
    # deliberate, documented behavioural difference; tests assert the two agree to rounding.
    out[idx] = c + (np.sqrt(c * c + 4.0 * stepsize) - c) / 2.0
    if family != MEANFIELD:   # restructure -> destructure re-projects onto LowerTriangular
        out[d:] = np.tril(out[d:].reshape(d, d, order="F")).reshape(-1, order="F")
    return out


def stepsize_from_optimizer_state(rule, eta=None, v=None, r=None):
    """Descent -> eta; DoG -> r / sqrt(v); DoWG -> r^2 / sqrt(v); proximal_location_scale_entropy.jl:26-42."""
    if rule == "descent":
        return float(eta)
    if rule == "dog":
        return float(r) / np.sqrt(float(v))
    if rule == "dowg":
        return float(r) * float(r) / np.sqrt(float(v))
    raise ValueError(f"`ProximalLocationScaleEntropy` does not support optimization rule {rule}.")


def dog_step(params, grad, state, kind):
    """One DoG (kind 0) / DoWG (kind 1) update; state = (x0, v, r); src/optimization/rules.jl:17-64."""
    x0, v, r = state
    params = np.asarray(params, dtype=np.float64)
    r = max(float(np.sqrt(np.sum((params - x0) ** 2))), r)
    g2 = float(np.sum(np.asarray(grad, dtype=np.float64) ** 2))
    if kind == 1:
        v = v + r * r * g2
        eta = r * r / np.sqrt(v)
    else:
        v = v + g2
        eta = r / np.sqrt(v)
    return params - eta * np.asarray(grad, dtype=np.float64), (x0, v, r)


# --------------------------------------------------------------------------------------
# Optimisers.jl rules used by `step` (src/algorithms/common.jl:92: `Optimisers.update!(opt_st, params, grad)`).
# Optimisers.jl is a dependency of the reference, NOT vendored in /root/reference (Project.toml:49 pins
# "0.2.16, 0.3, 0.4"); what follows restates its published rules (Optimisers.jl src/rules.jl, `apply!`):
#   Descent(eta):            dx' = eta * dx;                                   x <- x - dx'
#   Adam(eta, (b1, b2), eps): mt <- b1 mt + (1 - b1) dx;  vt <- b2 vt + (1 - b2) dx^2;
#                            dx' = mt / (1 - b1^t) / (sqrt(vt / (1 - b2^t)) + eps) * eta;   x <- x - dx'
#                            (state (mt, vt, beta^t) starts at (0, 0, beta): the first step divides by 1 - beta)
# Arithmetic is carried in `dtype` (Optimisers keeps the parameter eltype), so an f32 run can be followed to the ulp level.
# --------------------------------------------------------------------------------------
def descent_step(params, grad, eta, dtype=np.float64):
    x = np.asarray(params, dtype=dtype)
    return (x - dtype(eta) * np.asarray(grad, dtype=dtype)).astype(dtype)


def cocob_init(x):
    """Optimisers.init(::COCOB, x): (L, G, R, theta, x1) = (0, 0, 0, 0, copy(x))  (src/optimization/rules.jl:84-86)."""
    z = np.zeros_like(x)
    return (z.copy(), z.copy(), z.copy(), z.copy(), x.copy())


def cocob_step(x, dx, state, alpha=100.0):
    """Optimisers.apply!(::COCOB, ...) followed by x .-= dx' (src/optimization/rules.jl:88-96).  Coordinates with L == 0 (no non-zero
    gradient seen yet) are left unchanged: the reference's expression is 0/0 there."""
    L, G, R, th, x1 = state
    L = np.maximum(L, np.abs(dx))
    G = G + np.abs(dx)
    R = np.maximum(R + (x - x1) * -dx, 0.0)
    th = th + -dx
    with np.errstate(invalid="ignore", divide="ignore"):
        dxp = -(x1 - x) - (th / (L * np.maximum(G + L, alpha * L)) * (L + R))
    dxp = np.where(L > 0, dxp, 0.0)
    return x - dxp, (L, G, R, th, x1)


def adam_step(params, grad, state, t, eta=1e-3, beta=(0.9, 0.999), eps=1e-8, dtype=np.float64):
    """One Adam update at step t (1-based); state = (mt, vt) arrays (zeros before the first step)."""
    x, g = np.asarray(params, dtype=dtype), np.asarray(grad, dtype=dtype)
    mt, vt = (np.asarray(s, dtype=dtype) for s in state)
    b1, b2 = dtype(beta[0]), dtype(beta[1])
    mt = b1 * mt + (dtype(1) - b1) * g
    vt = b2 * vt + (dtype(1) - b2) * g * g
    c1 = dtype(1.0 - float(beta[0]) ** t)
    c2 = dtype(1.0 - float(beta[1]) ** t)
    step = (dtype(eta) * (mt / c1)) / (np.sqrt(vt / c2) + dtype(eps))
    return (x - step).astype(dtype), (mt.astype(dtype), vt.astype(dtype))


# --------------------------------------------------------------------------------------
# Counter-based RNG: Philox4x32-10 + Box-Muller (the eps stream of the HIP kernels)
# --------------------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al., SC'11; Random123 v1.09).  `ctr`: (..., 4) uint32 array,
    `key`: (k0, k1).  Vectorised over leading dims.  Known-answer vectors in tests."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for r in range(10):
        p0 = _PHILOX_M0 * c[0]
        p1 = _PHILOX_M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def philox_bits(seed: int, estimate_idx: int, d: int, m_lo: int, m_hi: int, pad: bool = False) -> np.ndarray:
    """Raw 32-bit words of the eps stream for global sample columns [m_lo, m_hi):
    element (i, m) uses word i%4 of the Philox block with
        counter = (lo32(q), hi32(q), lo32(estimate_idx), hi32(estimate_idx)),  q = m*ceil(d/4) + i//4
        key     = (lo32(seed), hi32(seed)).
    Returns uint32 array (d, m_hi-m_lo), or all 4*ceil(d/4) rows when `pad` (the last Philox block of a
    column is only partly consumed when d % 4 != 0).  Mirrors advancedvi.jl_amd/csrc/philox.h."""
    d4 = (d + 3) // 4
    m = np.arange(m_lo, m_hi, dtype=np.uint64)
    blk = np.arange(d4, dtype=np.uint64)
    q = m[None, :] * np.uint64(d4) + blk[:, None]          # (d4, M)
    ctr = np.empty(q.shape + (4,), dtype=np.uint32)
    ctr[..., 0] = (q & _MASK32).astype(np.uint32)
    ctr[..., 1] = (q >> np.uint64(32)).astype(np.uint32)
    ctr[..., 2] = np.uint32(estimate_idx & 0xFFFFFFFF)
    ctr[..., 3] = np.uint32((estimate_idx >> 32) & 0xFFFFFFFF)
    out = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))   # (d4, M, 4)
    words = np.transpose(out, (0, 2, 1)).reshape(d4 * 4, m_hi - m_lo)
    return words if pad else words[:d]


def box_muller_from_bits(words: np.ndarray, f64: bool = False) -> np.ndarray:
    """Box-Muller on the (4*ceil(d/4), M) word array produced by `philox_bits(..., pad=True)`; rows 4b, 4b+1 form one
    pair and 4b+2, 4b+3 the other:
        n_{4b}   = r(w0) cos(2 pi u(w1)),  n_{4b+1} = r(w0) sin(2 pi u(w1)),   r(w) = sqrt(-2 ln u(w))
    f32 stream: u(w) = ((w >> 9) + 0.5) * 2^-23   (exact in float32);
    f64 stream: u(w) = (w + 0.5) * 2^-32.
    Evaluated here in float64 (the device evaluates in its compute dtype)."""
    dpad, M = words.shape
    assert dpad % 4 == 0, "pass whole Philox blocks (philox_bits(..., pad=True))"
    w = words.reshape(dpad // 4, 4, M).astype(np.float64)
    if f64:
        u = (w + 0.5) * 2.0 ** -32
    else:
        u = (np.floor(w / 512.0) + 0.5) * 2.0 ** -23
    out = np.empty_like(u)
    for a, b in ((0, 1), (2, 3)):
        r = np.sqrt(-2.0 * np.log(u[:, a]))
        ang = 2.0 * np.pi * u[:, b]
        out[:, a] = r * np.cos(ang)
        out[:, b] = r * np.sin(ang)
    return out.reshape(dpad, M)


def philox_normal(seed: int, estimate_idx: int, d: int, m_lo: int, m_hi: int, f64: bool = False) -> np.ndarray:
    """eps[:, m_lo:m_hi] of the estimate `estimate_idx` (d x (m_hi-m_lo), float64 evaluation)."""
    return box_muller_from_bits(philox_bits(seed, estimate_idx, d, m_lo, m_hi, pad=True), f64=f64)[:d]
