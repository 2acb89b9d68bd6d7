"""Shared builders for the parity tests: seeded problems/families in both the product's
descriptors (advancedvi_jl_amd) and the oracle's restatement (oracle/)."""
import numpy as np

import advancedvi_jl_amd as avi
from oracle import oracle as O

SEED = 0x38BEF07CF9CC549D  # the reference tests' seed, test/algorithms/klminrepgraddescent.jl:43


def make_family(rng, d, family, dtype=np.float64, mu_scale=1.0):
    mu = (mu_scale * rng.normal(size=d)).astype(dtype)
    if family == avi.MEANFIELD:
        sig = rng.uniform(0.5, 1.5, size=d).astype(dtype)
        return avi.MeanFieldGaussian(mu, sig), O.MvLocationScale(mu.astype(np.float64), sig.astype(np.float64))
    C = np.tril(rng.normal(size=(d, d)) * (0.3 / np.sqrt(d)))
    C[np.diag_indices(d)] = rng.uniform(0.5, 1.5, size=d)
    C = C.astype(dtype)
    return avi.FullRankGaussian(mu, C), O.MvLocationScale(mu.astype(np.float64), C.astype(np.float64))


def make_problem(rng, kind, d, dtype=np.float64):
    if kind == "diag":
        m = rng.normal(size=d).astype(dtype)
        s = rng.uniform(0.5, 2.0, size=d).astype(dtype)
        return avi.DiagNormalProblem(m, s), O.DiagNormalTarget(m, s)
    if kind == "dense":
        m = rng.normal(size=d).astype(dtype)
        L = (np.tril(rng.normal(size=(d, d)) * (0.2 / np.sqrt(d))) + np.eye(d)).astype(dtype)
        return avi.DenseNormalProblem(m, L), O.DenseNormalTarget(m, L)
    if kind in ("logreg0", "logreg1"):
        n = 64
        X = (rng.normal(size=(n, d - 1)) / np.sqrt(d)).astype(dtype)
        y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
        variant = "logsigma_normal" if kind == "logreg0" else "lognormal_exp_bijector"
        adj = 1.7 if kind == "logreg0" else 1.0
        return avi.LogRegProblem(X, y, variant, adj), O.LogRegTarget(X, y, variant, adj)
    if kind == "funnel":
        return avi.FunnelProblem(d, 1.5), O.FunnelStackedTarget(d, 1.5)
    raise ValueError(kind)


class OraclePlugin:
    """A generic LogDensityProblems plugin (host callback route) backed by an oracle target."""

    def __init__(self, tgt):
        self.tgt = tgt
        self.calls = 0

    def dimension(self):
        return self.tgt.dimension()

    def logdensity(self, z):
        return self.tgt.logdensity(np.asarray(z, dtype=np.float64))

    def logdensity_and_gradient(self, z):
        self.calls += 1
        return self.tgt.logdensity_and_gradient(np.asarray(z, dtype=np.float64))


class ReadmeLogReg:
    """The reference README's model, restated line by line (README.md:42-66): theta = [beta; sigma] on the CONSTRAINED scale,
    `logdensity` only, capabilities LogDensityOrder{0}() -- the reference differentiates through it (repgradelbo.jl:50-57)."""

    def __init__(self, X, y):
        self.X, self.y = np.asarray(X, dtype=np.float64), np.asarray(y, dtype=np.float64)
        self.calls = 0

    def logdensity(self, theta):
        self.calls += 1
        X, y = self.X, self.y
        d = X.shape[1]
        beta, sigma = theta[:d], theta[d]
        logprior_beta = -0.5 * np.sum(beta * beta) / (sigma * sigma) - d * np.log(sigma) - 0.5 * d * np.log(2 * np.pi)   # MvNormal(Zeros(d), sigma)
        logprior_sigma = -np.log(sigma) - np.log(3.0) - 0.5 * np.log(2 * np.pi) - np.log(sigma) ** 2 / 18.0              # LogNormal(0, 3)
        logit = X @ beta
        loglike_y = np.sum(y * logit - np.logaddexp(0.0, logit))                                                         # BernoulliLogit
        return loglike_y + logprior_beta + logprior_sigma

    def dimension(self):
        return self.X.shape[1] + 1

    def capabilities(self):
        return avi.LogDensityOrder(0)


def readme_bijector(p):
    """README.md:76-82: Stacked([identity on beta, log-bijector on sigma]), inverted (theta = binv(eta))."""
    return avi.StackedBijector([(0, p, "identity"), (p, p + 1, "exp")])


class BenchDist:
    """bench/benchmarks.jl:25-47: `Dist(MvNormal(fill(5, d), I))` -- it HAS logdensity_and_gradient but declares
    LogDensityOrder{0}(), so the reference differentiates through `logdensity` and never calls it."""

    def __init__(self, n_dims=10):
        self.mu = np.full(n_dims, 5.0)
        self.grad_calls = 0

    def logdensity(self, x):
        r = x - self.mu
        return -0.5 * np.sum(r * r) - 0.5 * self.mu.size * np.log(2 * np.pi)

    def logdensity_and_gradient(self, x):
        self.grad_calls += 1
        return self.logdensity(x), -(np.asarray(x) - self.mu)

    def dimension(self):
        return self.mu.size

    def capabilities(self):
        return avi.LogDensityOrder(0)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def engine_shape(d, M, family=None, dtype=np.float32, kind="diag", n=2):
    """Does a batch of n estimates at this configuration run on the batch engine (csrc/kernels_fullrank_batch.hip; api_batch.hip fb_route)?
    There the products run on two-way f16 operand splits (f32-accurate, 2^-22 per term), so a batch's estimates equal the single calls'
    (exact three-way bf16 split) to rounding; every other route stays bitwise the single calls'."""
    family = avi.FULLRANK if family is None else family
    whole = d % 128 == 0 and M % 128 == 0     # round 6: d and n_mc multiples of 32 (geometry padded to whole tiles) with the diagonal target; the
    return (family == avi.FULLRANK and np.dtype(dtype) == np.float32 and kind in ("diag", "dense") and n >= 2 and d % 32 == 0 and   # dense target (and the STL term: ask ctx.batch_takes_engine) keep whole tiles
            128 <= d <= 2048 and M % 32 == 0 and 128 <= M <= 2048 and (kind == "diag" or whole))


# the stated tolerances of "a batch's estimate equals the single call's" on the batch engine: value relative, gradient relative l2
BATCH_VALUE_RTOL = 1e-6
BATCH_GRAD_RTOL = 2e-6


def assert_batch_matches_single(v, v1, g, g1, engine, what="", ulps=0):
    """v, v1: floats; g, g1: numpy gradients (or None).  engine False: bitwise (values within `ulps` f32 spacings)."""
    v, v1 = float(v), float(v1)
    if engine:
        assert abs(v - v1) <= BATCH_VALUE_RTOL * abs(v1), (what, v, v1)
        if g is not None:
            gb, gs = np.asarray(g, dtype=np.float64), np.asarray(g1, dtype=np.float64)
            assert np.linalg.norm(gb - gs) <= BATCH_GRAD_RTOL * max(1.0, np.linalg.norm(gs)), (what, np.linalg.norm(gb - gs) / max(1.0, np.linalg.norm(gs)))
            assert np.array_equal(gb == 0.0, gs == 0.0) or np.count_nonzero((gb == 0.0) != (gs == 0.0)) < 8, what   # the structural zeros agree
    else:
        assert abs(v - v1) <= ulps * float(np.spacing(np.float32(abs(v1)))), (what, v, v1)
        if g is not None:
            assert np.array_equal(g, g1), what
