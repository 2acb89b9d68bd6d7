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


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
