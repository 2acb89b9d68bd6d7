"""Target log-density problems: the LogDensityProblems plugin seam of the hot path
(`logdensity`, `logdensity_and_gradient`, `dimension`, `capabilities`; call sites
src/algorithms/repgradelbo.jl:32,50,85 and src/mixedad_logdensity.jl:13-28).

Built-in problems are *descriptors*: their arithmetic lives in libmivi's fused HIP kernels.
Any other object exposing `dimension()` and `logdensity_and_gradient(z) -> (ell, grad)` is a
generic plugin and is evaluated through the host-callback route (batched over the columns of Z,
exactly the `logdensity_and_gradient` contract of src/mixedad_logdensity.jl:28)."""
from __future__ import annotations

import numpy as np


class LogDensityOrder:
    """LogDensityProblems.LogDensityOrder{K}."""

    def __init__(self, k: int):
        self.k = k

    def __lt__(self, other):
        return self.k < other.k

    def __repr__(self):
        return f"LogDensityOrder{{{self.k}}}()"


def dimension(prob) -> int:
    return int(prob.dimension())


def capabilities(prob) -> LogDensityOrder:
    if hasattr(prob, "capabilities"):
        return prob.capabilities()
    return LogDensityOrder(1 if hasattr(prob, "logdensity_and_gradient") else 0)


class ADgradient:
    """LogDensityProblemsAD.ADgradient(kind, prob) (README.md:168-174): an order-0 problem (only `logdensity`) wrapped so that it
    provides `logdensity_and_gradient`.  kind "forwarddiff" = forward-mode dual numbers over numpy (`forwarddiff.py`; BASELINE.json
    configs[0] names ForwardDiff on the CPU).  `logdensity` must be written with numpy ufuncs / the array functions listed there."""

    KINDS = ("forwarddiff",)

    def __init__(self, kind, prob, chunk: int = 64):
        if kind not in self.KINDS:
            raise ValueError(f"ADgradient: unknown backend {kind!r} (available: {self.KINDS})")
        if not hasattr(prob, "logdensity"):
            raise TypeError("ADgradient: the problem must implement logdensity")
        self.kind, self.prob, self.chunk = kind, prob, int(chunk)

    def dimension(self):
        return dimension(self.prob)

    def capabilities(self):
        return LogDensityOrder(1)

    def logdensity(self, x):
        return self.prob.logdensity(x)

    def logdensity_and_gradient(self, x):
        from . import forwarddiff
        return forwarddiff.value_and_gradient(self.prob.logdensity, np.asarray(x, dtype=np.float64), self.chunk)

    def subsample(self, batch):   # AdvancedVI.subsample forwards to the wrapped problem (src/AdvancedVI.jl:313)
        return ADgradient(self.kind, subsample(self.prob, batch), self.chunk)


class DiagNormalProblem:
    """MvNormal(mean, Diagonal(std.^2)) -- test/models/normal.jl:56-75, bench/benchmarks.jl:43-47."""

    def __init__(self, mean, std, order=1):
        self.mean = np.asarray(mean)
        self.std = np.asarray(std)
        self.order = int(order)   # 2: declares logdensity_gradient_and_hessian (constant Hessian -1 / std^2)

    def dimension(self):
        return self.mean.shape[0]

    def capabilities(self):
        return LogDensityOrder(self.order)


class DenseNormalProblem:
    """MvNormal(mean, L L') -- test/models/normal.jl:36-54 (`normal_fullrank`)."""

    def __init__(self, mean, L, order=1):
        self.mean = np.asarray(mean)
        self.L = np.tril(np.asarray(L))
        self.order = int(order)   # 2: declares logdensity_gradient_and_hessian (constant Hessian -(L L')^-1)

    def dimension(self):
        return self.mean.shape[0]

    def capabilities(self):
        return LogDensityOrder(self.order)


class LogRegProblem:
    """Hierarchical logistic regression over theta = [beta; s].
    variant "logsigma_normal": docs/src/tutorials/subsampling.md:26-38
    variant "lognormal_exp_bijector": README.md:42-66 inside the TransformedLogDensityProblem of README.md:91-106.
    X is n x p (features), y in {0,1}; likeadj = n_data / n."""

    VARIANTS = {"logsigma_normal": 0, "lognormal_exp_bijector": 1}

    def __init__(self, X, y, variant="logsigma_normal", likeadj=1.0, order=1):
        self.X = np.asarray(X)
        self.y = np.asarray(y)
        if variant not in self.VARIANTS:
            raise ValueError(f"unknown variant {variant}")
        self.variant = variant
        self.likeadj = float(likeadj)
        self.order = int(order)   # 2: declares logdensity_gradient_and_hessian (the library averages the Hessians: csrc/kernels_hess2.hip)

    def dimension(self):
        return self.X.shape[1] + 1

    def capabilities(self):
        return LogDensityOrder(self.order)

    def n_rows(self):
        X = self.X
        return X.shape[1] if getattr(self, "x_is_colmajor_tensor", False) else X.shape[0]

    def subsample(self, batch):
        """AdvancedVI.subsample(prob, batch): the rows `batch` (0-based) with the likelihood scaled by n_data / n
        (docs/src/tutorials/subsampling.md:37,99-102).  The data stay on the device; see mivi_logreg_select_rows."""
        return LogRegSubset(self, batch)


class LogRegSubset:
    """A minibatch view of a LogRegProblem (what `subsample` returns); never copies X on the host."""

    def __init__(self, parent: LogRegProblem, batch):
        self.parent = parent
        self.batch = np.ascontiguousarray(np.asarray(batch, dtype=np.int64).reshape(-1))
        if self.batch.size == 0:
            raise ValueError("empty batch")
        self.likeadj = parent.likeadj * parent.n_rows() / self.batch.size

    def dimension(self):
        return self.parent.dimension()

    def capabilities(self):
        return self.parent.capabilities()

    def subsample(self, batch):
        return LogRegSubset(self.parent, self.batch[np.asarray(batch, dtype=np.int64)])


def subsample(model, batch):
    """AdvancedVI.subsample(model, batch): models that do not specialise it are returned unchanged
    (src/AdvancedVI.jl:313; the tutorial's warning at docs/src/tutorials/subsampling.md:106-110)."""
    fn = getattr(model, "subsample", None)
    return fn(batch) if fn is not None else model


class FunnelProblem:
    """Neal's funnel on the constrained scale under Stacked([log-bijector, identity]) (SURVEY.md 8d;
    wrapper pattern of README.md:76-82,102-106)."""

    def __init__(self, d, sigma_v=1.5, order=1):
        self.d = int(d)
        self.sigma_v = float(sigma_v)
        self.order = int(order)   # 2: declares logdensity_gradient_and_hessian (an arrow matrix: csrc/kernels_hess2.hip)

    def dimension(self):
        return self.d

    def capabilities(self):
        return LogDensityOrder(self.order)




class FunnelConstrainedProblem:
    """Neal's funnel on the constrained scale theta = [s; x] (s > 0), WITHOUT a bijector (mivi_set_target_funnel_constrained);
    TransformedProblem(FunnelConstrainedProblem(d, sv), StackedBijector([(0, 1, "exp"), (1, d, "identity")])) is FunnelProblem."""

    def __init__(self, d, sigma_v=1.5, order=1):
        self.d = int(d)
        self.sigma_v = float(sigma_v)
        self.order = int(order)   # 2: declares logdensity_gradient_and_hessian (an arrow matrix: csrc/kernels_hess2.hip)

    def dimension(self):
        return self.d

    def capabilities(self):
        return LogDensityOrder(self.order)


class StackedBijector:
    """inverse(Bijectors.Stacked(bijectors, ranges)) restricted to identity / exp blocks (README.md:76-82):
    blocks = [(begin, end, kind)], 0-based half-open ranges, kind in {"identity", "exp"}."""

    KINDS = {"identity": 0, "exp": 1}

    def __init__(self, blocks):
        self.blocks = [(int(lo), int(hi), str(kind)) for lo, hi, kind in blocks]
        for _, _, kind in self.blocks:
            if kind not in self.KINDS:
                raise ValueError(f"unknown bijector kind {kind}")


class TransformedProblem:
    """TransformedLogDensityProblem(prob, binv) of README.md:91-119: logdensity(eta) = logdensity(prob, binv(eta)) +
    logabsdetjac(binv, eta).  `prob` is any problem the context accepts (built-in descriptor or LogDensityProblems plugin)."""

    def __init__(self, prob, bijector: StackedBijector):
        self.prob = prob
        self.bijector = bijector

    def dimension(self):
        return self.prob.dimension()

    def capabilities(self):
        return self.prob.capabilities()


BUILTIN = (DiagNormalProblem, DenseNormalProblem, LogRegProblem, LogRegSubset, FunnelProblem, FunnelConstrainedProblem)
