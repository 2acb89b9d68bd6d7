"""Variational family containers mirroring src/families/location_scale.jl (AdvancedVI.jl v0.7.0).

Only what the RepGradELBO hot path needs: the `MvLocationScale` container, the two Gaussian
constructors, and `destructure` / restructure.  All arithmetic on samples (rand, entropy, logpdf
inside the estimators) runs in libmivi's HIP kernels."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

MEANFIELD, FULLRANK = 0, 1


@dataclass
class MvLocationScale:
    """MvLocationScale(location, scale, dist) with dist = Normal(0, 1)
    (src/families/location_scale.jl:15-19).  `scale` is a length-d vector (Diagonal) or a d x d
    lower-triangular matrix (LowerTriangular).  Arrays are numpy (host) float32/float64."""

    location: np.ndarray
    scale: np.ndarray

    def __post_init__(self):
        self.location = np.ascontiguousarray(self.location)
        self.scale = np.asarray(self.scale, dtype=self.location.dtype)
        if self.location.dtype not in (np.float32, np.float64):
            raise TypeError("MvLocationScale: eltype must be Float32 or Float64")
        d = self.location.shape[0]
        if self.scale.shape not in ((d,), (d, d)):
            raise ValueError("scale must be a length-d diagonal or a d x d lower-triangular matrix")

    @property
    def family(self) -> int:
        return MEANFIELD if self.scale.ndim == 1 else FULLRANK

    def __len__(self):  # Base.length(q), location_scale.jl:45
        return self.location.shape[0]

    @property
    def eltype(self):  # Base.eltype, location_scale.jl:49
        return self.location.dtype


def MeanFieldGaussian(mu, diag) -> MvLocationScale:
    """MeanFieldGaussian(mu, L::Diagonal): src/families/location_scale.jl:139-141."""
    mu = np.asarray(mu)
    diag = np.asarray(diag)
    if diag.ndim != 1:
        raise TypeError("MeanFieldGaussian expects the diagonal of the scale (Diagonal matrix)")
    return MvLocationScale(mu, diag)


def FullRankGaussian(mu, L) -> MvLocationScale:
    """FullRankGaussian(mu, L::AbstractTriangular): src/families/location_scale.jl:124-128."""
    mu = np.asarray(mu)
    L = np.asarray(L)
    if L.ndim != 2:
        raise TypeError("FullRankGaussian expects a lower-triangular Cholesky factor")
    return MvLocationScale(mu, np.tril(L))


class Restructure:
    """The `re` closure returned by Optimisers.destructure; RestructureMeanField for the
    Diagonal case (src/families/location_scale.jl:28-37)."""

    def __init__(self, d: int, family: int, dtype):
        self.d, self.family, self.dtype = d, family, dtype

    def __call__(self, flat) -> MvLocationScale:
        flat = np.asarray(flat, dtype=self.dtype)
        d = self.d
        if self.family == MEANFIELD:
            if flat.shape[0] != 2 * d:
                raise ValueError("flat parameter vector has the wrong length")
            return MvLocationScale(flat[:d].copy(), flat[d:].copy())
        if flat.shape[0] != d + d * d:
            raise ValueError("flat parameter vector has the wrong length")
        return MvLocationScale(flat[:d].copy(), np.tril(flat[d:].reshape(d, d, order="F")))


def destructure(q: MvLocationScale):
    """Optimisers.destructure(q) -> (flat, re).
    mean-field: [location; diag(scale)], length 2d   (src/families/location_scale.jl:39-43)
    full-rank : [location; vec(scale)] column-major, length d + d^2, strict upper triangle zero."""
    d = len(q)
    if q.family == MEANFIELD:
        flat = np.concatenate([q.location, q.scale])
    else:
        flat = np.concatenate([q.location, np.tril(q.scale).reshape(-1, order="F")])
    return flat.astype(q.eltype, copy=False), Restructure(d, q.family, q.eltype)
