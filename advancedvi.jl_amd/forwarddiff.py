"""Forward-mode automatic differentiation with dual numbers over numpy arrays: the host-side gradient provider for
ORDER-0 targets (a LogDensityProblem that only declares `logdensity`).

Why it exists.  The reference accepts such a problem and differentiates THROUGH `LogDensityProblems.logdensity` with
the AD backend it was handed (src/algorithms/repgradelbo.jl:50-57, src/AdvancedVI.jl:47-55).  Its README model
(README.md:42-66) and its benchmark target (bench/benchmarks.jl:25-41) both declare `LogDensityOrder{0}`, and
BASELINE.json configs[0] names the backend: "ForwardDiff on CPU".  libmivi's estimator needs the target's gradient
per sample (the closed-form VJP replaces AD of the ESTIMATOR, not of the user's target), so for an order-0 plugin the
host computes `grad logdensity` with this module and hands (ell, G) to the callback seam (`mivi_set_target_callback`,
the batched `logdensity_and_gradient` contract of src/mixedad_logdensity.jl:23-34).  This is plumbing for the
PCIe-bound plugin route -- the device kernels never see it.

`Dual(val, eps)`: `val` an ndarray of shape S, `eps` the partials, shape S + (k,).  numpy ufuncs and a handful of array
functions dispatch on it (`__array_ufunc__` / `__array_function__`), so a `logdensity` written with ordinary numpy calls
works unchanged.  An operation without a rule raises TypeError naming it (never a silent wrong derivative).
"""
from __future__ import annotations

import numpy as np

__all__ = ["Dual", "value_and_gradient", "gradient"]


def _val(x):
    return x.val if isinstance(x, Dual) else np.asarray(x)


def _lift(eps, shape):
    """partials `eps` (shape S' + (k,)) of an operand broadcast to the result's value shape `shape`."""
    return np.broadcast_to(eps, tuple(shape) + eps.shape[-1:]) if eps.shape[:-1] != tuple(shape) else eps


class Dual:
    __slots__ = ("val", "eps")
    __array_priority__ = 1000.0

    def __init__(self, val, eps):
        self.val = np.asarray(val, dtype=np.float64)
        self.eps = np.asarray(eps, dtype=np.float64)
        if self.eps.shape[:-1] != self.val.shape:
            raise ValueError(f"Dual: partials of shape {self.eps.shape} do not extend a value of shape {self.val.shape}")

    # ---- array protocol -----------------------------------------------------------------------------------------
    @property
    def shape(self):
        return self.val.shape

    @property
    def ndim(self):
        return self.val.ndim

    @property
    def size(self):
        return self.val.size

    @property
    def T(self):
        n = self.val.ndim
        return Dual(self.val.T, np.transpose(self.eps, tuple(range(n - 1, -1, -1)) + (n,)))

    def __len__(self):
        return len(self.val)

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        return Dual(self.val[idx], self.eps[idx + (slice(None),)])

    def __iter__(self):
        for i in range(len(self.val)):
            yield self[i]

    def __float__(self):
        return float(self.val)

    def __repr__(self):
        return f"Dual({self.val!r}, partials={self.eps.shape[-1]})"

    def sum(self, axis=None, keepdims=False):
        if axis is None:
            ax = tuple(range(self.val.ndim))
        else:
            ax = tuple(a % self.val.ndim for a in (axis if isinstance(axis, tuple) else (axis,)))
        return Dual(self.val.sum(axis=ax, keepdims=keepdims), self.eps.sum(axis=ax, keepdims=keepdims))

    def mean(self, axis=None):
        s = self.sum(axis)
        return s * (s.val.size / self.val.size)

    def dot(self, other):
        return _matmul(self, other)

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        v = self.val.reshape(shape)
        return Dual(v, self.eps.reshape(v.shape + self.eps.shape[-1:]))

    # ---- arithmetic (all through the ufunc rules) ---------------------------------------------------------------
    def __add__(self, o): return _binary(np.add, self, o)
    def __radd__(self, o): return _binary(np.add, o, self)
    def __sub__(self, o): return _binary(np.subtract, self, o)
    def __rsub__(self, o): return _binary(np.subtract, o, self)
    def __mul__(self, o): return _binary(np.multiply, self, o)
    def __rmul__(self, o): return _binary(np.multiply, o, self)
    def __truediv__(self, o): return _binary(np.true_divide, self, o)
    def __rtruediv__(self, o): return _binary(np.true_divide, o, self)
    def __pow__(self, o): return _binary(np.power, self, o)
    def __rpow__(self, o): return _binary(np.power, o, self)
    def __neg__(self): return Dual(-self.val, -self.eps)
    def __pos__(self): return self
    def __abs__(self): return _unary(np.absolute, self)
    def __matmul__(self, o): return _matmul(self, o)
    def __rmatmul__(self, o): return _matmul(o, self)

    # comparisons act on the values (like ForwardDiff's)
    def __lt__(self, o): return self.val < _val(o)
    def __le__(self, o): return self.val <= _val(o)
    def __gt__(self, o): return self.val > _val(o)
    def __ge__(self, o): return self.val >= _val(o)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs.get("out") is not None:
            raise TypeError(f"forwarddiff: {ufunc.__name__}.{method} has no differentiation rule")
        if ufunc is np.matmul:
            return _matmul(*inputs)
        if ufunc in _UNARY and len(inputs) == 1:
            return _unary(ufunc, inputs[0])
        if ufunc in _BINARY and len(inputs) == 2:
            return _binary(ufunc, *inputs)
        if ufunc in (np.isfinite, np.isnan, np.isinf, np.sign, np.signbit):
            return ufunc(*[_val(x) for x in inputs])
        raise TypeError(f"forwarddiff: numpy.{ufunc.__name__} has no differentiation rule")

    def __array_function__(self, func, types, args, kwargs):
        rule = _FUNCS.get(func)
        if rule is None:
            raise TypeError(f"forwarddiff: numpy.{func.__name__} has no differentiation rule")
        return rule(*args, **kwargs)


# d/dx of the unary ufuncs, as a function of (x, f(x))
_UNARY = {
    np.negative: lambda x, y: -np.ones_like(x),
    np.positive: lambda x, y: np.ones_like(x),
    np.exp: lambda x, y: y,
    np.expm1: lambda x, y: y + 1.0,
    np.log: lambda x, y: 1.0 / x,
    np.log2: lambda x, y: 1.0 / (x * np.log(2.0)),
    np.log10: lambda x, y: 1.0 / (x * np.log(10.0)),
    np.log1p: lambda x, y: 1.0 / (1.0 + x),
    np.sqrt: lambda x, y: 0.5 / y,
    np.square: lambda x, y: 2.0 * x,
    np.reciprocal: lambda x, y: -y * y,
    np.sin: lambda x, y: np.cos(x),
    np.cos: lambda x, y: -np.sin(x),
    np.tan: lambda x, y: 1.0 + y * y,
    np.tanh: lambda x, y: 1.0 - y * y,
    np.sinh: lambda x, y: np.cosh(x),
    np.cosh: lambda x, y: np.sinh(x),
    np.arctan: lambda x, y: 1.0 / (1.0 + x * x),
    np.absolute: lambda x, y: np.sign(x),
}


def _unary(ufunc, a):
    y = ufunc(a.val)
    return Dual(y, _UNARY[ufunc](a.val, y)[..., None] * a.eps)


def _d_power(x, p, y):
    with np.errstate(divide="ignore", invalid="ignore"):
        dx = np.where(p == 0, 0.0, p * np.power(x, p - 1))
        dp = np.where(x > 0, y * np.log(np.where(x > 0, x, 1.0)), 0.0)
    return dx, dp


def _sigmoid(t):
    return 0.5 * (1.0 + np.tanh(0.5 * t))


# partials (d/da, d/db) of the binary ufuncs, as functions of (a, b, f(a, b))
_BINARY = {
    np.add: lambda a, b, y: (1.0, 1.0),
    np.subtract: lambda a, b, y: (1.0, -1.0),
    np.multiply: lambda a, b, y: (b, a),
    np.true_divide: lambda a, b, y: (1.0 / b, -y / b),
    np.power: _d_power,
    np.maximum: lambda a, b, y: ((a >= b) * 1.0, (a < b) * 1.0),
    np.minimum: lambda a, b, y: ((a <= b) * 1.0, (a > b) * 1.0),
    np.logaddexp: lambda a, b, y: (_sigmoid(a - b), _sigmoid(b - a)),
    np.hypot: lambda a, b, y: (a / y, b / y),
}


def _binary(ufunc, a, b):
    av, bv = _val(a), _val(b)
    y = ufunc(av, bv)
    da, db = _BINARY[ufunc](av, bv, y)
    eps = None
    if isinstance(a, Dual):
        eps = np.asarray(da)[..., None] * _lift(a.eps, np.broadcast_shapes(av.shape, y.shape)) if np.ndim(da) else da * a.eps
        eps = _lift(eps, y.shape)
    if isinstance(b, Dual):
        e2 = np.asarray(db)[..., None] * _lift(b.eps, np.broadcast_shapes(bv.shape, y.shape)) if np.ndim(db) else db * b.eps
        e2 = _lift(e2, y.shape)
        eps = e2 if eps is None else eps + e2
    return Dual(y, eps)


def _matmul_const_dual(A, x):
    """A (ndarray, 1-d or 2-d) @ x (Dual, 1-d or 2-d)."""
    A = np.asarray(A, dtype=np.float64)
    v = A @ x.val
    if x.val.ndim == 1:                       # (..., n) @ (n,): eps (n, k)
        e = A @ x.eps
    else:                                     # (..., n) @ (n, p): eps (n, p, k)
        e = np.tensordot(A, x.eps, axes=([-1], [0]))
    return Dual(v, e)


def _matmul_dual_const(x, B):
    B = np.asarray(B, dtype=np.float64)
    v = x.val @ B
    if B.ndim == 1:                           # (..., n) @ (n,): contract eps's second-to-last axis
        e = np.tensordot(x.eps, B, axes=([-2], [0]))
    else:                                     # (..., n) @ (n, p)
        e = np.moveaxis(np.tensordot(x.eps, B, axes=([-2], [0])), -2, -1)
    return Dual(v, e)


def _matmul(a, b):
    if isinstance(a, Dual) and isinstance(b, Dual):
        r1 = _matmul_dual_const(a, b.val)
        r2 = _matmul_const_dual(a.val, b)
        return Dual(r1.val, r1.eps + r2.eps)
    if isinstance(a, Dual):
        return _matmul_dual_const(a, b)
    return _matmul_const_dual(a, b)


def _sum(a, axis=None, keepdims=False, **kw):
    return a.sum(axis=axis, keepdims=keepdims)


def _concatenate(seq, axis=0, **kw):
    seq = list(seq)
    k = next(x.eps.shape[-1] for x in seq if isinstance(x, Dual))
    vals = [np.atleast_1d(_val(x)).astype(np.float64) for x in seq]
    eps = [np.atleast_2d(x.eps) if isinstance(x, Dual) and x.val.ndim == 0 else
           (x.eps if isinstance(x, Dual) else np.zeros(v.shape + (k,))) for x, v in zip(seq, vals)]
    ax = axis % vals[0].ndim
    return Dual(np.concatenate(vals, axis=ax), np.concatenate(eps, axis=ax))


def _where(cond, a, b):
    cond = _val(cond).astype(bool)
    av, bv = _val(a), _val(b)
    y = np.where(cond, av, bv)
    k = next(x.eps.shape[-1] for x in (a, b) if isinstance(x, Dual))
    ea = _lift(a.eps, np.broadcast_shapes(av.shape, y.shape)) if isinstance(a, Dual) else np.zeros(y.shape + (k,))
    eb = _lift(b.eps, np.broadcast_shapes(bv.shape, y.shape)) if isinstance(b, Dual) else np.zeros(y.shape + (k,))
    return Dual(y, np.where(cond[..., None], _lift(ea, y.shape), _lift(eb, y.shape)))


_FUNCS = {
    np.sum: _sum,
    np.mean: lambda a, axis=None, **kw: a.mean(axis),
    np.dot: _matmul,
    np.matmul: _matmul,
    np.inner: _matmul,
    np.concatenate: _concatenate,
    np.hstack: lambda seq, **kw: _concatenate(seq, axis=-1),
    np.where: _where,
    np.shape: lambda a: a.val.shape,
    np.ndim: lambda a: a.val.ndim,
    np.size: lambda a, axis=None: a.val.size if axis is None else a.val.shape[axis],
    np.reshape: lambda a, shape, **kw: a.reshape(shape),
    np.ravel: lambda a, **kw: a.reshape(-1),
    np.transpose: lambda a, axes=None: a.T,
    np.linalg.norm: lambda a, ord=None, axis=None, **kw: np.sqrt((a * a).sum(axis)),
}


def value_and_gradient(f, x, chunk: int = 64):
    """(f(x), grad f(x)) for scalar-valued `f` of a vector: forward mode, `chunk` partials per sweep
    (ForwardDiff's chunking: ceil(len(x) / chunk) evaluations of f)."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    n = x.size
    g = np.empty(n)
    val = None
    for lo in range(0, max(n, 1), chunk):
        hi = min(n, lo + chunk)
        seed = np.zeros((n, hi - lo))
        seed[np.arange(lo, hi), np.arange(hi - lo)] = 1.0
        y = f(Dual(x, seed))
        if not isinstance(y, Dual):      # f does not depend on x (or dropped the partials): derivative 0
            val, g[lo:hi] = float(y), 0.0
            continue
        if y.val.ndim != 0 and y.val.size != 1:
            raise ValueError("forwarddiff.value_and_gradient: f must return a scalar")
        val = float(y.val.reshape(()))
        g[lo:hi] = y.eps.reshape(-1)
    return val, g


def gradient(f, x, chunk: int = 64):
    return value_and_gradient(f, x, chunk)[1]
