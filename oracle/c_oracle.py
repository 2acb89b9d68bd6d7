"""ctypes loader for the C leg of the oracle (oracle/mivi_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "libmivi_oracle.so")


NATIVE_PATH = os.path.join(_HERE, "_native", "libmivi_oracle.so")


def load(path=None):
    lib = C.CDLL(path or PATH)
    for pfx, ct in (("mo64_", C.c_double), ("mo32_", C.c_float)):
        f = getattr(lib, pfx + "estimate_gradient")
        f.restype = C.c_double
        f.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_void_p]
        g = getattr(lib, pfx + "fill_eps")
        g.restype = None
        g.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]
        getattr(lib, pfx + "set_threads").argtypes = [C.c_int]
        getattr(lib, pfx + "max_threads").restype = C.c_int
    return lib


def estimate_gradient(lib, dtype, family, d, M, params, eps, t_mean, t_std, ent_kind, work=None, grad=None):
    dt = np.dtype(dtype)
    pfx = "mo64_" if dt == np.float64 else "mo32_"
    params = np.ascontiguousarray(params, dtype=dt)
    eps = np.asfortranarray(eps, dtype=dt)
    t_mean = np.ascontiguousarray(t_mean, dtype=dt)
    t_std = np.ascontiguousarray(t_std, dtype=dt)
    if grad is None:
        grad = np.empty_like(params)
    if work is None:
        work = np.empty(2 * d * M, dtype=dt)
    v = getattr(lib, pfx + "estimate_gradient")(family, d, M, params.ctypes.data, eps.ctypes.data, t_mean.ctypes.data,
                                                  t_std.ctypes.data, ent_kind, grad.ctypes.data, work.ctypes.data)
    return v, grad


def fill_eps(lib, dtype, seed, idx, d, M, m_offset=0, out=None):
    dt = np.dtype(dtype)
    pfx = "mo64_" if dt == np.float64 else "mo32_"
    eps = np.empty((d, M), dtype=dt, order="F") if out is None else out
    getattr(lib, pfx + "fill_eps")(seed, idx, d, M, m_offset, eps.ctypes.data)
    return eps
