"""ctypes binding of libmivi.so (include/mivi.h).  There is NO fallback: if the HIP library is
missing or cannot be loaded, importing the hot path raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmivi.so")

MIVI_OK, ERR_BAD_ARG, ERR_NONFINITE, ERR_NONPOSITIVE_SCALE, ERR_HIP, ERR_NO_TARGET, ERR_UNSUPPORTED = range(7)
F32, F64 = 0, 1
MEANFIELD, FULLRANK = 0, 1


class MiviConfig(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("family", C.c_int32), ("d", C.c_int32), ("n_mc", C.c_int32),
        ("entropy", C.c_int32), ("device", C.c_int32), ("seed", C.c_uint64),
        ("m_offset", C.c_int32), ("m_total", C.c_int32), ("stream", C.c_void_p),
        ("own_stream", C.c_int32), ("reserved", C.c_int32),
    ]


LOGDENSITY_AND_GRADIENT_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p)
LOGDENSITY_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p)
LOGDENSITY_GRADIENT_AND_HESSIAN_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)

# name -> (restype, argtypes): every symbol include/mivi.h declares
SIGNATURES = {
    "mivi_create": (C.c_int32, [C.POINTER(MiviConfig), C.POINTER(C.c_void_p)]),
    "mivi_destroy": (C.c_int32, [C.c_void_p]),
    "mivi_last_error": (C.c_char_p, [C.c_void_p]),
    "mivi_version": (C.c_int32, []),
    "mivi_set_stream": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mivi_synchronize": (C.c_int32, [C.c_void_p]),
    "mivi_params_len": (C.c_int64, [C.c_void_p]),
    "mivi_partials_len": (C.c_int64, [C.c_void_p]),
    "mivi_optimize_loop": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mivi_logreg_select_rows": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double]),
    "mivi_prox_scale_entropy": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int32]),
    "mivi_set_target_diag_gauss": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mivi_set_target_dense_gauss": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mivi_set_target_logreg": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int32]),
    "mivi_set_target_funnel": (C.c_int32, [C.c_void_p, C.c_double]),
    "mivi_set_target_callback": (C.c_int32, [C.c_void_p, LOGDENSITY_AND_GRADIENT_FN, LOGDENSITY_FN, C.c_void_p]),
    "mivi_sample": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mivi_estimate_gradient": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mivi_estimate_gradient_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mivi_estimate_gradient_n": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]),
    "mivi_estimate_gradient_each": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]),
    "mivi_profile_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "mivi_batch_lanes": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mivi_p2p_selfcheck": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]),
    "mivi_batch_info": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32]),
    "mivi_estimate_objective": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p]),
    "mivi_estimate_objective_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p]),
    "mivi_gauss_expected_grad_hess": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mivi_gauss_expected_grad_hess_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mivi_gauss_expected_grad_hess2": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mivi_gauss_expected_grad_hess2_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mivi_set_target_hess_callback": (C.c_int32, [C.c_void_p, LOGDENSITY_GRADIENT_AND_HESSIAN_FN, C.c_void_p]),
    "mivi_set_logreg_route": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mivi_estimate_partials": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "mivi_finalize": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mivi_clip_scale": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_double]),
    "mivi_descent_update": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]),
    "mivi_cocob_update": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]),
    "mivi_adam_update": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double]),
    "mivi_axpby": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_int64]),
    "mivi_dog_state_bytes": (C.c_int64, [C.c_void_p]),
    "mivi_dog_init": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]),
    "mivi_dog_update": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "mivi_optimize_steps": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p]),
    "mivi_set_index_source": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mivi_debug_timeline": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mivi_profile_kernel": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    "mivi_fullrank_route": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mivi_logreg_kernels": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mivi_set_target_funnel_constrained": (C.c_int32, [C.c_void_p, C.c_double]),
    "mivi_slice_len": (C.c_int64, [C.c_void_p, C.c_int32]),
    "mivi_finalize_slice": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mivi_unpack_final": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mivi_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "mivi_comm_init": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "mivi_comm_destroy": (C.c_int32, [C.c_void_p]),
    "mivi_estimate_gradient_dist": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mivi_p2p_export": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mivi_p2p_attach": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mivi_p2p_detach": (C.c_int32, [C.c_void_p]),
    "mivi_p2p_geometry": (None, [C.c_int64, C.c_int32, C.POINTER(C.c_int64)]),
    "mivi_comm_enable_p2p": (C.c_int32, [C.c_void_p]),
    "mivi_p2p_debug_words": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mivi_p2p_set_pipeline": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mivi_p2p_set_spin_budget": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mivi_comm_set_route": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mivi_comm_route": (C.c_int32, [C.c_void_p]),
    "mivi_estimate_gradient_dist_n": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]),
    "mivi_p2p_exchange": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "mivi_p2p_partials_direct": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "mivi_p2p_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double), C.c_int32]),
    "mivi_profile_dist": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    "mivi_set_bijector_stacked": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mivi_philox4x32_10": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mivi_eps_bits_host": (None, [C.c_uint64, C.c_uint64, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_uint32)]),
    "mivi_eps_host": (None, [C.c_uint64, C.c_uint64, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
}

_lib = None


def load():
    """Load libmivi.so (once).  Raises RuntimeError when it has not been built: the product path
    never falls back to a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libmivi.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # torch first: this mirror takes its device memory and streams from torch, whose wheel carries its own HIP runtime.  libmivi must bind to
    # THAT copy -- loaded before torch it pulls in the system's libamdhip64, the process then holds two HIP / HSA runtimes and the second one
    # to initialise finds no device ("no usable HIP device 0 (found 0)"; seen with build() and smoke() in one process)
    try:
        import torch  # noqa: F401
    except ImportError:   # (a host without torch: the C ABI alone, the system runtime)
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class MiviError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"libmivi status {status}: {msg}")
        self.status = status


class MiviLoop(C.Structure):
    """mivi_loop_t (include/mivi.h)"""
    _fields_ = [("rule", C.c_int32), ("op", C.c_int32), ("averager", C.c_int32), ("n_steps", C.c_int32),
                ("eta", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("adam_eps", C.c_double),
                ("clip_epsilon", C.c_double), ("avg_eta", C.c_double),
                ("opt_state_dev", C.c_void_p), ("avg_params_dev", C.c_void_p),
                ("estimate_idx0", C.c_uint64), ("t0", C.c_int64), ("elbo_dev", C.c_void_p)]


def check(lib, ctx, status):
    if status != MIVI_OK:
        msg = lib.mivi_last_error(ctx).decode() if ctx else "context creation failed"
        raise MiviError(status, msg)
