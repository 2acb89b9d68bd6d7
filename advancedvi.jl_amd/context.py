"""Thin object wrapper over the libmivi C ABI (include/mivi.h).  Device memory and the HIP stream
come from torch (plumbing only); every computation is a libmivi kernel."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .families import FULLRANK, MEANFIELD
from . import problems as P

_NP2MIVI = {np.dtype(np.float32): _lib.F32, np.dtype(np.float64): _lib.F64}


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("libmivi needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch


def torch_dtype(np_dtype):
    torch = _torch()
    return torch.float32 if np.dtype(np_dtype) == np.float32 else torch.float64


class MiviContext:
    """One RepGradELBO estimator instance on one GPU (mivi_create ... mivi_destroy)."""

    def __init__(self, dtype, family, d, n_mc, entropy, seed, device=0, m_offset=0, m_total=0, stream=None):
        torch = _torch()
        self.lib = _lib.load()
        self.np_dtype = np.dtype(dtype)
        self.family, self.d, self.n_mc, self.entropy = int(family), int(d), int(n_mc), int(entropy)
        self.device = int(device)
        self.tdtype = torch_dtype(self.np_dtype)
        self.tdevice = torch.device("cuda", self.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.tdevice).cuda_stream
        cfg = _lib.MiviConfig(_NP2MIVI[self.np_dtype], self.family, self.d, self.n_mc, self.entropy, self.device,
                              C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), int(m_offset), int(m_total), C.c_void_p(stream), 0, 0)
        h = C.c_void_p()
        st = self.lib.mivi_create(C.byref(cfg), C.byref(h))
        _lib.check(self.lib, None, st)
        self.h = h
        self.m_total = int(m_total) if m_total else self.n_mc
        self._keep = []      # keep-alive for callbacks / borrowed device arrays
        self.problem = None

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.mivi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        _lib.check(self.lib, self.h, st)

    @property
    def params_len(self):
        return int(self.lib.mivi_params_len(self.h))

    @property
    def partials_len(self):
        return int(self.lib.mivi_partials_len(self.h))

    def synchronize(self):
        self._chk(self.lib.mivi_synchronize(self.h))

    # -- tensors ----------------------------------------------------------------------------------
    def to_device(self, x):
        torch = _torch()
        if isinstance(x, torch.Tensor):
            if x.device != self.tdevice or x.dtype != self.tdtype or not x.is_contiguous():
                x = x.to(device=self.tdevice, dtype=self.tdtype).contiguous()
            return x
        return torch.as_tensor(np.ascontiguousarray(x, dtype=self.np_dtype)).to(self.tdevice)

    def empty(self, n):
        return _torch().empty(int(n), dtype=self.tdtype, device=self.tdevice)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    # -- targets ----------------------------------------------------------------------------------
    def set_problem(self, prob):
        if P.dimension(prob) != self.d:
            raise ValueError("dimension(prob) does not match the variational family")
        dt = self.np_dtype
        if isinstance(prob, P.TransformedProblem):   # inner target first, then the Stacked bijector around it
            self.set_problem(prob.prob)
            self.set_bijector(prob.bijector)
            self.problem = prob
            return
        self.set_bijector(None)
        if not isinstance(prob, (P.LogRegProblem, P.LogRegSubset)):
            self._lr_full = None   # another target kind replaces the resident data set
        if isinstance(prob, P.DiagNormalProblem):
            m = np.ascontiguousarray(prob.mean, dtype=dt)
            s = np.ascontiguousarray(np.broadcast_to(prob.std, m.shape), dtype=dt)
            self._chk(self.lib.mivi_set_target_diag_gauss(self.h, m.ctypes.data, s.ctypes.data))
        elif isinstance(prob, P.DenseNormalProblem):
            m = np.ascontiguousarray(prob.mean, dtype=dt)
            L = np.asfortranarray(prob.L, dtype=dt)
            self._chk(self.lib.mivi_set_target_dense_gauss(self.h, m.ctypes.data, L.ctypes.data))
        elif isinstance(prob, P.LogRegProblem) and getattr(self, "_lr_full", None) is prob:
            self._chk(self.lib.mivi_logreg_select_rows(self.h, None, 0, 1.0))   # already resident: back to all rows
        elif isinstance(prob, P.LogRegProblem):
            self._lr_full = prob
            torch = _torch()
            X = prob.X
            if isinstance(X, torch.Tensor):   # device-resident, column-major n x p expected
                Xd, yd = X, prob.y
                n = X.shape[1] if getattr(prob, "x_is_colmajor_tensor", False) else X.shape[0]
                self._keep += [Xd, yd]
                self._chk(self.lib.mivi_set_target_logreg(self.h, self._p(Xd), self._p(yd), n,
                                                          P.LogRegProblem.VARIANTS[prob.variant], prob.likeadj, 1))
            else:
                Xf = np.asfortranarray(X, dtype=dt)
                y = np.ascontiguousarray(prob.y, dtype=np.uint8)
                self._chk(self.lib.mivi_set_target_logreg(self.h, Xf.ctypes.data, y.ctypes.data, Xf.shape[0],
                                                          P.LogRegProblem.VARIANTS[prob.variant], prob.likeadj, 0))
        elif isinstance(prob, P.LogRegSubset):
            if getattr(self, "_lr_full", None) is not prob.parent:   # upload the full data set once, then only select rows
                self.set_problem(prob.parent)
            self._chk(self.lib.mivi_logreg_select_rows(self.h, prob.batch.ctypes.data, prob.batch.size, prob.likeadj))
        elif isinstance(prob, P.FunnelProblem):
            self._chk(self.lib.mivi_set_target_funnel(self.h, prob.sigma_v))
        elif isinstance(prob, P.FunnelConstrainedProblem):
            self._chk(self.lib.mivi_set_target_funnel_constrained(self.h, prob.sigma_v))
        else:
            self._set_callback(prob)
        self.problem = prob

    def set_bijector(self, bij):
        """mivi_set_bijector_stacked: a P.StackedBijector around whatever target is set (None removes it)."""
        if bij is None:
            self._chk(self.lib.mivi_set_bijector_stacked(self.h, 0, None, None))
            return
        rng = np.ascontiguousarray([[lo, hi] for lo, hi, _ in bij.blocks], dtype=np.int32)
        kinds = np.ascontiguousarray([P.StackedBijector.KINDS[k] for _, _, k in bij.blocks], dtype=np.int32)
        self._chk(self.lib.mivi_set_bijector_stacked(self.h, len(bij.blocks), rng.ctypes.data, kinds.ctypes.data))

    def _set_callback(self, prob):
        has_grad = hasattr(prob, "logdensity_and_gradient") or hasattr(prob, "logdensity_and_gradient_batch")
        if not has_grad and not hasattr(prob, "logdensity"):
            raise TypeError("generic targets must implement logdensity (LogDensityOrder 0: values only -- estimate_objective) or "
                            "logdensity_and_gradient (LogDensityOrder >= 1); see INTEGRATION.md")
        dt = self.np_dtype

        def as_mat(ptr, d, M):
            buf = (C.c_char * (d * M * dt.itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dt).reshape((d, M), order="F")

        def fg(user, Zp, d, M, ellp, Gp):
            try:
                if not has_grad:   # an order-0 problem set directly on a context: values only (objectives.init wraps it in ADgradient)
                    raise TypeError("the target only implements logdensity (LogDensityOrder{0}()): a gradient estimate needs "
                                    "logdensity_and_gradient -- wrap it in ADgradient(\"forwarddiff\", prob), as init(..., AutoMIVI(), ...) does")
                Z = as_mat(Zp, d, M)
                ell = as_mat(ellp, 1, M)
                G = as_mat(Gp, d, M)
                if hasattr(prob, "logdensity_and_gradient_batch"):
                    l, g = prob.logdensity_and_gradient_batch(Z)
                    ell[0, :] = l
                    G[:, :] = g
                else:
                    for m in range(M):
                        l, g = prob.logdensity_and_gradient(Z[:, m])   # column view, like eachcol(samples)
                        ell[0, m] = l
                        G[:, m] = g
                return 0
            except Exception as e:  # no exceptions across the ABI
                self._cb_error = e
                return 1

        def fv(user, Zp, d, M, ellp):
            try:
                Z = as_mat(Zp, d, M)
                ell = as_mat(ellp, 1, M)
                for m in range(M):
                    ell[0, m] = prob.logdensity(Z[:, m])
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        def fh(user, Zp, d, M, ellp, Gp, Hp):   # second-order plugin: ell, G and the SUM of the Hessians over the M columns
            try:
                Z = as_mat(Zp, d, M)
                ell = as_mat(ellp, 1, M)
                G = as_mat(Gp, d, M)
                H = as_mat(Hp, d, d)
                Hacc = np.zeros((d, d), dtype=np.float64)   # (the chunk's sum in f64, rounded once to the context's dtype)
                for m in range(M):
                    l, g, h = prob.logdensity_gradient_and_hessian(Z[:, m])
                    ell[0, m] = l
                    G[:, m] = g
                    Hacc += np.asarray(h, dtype=np.float64)
                H[:, :] = Hacc
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        cfg = _lib.LOGDENSITY_AND_GRADIENT_FN(fg)
        cfv = _lib.LOGDENSITY_FN(fv) if hasattr(prob, "logdensity") else C.cast(None, _lib.LOGDENSITY_FN)
        cfh = (_lib.LOGDENSITY_GRADIENT_AND_HESSIAN_FN(fh) if hasattr(prob, "logdensity_gradient_and_hessian")
               else C.cast(None, _lib.LOGDENSITY_GRADIENT_AND_HESSIAN_FN))
        self._keep += [cfg, cfv, cfh]
        self._cb_error = None
        self._chk(self.lib.mivi_set_target_callback(self.h, cfg, cfv, None))
        self._chk(self.lib.mivi_set_target_hess_callback(self.h, cfh, None))

    def _raise_cb(self, st):
        err = getattr(self, "_cb_error", None)
        if st != 0 and err is not None:
            self._cb_error = None
            raise err
        self._chk(st)

    # -- hot path ---------------------------------------------------------------------------------
    def sample(self, params, idx, want_eps=True):
        p = self.to_device(params)
        Z = self.empty(self.d * self.n_mc)
        eps = self.empty(self.d * self.n_mc) if want_eps else None
        self._chk(self.lib.mivi_sample(self.h, self._p(p), idx, self._p(Z), self._p(eps) if want_eps else None))
        # d x M column-major -> torch (M, d) row-major view transposed
        Zv = Z.view(self.n_mc, self.d).t()
        return (Zv, eps.view(self.n_mc, self.d).t()) if want_eps else Zv

    def estimate_gradient(self, params, idx, value=None, grad=None):
        p = self.to_device(params)
        value = self.empty(1) if value is None else value
        grad = self.empty(self.params_len) if grad is None else grad
        self._raise_cb(self.lib.mivi_estimate_gradient(self.h, self._p(p), idx, self._p(value), self._p(grad)))
        return value, grad

    def estimate_gradient_n(self, params, idx0, count, value, grad):
        self._chk(self.lib.mivi_estimate_gradient_n(self.h, self._p(params), idx0, int(count), self._p(value), self._p(grad)))

    def estimate_gradient_each(self, params, idx0, count, values=None, grads=None, want_grads=True):
        """Every estimate of the batch idx0 .. idx0 + count - 1 at the same parameters (mivi_estimate_gradient_each):
        values T[count], grads (count, params_len) -- row i equals mivi_estimate_gradient(idx0 + i): bitwise on the generic route, to rounding (1e-6 / 2e-6) on the batch engine."""
        p = self.to_device(params)
        values = self.empty(int(count)) if values is None else values
        if grads is None and want_grads:
            grads = self.empty(int(count) * self.params_len)
        self._chk(self.lib.mivi_estimate_gradient_each(self.h, self._p(p), idx0, int(count), self._p(values), self._p(grads) if grads is not None else None))
        return values, (grads.view(int(count), self.params_len) if grads is not None else None)

    def estimate_objective(self, params, idx, n_samples=0, entropy=-1, value=None):
        p = self.to_device(params)
        value = self.empty(1) if value is None else value
        self._raise_cb(self.lib.mivi_estimate_objective(self.h, self._p(p), idx, int(n_samples), int(entropy), self._p(value)))
        return value

    def gauss_expected_grad_hess(self, params, idx, n_samples=0, grad=None, hess=None, second_order=False):
        """(logpi_avg T[1], grad T[d], hess (d, d)) of mivi_gauss_expected_grad_hess (Stein / Price branch) or, with
        `second_order`, of mivi_gauss_expected_grad_hess2 (sample average of the Hessians); `hess[i, j]` is the matrix entry."""
        p = self.to_device(params)
        logpi = self.empty(1)
        grad = self.empty(self.d) if grad is None else grad
        hess = self.empty(self.d * self.d) if hess is None else hess
        fn = self.lib.mivi_gauss_expected_grad_hess2 if second_order else self.lib.mivi_gauss_expected_grad_hess
        self._raise_cb(fn(self.h, self._p(p), idx, int(n_samples), self._p(logpi), self._p(grad), self._p(hess)))
        return logpi, grad, hess.view(self.d, self.d).t()   # column-major d x d

    def estimate_partials(self, params, idx, partials=None):
        p = self.to_device(params)
        partials = self.empty(self.partials_len) if partials is None else partials
        self._raise_cb(self.lib.mivi_estimate_partials(self.h, self._p(p), idx, self._p(partials)))
        return partials

    def finalize(self, params, partials, value=None, grad=None):
        p = self.to_device(params)
        value = self.empty(1) if value is None else value
        grad = self.empty(self.params_len) if grad is None else grad
        self._chk(self.lib.mivi_finalize(self.h, self._p(p), self._p(partials), self._p(value), self._p(grad)))
        return value, grad

    def set_logreg_route(self, route):
        """0 = by problem size, 1 = matrix-core kernels, 2 = VALU kernels (built-in logistic regression, f32)."""
        self._chk(self.lib.mivi_set_logreg_route(self.h, int(route)))

    def set_index_source(self, idx_tensor):
        """idx_tensor: 1-element int64/uint64 device tensor added to every estimate index (None to unset)."""
        self._idx_src = idx_tensor
        self._chk(self.lib.mivi_set_index_source(self.h, self._p(idx_tensor) if idx_tensor is not None else None))

    def profile_kernel(self, which, params, reps):
        """ms per launch of one pipeline stage, hipEvent-timed on the context's stream (mivi_profile_kernel)."""
        ms = C.c_double(0.0)
        self._chk(self.lib.mivi_profile_kernel(self.h, int(which), self._p(params), int(reps), C.byref(ms)))
        return ms.value

    def batch_lanes(self, count):
        """Estimates per launch the batch engine uses for a `count`-estimate call (mivi_batch_lanes)."""
        return int(self.lib.mivi_batch_lanes(self.h, int(count)))

    def batch_takes_engine(self, params):
        """True when batches of estimates at this configuration run on the batch engine (mivi_batch_info what = 0)."""
        return int(self.lib.mivi_batch_info(self.h, self._p(params), 0)) == 1

    def exchange_lost(self):
        """True once a launch-free loop's grid-wide exchange was lost on this context: the loops with a per-step exchange are then not taken
        again, their graph-of-launches equivalents run instead (mivi_batch_info what = 3)."""
        return int(self.lib.mivi_batch_info(self.h, None, 3)) == 1

    def split_products(self):
        """Matrix-pipe products per product block of the engine's split-operand contractions (mivi_batch_info what = 1)."""
        return int(self.lib.mivi_batch_info(self.h, None, 1))

    def plane_bytes(self):
        """Bytes per operand-plane element of the batch engine (mivi_batch_info what = 2)."""
        return int(self.lib.mivi_batch_info(self.h, None, 2))

    def profile_batch(self, params, lanes, reps):
        """Average launch duration (us) of the batch engine's kernels for `lanes` estimates (mivi_profile_batch):
        dict(eps=.., product=.., vjp=.., dense_product=.., stl_product=..) -- dense_product 0 unless the target is the dense Gaussian,
        stl_product 0 unless the estimator is one of the sticking-the-landing ones."""
        us = (C.c_double * 5)()
        self._chk(self.lib.mivi_profile_batch(self.h, self._p(params), int(lanes), int(reps), us))
        return dict(eps=us[0], product=us[1], vjp=us[2], dense_product=us[3], stl_product=us[4])

    # -- sharded finalisation / collective behind the ABI ----------------------------------------------------------
    def slice_len(self, world):
        return int(self.lib.mivi_slice_len(self.h, int(world)))

    def finalize_slice(self, params, slice_sum, rank, world, out=None):
        out = self.empty(self.slice_len(world)) if out is None else out
        self._chk(self.lib.mivi_finalize_slice(self.h, self._p(params), self._p(slice_sum), int(rank), int(world), self._p(out)))
        return out

    def unpack_final(self, packed_final, value=None, grad=None):
        value = self.empty(1) if value is None else value
        grad = self.empty(self.params_len) if grad is None else grad
        self._chk(self.lib.mivi_unpack_final(self.h, self._p(packed_final), self._p(value), self._p(grad)))
        return value, grad

    def comm_unique_id(self):
        buf = (C.c_char * 128)()
        _lib.check(self.lib, self.h, self.lib.mivi_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        self._chk(self.lib.mivi_comm_init(self.h, unique_id, int(rank), int(world)))

    # -- the exchange written for xGMI (kernels_p2p.hip) ------------------------------------------------------------------
    P2P_HANDLE_BYTES = 256
    ROUTES = {"auto": 0, "allreduce": 1, "rsag": 2, "p2p": 3}

    def p2p_export(self, rank, world):
        buf = (C.c_char * self.P2P_HANDLE_BYTES)()
        self._chk(self.lib.mivi_p2p_export(self.h, int(rank), int(world), buf))
        return bytes(buf)

    def p2p_attach(self, handles):
        """handles: the `world` blobs of p2p_export in rank order (bytes or a list of bytes)."""
        blob = b"".join(handles) if isinstance(handles, (list, tuple)) else bytes(handles)
        self._chk(self.lib.mivi_p2p_attach(self.h, blob))

    def p2p_detach(self):
        self._chk(self.lib.mivi_p2p_detach(self.h))

    def p2p_set_pipeline(self, on):
        """False: serial steps; True (default): the persistent exchange kernel beside the compute chain."""
        self._chk(self.lib.mivi_p2p_set_pipeline(self.h, 1 if on else 0))

    def p2p_set_spin_budget(self, polls):
        self._chk(self.lib.mivi_p2p_set_spin_budget(self.h, int(polls)))

    def comm_enable_p2p(self):
        self._chk(self.lib.mivi_comm_enable_p2p(self.h))

    def p2p_selfcheck(self, params, idx=3):
        """mivi_p2p_selfcheck: the peer-to-peer exchange against the RCCL all-reduce on one sharded estimate, collectively; returns
        dict(value_rel, grad_rel_l2, verified).  Only a verified context takes the peer-to-peer route automatically at world > 1."""
        out = (C.c_double * 3)()
        self._chk(self.lib.mivi_p2p_selfcheck(self.h, self._p(params), int(idx), out))
        return dict(value_rel=out[0], grad_rel_l2=out[1], verified=bool(out[2]))

    def comm_set_route(self, route):
        self._chk(self.lib.mivi_comm_set_route(self.h, self.ROUTES.get(route, route)))

    def comm_route(self):
        return {0: "none", 1: "allreduce", 2: "rsag", 3: "p2p"}[int(self.lib.mivi_comm_route(self.h))]

    def p2p_exchange(self, params, partials, value, grad, phases=7):
        """partials = None: direct mode (p2p_partials_direct stored the vector into the owners' staging areas)."""
        self._chk(self.lib.mivi_p2p_exchange(self.h, self._p(params), self._p(partials) if partials is not None else None, self._p(value),
                                             self._p(grad), int(phases)))

    def p2p_stats(self, reset=False):
        """Exchange diagnostics since the last reset (mivi_p2p_stats): waits in us, groups served, bytes per peer and estimate."""
        out = (C.c_double * 6)()
        self._chk(self.lib.mivi_p2p_stats(self.h, out, 1 if reset else 0))
        return dict(wait_handover_us=out[0], wait_pushes_us=out[1], wait_finals_us=out[2], groups=int(out[3]), bytes_per_peer_per_estimate=out[4],
                    slice_elements=int(out[5]))

    def p2p_partials_direct(self, params, idx):
        self._chk(self.lib.mivi_p2p_partials_direct(self.h, self._p(params), int(idx)))

    def estimate_gradient_dist_n(self, params, idx0, count, value, grad):
        self._chk(self.lib.mivi_estimate_gradient_dist_n(self.h, self._p(params), idx0, int(count), self._p(value), self._p(grad)))

    def profile_dist(self, params, reps):
        """us per estimate: dict(partials, exchange, serial, pipelined) (mivi_profile_dist; every rank calls it collectively)."""
        out = (C.c_double * 4)()
        self._chk(self.lib.mivi_profile_dist(self.h, self._p(params), int(reps), out))
        return dict(partials=out[0], exchange=out[1], serial=out[2], pipelined=out[3])

    def estimate_gradient_dist(self, params, idx, value=None, grad=None):
        p = self.to_device(params)
        value = self.empty(1) if value is None else value
        grad = self.empty(self.params_len) if grad is None else grad
        self._raise_cb(self.lib.mivi_estimate_gradient_dist(self.h, self._p(p), idx, self._p(value), self._p(grad)))
        return value, grad

    def fullrank_route(self, n_samples=0):
        """(generation, bf16x3): which full-rank kernels run for n_samples per launch (mivi_fullrank_route)."""
        r = int(self.lib.mivi_fullrank_route(self.h, int(n_samples)))
        return r & 3, bool(r & 16)

    def logreg_kernels(self, n_samples=0):
        """dict(mfma, logits_planes, xtr_planes): which kernels the native logistic-regression target's contractions run (mivi_logreg_kernels)."""
        r = int(self.lib.mivi_logreg_kernels(self.h, int(n_samples)))
        return dict(mfma=bool(r & 1), logits_planes=bool(r & 2), xtr_planes=bool(r & 4))

    # -- next to the hot path ---------------------------------------------------------------------
    def clip_scale(self, params, epsilon):
        self._chk(self.lib.mivi_clip_scale(self.h, self._p(params), float(epsilon)))

    def prox_scale_entropy(self, params, stepsize=0.0, dog_state=None, dog_kind=0):
        self._chk(self.lib.mivi_prox_scale_entropy(self.h, self._p(params), float(stepsize),
                                                   self._p(dog_state) if dog_state is not None else None, int(dog_kind)))

    def descent_update(self, params, grad, eta):
        self._chk(self.lib.mivi_descent_update(self.h, self._p(params), self._p(grad), float(eta)))

    def adam_update(self, params, grad, state, t, eta, beta1=0.9, beta2=0.999, eps=1e-8):
        self._chk(self.lib.mivi_adam_update(self.h, self._p(params), self._p(grad), self._p(state), int(t), eta, beta1, beta2, eps))

    def cocob_update(self, params, grad, state, alpha=100.0):
        self._chk(self.lib.mivi_cocob_update(self.h, self._p(params), self._p(grad), self._p(state), float(alpha)))

    def axpby(self, y, a, x, b):
        self._chk(self.lib.mivi_axpby(self.h, self._p(y), float(a), self._p(x), float(b), y.numel()))

    def dog_state(self):
        torch = _torch()
        return torch.zeros(int(self.lib.mivi_dog_state_bytes(self.h)), dtype=torch.uint8, device=self.tdevice)

    def dog_init(self, params, state, alpha):
        self._chk(self.lib.mivi_dog_init(self.h, self._p(params), self._p(state), float(alpha)))

    def dog_update(self, params, grad, state, kind):
        self._chk(self.lib.mivi_dog_update(self.h, self._p(params), self._p(grad), self._p(state), int(kind)))

    def optimize_loop(self, params, n_steps, idx0, t0, rule=0, op=0, averager=0, eta=0.0, beta=(0.9, 0.999), adam_eps=1e-8,
                      clip_epsilon=0.0, avg_eta=8.0, opt_state=None, avg_params=None, elbo=None):
        """mivi_optimize_loop: n_steps iterations of {estimate, update, operator, averager} on the device."""
        import ctypes as C
        l = _lib.MiviLoop(int(rule), int(op), int(averager), int(n_steps), float(eta), float(beta[0]), float(beta[1]),
                          float(adam_eps), float(clip_epsilon), float(avg_eta),
                          self._p(opt_state) if opt_state is not None else None,
                          self._p(avg_params) if avg_params is not None else None, int(idx0), int(t0),
                          self._p(elbo) if elbo is not None else None)
        self._chk(self.lib.mivi_optimize_loop(self.h, self._p(params), C.byref(l)))

    def optimize_steps(self, params, opt_state, idx0, t0, n_steps, rule, eta, clip_epsilon, elbo=None):
        st = self.lib.mivi_optimize_steps(self.h, self._p(params), self._p(opt_state) if opt_state is not None else None,
                                          idx0, int(t0), int(n_steps), int(rule), float(eta), float(clip_epsilon),
                                          self._p(elbo) if elbo is not None else None)
        self._chk(st)
