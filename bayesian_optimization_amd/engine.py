"""Host-side handle around the C ABI: torch tensors for device memory and streams, nothing else.

`DeviceGP` owns one `oei_model` (the factorised GP of one BO iteration) on one GPU and exposes
the batched entry points with [B, q, d] candidate tensors.  The reference never evaluates more
than one candidate batch at a time (bo.py:142-175); batching B candidates is what the GPU path adds.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import lib, check

KERNEL_RBF = 0
KERNEL_MATERN32 = 1
FLOOR_NONE, FLOOR_SAFE, FLOOR_STOCK = 0, 1, 2

STATUS_TEXT = {0: 'Solved', 1: 'Iteration limit', 2: 'Numerical breakdown'}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f64(x, device):
    t = torch.as_tensor(x, dtype=torch.float64, device=device)
    return t.contiguous()


class DeviceGP:
    """A GP regression model factorised once on the current GPU."""

    def __init__(self, X, y, kernel_kind, variance, lengthscales, noise, floor_mode=None,
                 device=None):
        if not torch.cuda.is_available():
            raise _lib.OeiError('no GPU visible: the OEI path has no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None \
            else torch.device(device)
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        assert X.ndim == 2 and X.shape[0] == y.shape[0]
        self.n, self.dim = X.shape
        ls = np.ascontiguousarray(np.atleast_1d(np.asarray(lengthscales, dtype=np.float64)))
        if floor_mode is None:
            floor_mode = FLOOR_NONE if kernel_kind == KERNEL_RBF else FLOOR_SAFE
        self.kernel_kind, self.floor_mode = int(kernel_kind), int(floor_mode)
        self.variance, self.lengthscales, self.noise = float(variance), ls, float(noise)
        self.X_train, self.y_train = X, y
        handle = ctypes.c_void_p(0)
        with torch.cuda.device(self.device):
            check(lib.oei_model_create(ctypes.byref(handle), self.n, self.dim,
                                       X.ctypes.data_as(ctypes.c_void_p),
                                       y.ctypes.data_as(ctypes.c_void_p),
                                       self.kernel_kind, self.floor_mode, self.variance,
                                       ls.ctypes.data_as(ctypes.c_void_p), int(ls.size),
                                       self.noise), 'oei_model_create')
        self._h = handle
        fmin = ctypes.c_double(0.0)
        check(lib.oei_model_info(self._h, ctypes.byref(fmin), None, None), 'oei_model_info')
        self.fmin = fmin.value

    def refactor(self, variance, lengthscales, noise):
        """New hyper-parameters on the same data, in the handle's buffers (no device allocation):
        one trial point of the hyper-parameter fit (bo.py:194-221).  Raises OeiError when the
        covariance is not positive definite; the handle can then be refactored again."""
        ls = np.ascontiguousarray(np.atleast_1d(np.asarray(lengthscales, dtype=np.float64)))
        with torch.cuda.device(self.device):
            check(lib.oei_model_refactor(self._h, self.kernel_kind, self.floor_mode, float(variance),
                                         ls.ctypes.data_as(ctypes.c_void_p), int(ls.size), float(noise)),
                  'oei_model_refactor')
        self.variance, self.lengthscales, self.noise = float(variance), ls, float(noise)
        fmin = ctypes.c_double(0.0)
        check(lib.oei_model_info(self._h, ctypes.byref(fmin), None, None), 'oei_model_info')
        self.fmin = fmin.value

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            lib.oei_model_destroy(self._h)          # leaves the caller's current device as it was
            self._h = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def factor(self):
        L = np.empty((self.n, self.n))
        check(lib.oei_model_get_factor(self._h, L.ctypes.data_as(ctypes.c_void_p)), 'oei_model_get_factor')
        return L

    def set_profiling(self, on):
        check(lib.oei_set_profiling(self._h, int(bool(on))), 'oei_set_profiling')

    def set_pipelining(self, on):
        """Large value+gradient batches as two halves on two internal streams (default on; an
        integer n >= 2 forces n pieces at any batch size, for measurements)."""
        check(lib.oei_set_pipelining(self._h, int(on)), 'oei_set_pipelining')

    def set_z_free(self, on):
        """Second GEMM on the A rows with M11 applied in its epilogue instead of Z = M11 A from the
        solver kernel (default off)."""
        check(lib.oei_set_z_free(self._h, int(bool(on))), 'oei_set_z_free')

    def stage_times_ms(self):
        """Durations (ms) of [gram, GEMM1, moments+SDP, GEMM2, gradient] of the last profiled call."""
        ms = (ctypes.c_float * 5)()
        check(lib.oei_get_stage_times(self._h, ctypes.cast(ms, ctypes.c_void_p)), 'oei_get_stage_times')
        return [float(x) for x in ms]

    def solver_kernel_ms(self):
        """Duration (ms) of the moments+SDP kernel of the last profiled call, from events attached
        to its own dispatch (no launch gaps)."""
        ms = ctypes.c_float()
        check(lib.oei_get_solver_kernel_time(self._h, ctypes.cast(ctypes.byref(ms), ctypes.c_void_p)),
              'oei_get_solver_kernel_time')
        return float(ms.value)

    # ---- batched device API -----------------------------------------------------------------
    def nlml(self, want_grad=True, ard=False):
        """Negative log marginal likelihood of the model and its gradient with respect to
        (variance, lengthscale(s), noise): the objective of the hyper-parameter fit
        (bo.py:194-221).  Returns (value, grad numpy [2 + nls]) with nls = d if `ard` else 1."""
        import ctypes
        nls = self.dim if ard else 1
        value = ctypes.c_double(0.0)
        grad = (ctypes.c_double * (2 + nls))() if want_grad else None
        with torch.cuda.device(self.device):
            check(lib.oei_model_nlml(self._h, ctypes.byref(value), grad, nls), 'oei_model_nlml')
        return value.value, (np.array(grad[:]) if want_grad else None)

    def predict_f(self, X):
        X = _dev_f64(X, self.device).reshape(-1, self.dim)
        m = X.shape[0]
        mean = torch.empty(m, dtype=torch.float64, device=self.device)
        var = torch.empty(m, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.oei_predict_f(self._h, m, _ptr(X), _ptr(mean), _ptr(var), _stream()), 'oei_predict_f')
        return mean, var

    def _batch(self, X):
        X = _dev_f64(X, self.device)
        if X.dim() == 2:
            X = X.unsqueeze(0)
        assert X.dim() == 3 and X.shape[2] == self.dim, 'X must be [B, q, d]'
        return X.contiguous()

    def omega(self, X):
        X = self._batch(X)
        B, q, _ = X.shape
        om = torch.empty((B, q + 1, q + 1), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.oei_omega_batched(self._h, B, q, _ptr(X), _ptr(om), _stream()), 'oei_omega_batched')
        return om

    def value_grad(self, X, want_flags=True, out=None, no_alloc=False):
        """[B,q,d] -> value [B], grad [B,q,d], status int32 [B], flags int32 [B].
        `out`: (value, grad, status, flags) tensors to write into instead of new ones.
        `no_alloc`: use the entry point that works in the library's fixed scratch array (small
        batches only); with `out` given, the call then allocates no device memory at all."""
        X = self._batch(X)
        B, q, d = X.shape
        if out is not None:
            value, grad, status, flags = out
        else:
            value = torch.empty(B, dtype=torch.float64, device=self.device)
            grad = torch.empty((B, q, d), dtype=torch.float64, device=self.device)
            status = torch.empty(B, dtype=torch.int32, device=self.device)
            flags = torch.empty(B, dtype=torch.int32, device=self.device) if want_flags else None
        fn = lib.oei_value_grad_batched_noalloc if no_alloc else lib.oei_value_grad_batched
        with torch.cuda.device(self.device):
            check(fn(self._h, B, q, _ptr(X), _ptr(value), _ptr(grad), _ptr(status), _ptr(flags), _stream()),
                  'oei_value_grad_batched_noalloc' if no_alloc else 'oei_value_grad_batched')
        return value, grad, status, flags

    def value_grad_warm(self, X, p_init=None, want_iters=False):
        """value_grad started from the solver state `p_init` [B,q] that a previous call returned
        for a nearby X (None: the closed-form start).  Returns (value, grad, status, flags,
        state [B,q], iters or None); feed `state` back as the next call's `p_init`.  The role of
        the reference's SDP warm start (sdp.py:77-121) with the history kept by the caller."""
        X = self._batch(X)
        B, q, d = X.shape
        value = torch.empty(B, dtype=torch.float64, device=self.device)
        grad = torch.empty((B, q, d), dtype=torch.float64, device=self.device)
        status = torch.empty(B, dtype=torch.int32, device=self.device)
        flags = torch.empty(B, dtype=torch.int32, device=self.device)
        p_out = torch.empty((B, q), dtype=torch.float64, device=self.device)
        iters = torch.empty(B, dtype=torch.int32, device=self.device) if want_iters else None
        if p_init is not None:
            p_init = _dev_f64(p_init, self.device).reshape(B, q).contiguous()
        with torch.cuda.device(self.device):
            check(lib.oei_value_grad_warm_batched(self._h, B, q, _ptr(X), _ptr(value), _ptr(grad),
                                                  _ptr(status), _ptr(flags), _ptr(p_init), _ptr(p_out),
                                                  _ptr(iters), _stream()), 'oei_value_grad_warm_batched')
        return value, grad, status, flags, p_out, iters

    def hessian(self, X):
        X = self._batch(X)
        B, q, d = X.shape
        H = torch.empty((B, q * d, q * d), dtype=torch.float64, device=self.device)
        status = torch.empty(B, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.oei_hessian_batched(self._h, B, q, _ptr(X), _ptr(H), _ptr(status), _stream()),
                  'oei_hessian_batched')
        return H, status

    def project_flags(self, X):
        """[B,q,d] -> int32 [B]: 1 where the reference's `project` (oei.py:181-192) would remove a
        point of the candidate (duplicate of a training input, or two equal points)."""
        X = self._batch(X)
        B, q, _ = X.shape
        flags = torch.empty(B, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.oei_project_flags(self._h, B, q, _ptr(X), _ptr(flags), _stream()), 'oei_project_flags')
        return flags

    def domega(self, X):
        X = self._batch(X)
        B, q, d = X.shape
        out = torch.empty((B, q + 1, q + 1, q * d), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.oei_domega_batched(self._h, B, q, _ptr(X), _ptr(out), _stream()), 'oei_domega_batched')
        return out


# ---- model-free entry points ------------------------------------------------------------------
def sdp_batched(omega, fmin, want_Y=True, device=None):
    """omega [B,k_,k_], fmin [B] or scalar -> value [B], M [B,k_,k_], Y [B,k_,k_,k_] or None,
    status [B], iters [B] (device tensors)."""
    if not torch.cuda.is_available():
        raise _lib.OeiError('no GPU visible: the OEI path has no CPU fallback')
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    om = _dev_f64(omega, device)
    if om.dim() == 2:
        om = om.unsqueeze(0)
    B, k_, _ = om.shape
    fm = _dev_f64(fmin, device).reshape(-1)
    if fm.numel() == 1 and B > 1:
        fm = fm.expand(B).contiguous()
    assert fm.numel() == B
    value = torch.empty(B, dtype=torch.float64, device=device)
    M = torch.empty((B, k_, k_), dtype=torch.float64, device=device)
    Y = torch.empty((B, k_, k_, k_), dtype=torch.float64, device=device) if want_Y else None
    status = torch.empty(B, dtype=torch.int32, device=device)
    iters = torch.empty(B, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        check(lib.oei_sdp_batched(B, k_, _ptr(om), _ptr(fm), _ptr(value), _ptr(M), _ptr(Y),
                                  _ptr(status), _ptr(iters), _stream()), 'oei_sdp_batched')
    return value, M, Y, status, iters


def solution_derivative_batched(M, Y, fmin, domega, want_dY=False, device=None):
    """M [B,k_,k_], Y [B,k_,k_,k_], fmin [B], domega [B,k_,k_,K] -> dM [B,k_,k_,K] (, dY)."""
    if not torch.cuda.is_available():
        raise _lib.OeiError('no GPU visible: the OEI path has no CPU fallback')
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    M = _dev_f64(M, device)
    Y = _dev_f64(Y, device)
    dom = _dev_f64(domega, device)
    if M.dim() == 2:
        M, Y, dom = M.unsqueeze(0), Y.unsqueeze(0), dom.unsqueeze(0)
    B, k_, _ = M.shape
    K = dom.shape[3]
    fm = _dev_f64(fmin, device).reshape(-1)
    if fm.numel() == 1 and B > 1:
        fm = fm.expand(B).contiguous()
    dM = torch.empty((B, k_, k_, K), dtype=torch.float64, device=device)
    dY = torch.empty((B, k_, k_, k_), dtype=torch.float64, device=device) if want_dY else None
    status = torch.empty(B, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        check(lib.oei_solution_derivative_batched(B, k_, K, _ptr(M), _ptr(Y), _ptr(fm), _ptr(dom),
                                                  _ptr(dM), _ptr(dY), _ptr(status), _stream()),
              'oei_solution_derivative_batched')
    return dM, dY, status


def argmin(value, index_base=0):
    """Device arg-min with ties to the lowest index -> (best_value [1], best_index int64 [1])."""
    value = value.contiguous()
    bv = torch.empty(1, dtype=torch.float64, device=value.device)
    bi = torch.empty(1, dtype=torch.int64, device=value.device)
    with torch.cuda.device(value.device):
        check(lib.oei_argmin(int(value.numel()), _ptr(value), int(index_base), _ptr(bv), _ptr(bi),
                             _stream()), 'oei_argmin')
    return bv, bi


def best_record(value, X, index_base=0):
    """Device arg-min (ties to the lowest index, NaN never wins) returning the winner's record
    [value, index bits, X row ...] as one float64 tensor of 2 + X[0].numel() entries."""
    value = value.contiguous()
    X = X.contiguous()
    B = int(value.numel())
    xlen = int(X.numel() // B) if B else 0
    rec = torch.empty(2 + xlen, dtype=torch.float64, device=value.device)
    with torch.cuda.device(value.device):
        check(lib.oei_best_record(B, _ptr(value), _ptr(X), xlen, int(index_base), _ptr(rec), _stream()),
              'oei_best_record')
    return rec


def select_record(records):
    """Winner of the gathered per-rank records [nrec, reclen] (device tensor): smallest value, NaN as
    +inf, ties to the smallest index; one launch.  Returns the winning record [reclen]."""
    records = records.contiguous()
    nrec, reclen = int(records.shape[0]), int(records.shape[1])
    out = torch.empty(reclen, dtype=torch.float64, device=records.device)
    with torch.cuda.device(records.device):
        check(lib.oei_select_record(nrec, reclen, _ptr(records), _ptr(out), _stream()), 'oei_select_record')
    return out
