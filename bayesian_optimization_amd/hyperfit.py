"""Maximum-likelihood (or MAP) fit of the GP hyper-parameters with random restarts.

The reference does this step with gpflow before every batch: `BO.optimize_restarts`
(bo.py:194-221) keeps the best of `model_restarts` runs of gpflow's `optimize()` (scipy L-BFGS-B
on the negative log marginal likelihood, plus log-priors when `--priors 1`: Gamma(2, 0.5) on the
lengthscales and Gaussian(1, 2) on the kernel variance, run.py:95-97), each started from a random
draw of the free parameters.  Here the objective and its gradient come from the device
(`oei_model_nlml`: Cholesky, triangular inverse and the trace terms run on the GPU); the outer
optimiser is the same scipy L-BFGS-B.  Parameters are optimised through theta = log(parameter)
(gpflow uses a softplus transform; both keep the parameters positive, restarts are drawn as
theta ~ N(0, 1) around zero as gpflow's `randomize()` does in its free space).
"""
import numpy as np
import scipy.optimize

from . import _lib
from .engine import DeviceGP


def _neg_log_prior(variance, ls, priors):
    """-(log Gamma(ls; shape 2, scale 0.5) + log N(variance; 1, 2)) and derivatives."""
    if not priors:
        return 0.0, 0.0, np.zeros_like(ls)
    shape, scale = 2.0, 0.5
    val = -np.sum((shape - 1.0) * np.log(ls) - ls / scale)
    dls = -((shape - 1.0) / ls - 1.0 / scale)
    val += 0.5 * (variance - 1.0) ** 2 / 2.0
    dvar = (variance - 1.0) / 2.0
    return val, dvar, dls


def objective(theta, X, y, kind, floor_mode, ard, fixed_noise, priors, device=None, gp=None):
    """theta = log([variance, lengthscale(s), noise]) (noise absent when fixed).  Returns
    (value, gradient with respect to theta); +inf when the covariance is not positive definite.
    `gp`: a one-element list holding the DeviceGP to re-factorise in place (created on first use);
    without it a model is built and released for this one evaluation."""
    d = X.shape[1]
    nls = d if ard else 1
    variance = float(np.exp(theta[0]))
    ls = np.exp(theta[1:1 + nls])
    noise = float(fixed_noise) if fixed_noise is not None else float(np.exp(theta[1 + nls]))
    try:
        if gp is None:
            model = DeviceGP(X, y, kind, variance, ls, noise, floor_mode=floor_mode, device=device)
        elif gp[0] is None:
            model = gp[0] = DeviceGP(X, y, kind, variance, ls, noise, floor_mode=floor_mode, device=device)
        else:
            model = gp[0]
            model.refactor(variance, ls, noise)
    except _lib.OeiError:
        return np.inf, np.zeros_like(theta)
    try:
        value, grad = model.nlml(want_grad=True, ard=ard)
    finally:
        if gp is None:
            model.close()
    pv, dvar, dls = _neg_log_prior(variance, ls, priors)
    value += pv
    g = np.empty_like(theta)
    g[0] = (grad[0] + dvar) * variance
    g[1:1 + nls] = (grad[1:1 + nls] + dls) * ls
    if fixed_noise is None:
        g[1 + nls] = grad[1 + nls] * noise
    if not np.isfinite(value):
        return np.inf, np.zeros_like(theta)
    return float(value), g


class BatchedObjective:
    """R trial parameter vectors on the same data in one device pass (`oei_fit_nlml_batched`):
    theta [R, nt] GPU tensor of log-parameters -> (value [R], gradient [R, nt]) GPU tensors."""

    def __init__(self, X, y, kind, floor_mode, ard, fixed_noise, priors, max_batch, device=None):
        import ctypes
        import torch
        if not torch.cuda.is_available():
            raise _lib.OeiError('no GPU visible: the hyper-parameter objective has no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        self.nls = X.shape[1] if ard else 1
        self.fixed_noise = -1.0 if fixed_noise is None else float(fixed_noise)
        self.nt = 1 + self.nls + (1 if fixed_noise is None else 0)
        self.priors = int(bool(priors))
        self.cap = int(max_batch)
        if floor_mode is None:
            floor_mode = 0 if kind == 0 else 1
        handle = ctypes.c_void_p(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.oei_fit_create(ctypes.byref(handle), X.shape[0], X.shape[1],
                                               X.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p),
                                               int(kind), int(floor_mode), self.cap), 'oei_fit_create')
        self._h = handle

    def __call__(self, theta):
        import ctypes
        import torch
        theta = torch.as_tensor(theta, dtype=torch.float64, device=self.device).reshape(-1, self.nt).contiguous()
        R = theta.shape[0]
        value = torch.empty(R, dtype=torch.float64, device=self.device)
        grad = torch.empty((R, self.nt), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.oei_fit_nlml_batched(self._h, R, ctypes.c_void_p(theta.data_ptr()), self.nls,
                                                     self.fixed_noise, self.priors, ctypes.c_void_p(value.data_ptr()),
                                                     ctypes.c_void_p(grad.data_ptr()),
                                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       'oei_fit_nlml_batched')
        return value, grad

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            _lib.lib.oei_fit_destroy(self._h)
            self._h.value = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fit(X, y, kind, variance, lengthscales, noise, floor_mode=None, ard=False, fixed_noise=True,
        priors=False, restarts=1, rng=np.random, device=None, maxiter=1000):
    """Best of `restarts` bounded L-BFGS runs from random starts, plus the current parameters as
    a candidate (bo.py:204-219).  The restarts advance together: one batched objective evaluation
    per round (`oei_fit_nlml_batched`) and the device multistart optimiser
    (`lockstep.minimize_lockstep_device`), so a fit costs the rounds of its slowest restart.
    Returns dict(variance, lengthscales, noise, nlml)."""
    import torch
    from .lockstep import minimize_lockstep_device
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
    d = X.shape[1]
    nls = d if ard else 1
    ls0 = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64).reshape(-1), (nls,)) if ard \
        else np.asarray(lengthscales, dtype=np.float64).reshape(-1)[:1]
    fn = float(noise) if fixed_noise else None
    theta0 = np.log(np.concatenate([[variance], ls0] + ([] if fixed_noise else [[max(noise, 1e-12)]])))
    restarts = int(restarts)
    starts = np.array([rng.randn(theta0.size) for _ in range(restarts)]).reshape(restarts, theta0.size)
    obj = BatchedObjective(X, y, kind, floor_mode, ard, fn, priors, max(restarts, 1), device=device)
    try:
        best_theta, best_val = theta0, float(obj(theta0[None])[0][0])
        if restarts > 0:
            res = minimize_lockstep_device(obj, torch.as_tensor(starts, device=obj.device), -12.0, 8.0,
                                           max_iter=maxiter, max_ls=20)
            vals = torch.where(torch.isfinite(res.fun), res.fun, torch.full_like(res.fun, float('inf')))
            k = int(torch.argmin(vals))
            if float(vals[k]) < best_val:
                best_theta, best_val = res.x[k].cpu().numpy(), float(vals[k])
    finally:
        obj.close()
    out_noise = float(noise) if fixed_noise else float(np.exp(best_theta[1 + nls]))
    return dict(variance=float(np.exp(best_theta[0])), lengthscales=np.exp(best_theta[1:1 + nls]),
                noise=out_noise, nlml=float(best_val))


def fit_sequential(X, y, kind, variance, lengthscales, noise, floor_mode=None, ard=False, fixed_noise=True,
                   priors=False, restarts=1, rng=np.random, device=None, maxiter=1000):
    """The same fit with one scipy L-BFGS-B run per restart, one after the other, on one device
    model that is re-factorised per trial point (the reference's order of work, bo.py:194-221)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
    d = X.shape[1]
    nls = d if ard else 1
    ls0 = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64).reshape(-1), (nls,)) if ard \
        else np.asarray(lengthscales, dtype=np.float64).reshape(-1)[:1]
    fn = float(noise) if fixed_noise else None
    theta0 = np.log(np.concatenate([[variance], ls0] + ([] if fixed_noise else [[max(noise, 1e-12)]])))

    holder = [None]         # one device model for the whole fit, re-factorised per trial point

    def f(th):
        return objective(th, X, y, kind, floor_mode, ard, fn, priors, device, gp=holder)
    try:
        best_theta, best_val = theta0, f(theta0)[0]
        for _ in range(int(restarts)):
            start = rng.randn(theta0.size)
            try:
                res = scipy.optimize.minimize(f, start, jac=True, method='L-BFGS-B',
                                              bounds=[(-12.0, 8.0)] * theta0.size, options={'maxiter': maxiter})
                val, th = float(res.fun), res.x
            except KeyboardInterrupt:
                raise
            except Exception:
                val, th = np.inf, start
            if val < best_val:
                best_theta, best_val = th, val
    finally:
        if holder[0] is not None:
            holder[0].close()
    out_noise = float(noise) if fixed_noise else float(np.exp(best_theta[1 + nls]))
    return dict(variance=float(np.exp(best_theta[0])), lengthscales=np.exp(best_theta[1:1 + nls]),
                noise=out_noise, nlml=float(best_val))
