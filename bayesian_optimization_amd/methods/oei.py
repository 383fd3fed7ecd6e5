"""Drop-in for the reference's methods/oei.py: the Optimistic Expected Improvement acquisition
function, its gradient and Hessian, evaluated by the HIP library.

    OEI(options)
    OEI.acquisition(x)          -> (np.array([value]), gradient [q*d])         oei.py:19-39
    OEI.acquisition_hessian(x)  -> [q*d, q*d]                                  oei.py:41-64
    OEI.omega(X)                -> [q+1, q+1]                                  oei.py:98-103
    OEI.domega(x)               -> [q+1, q+1, q*d]                             oei.py:125-135
    OEI.project(X)              -> (X_u, V)                                    oei.py:157-199
    OEI.acquisition_batched(X)  -> (values [B], gradients [B, q*d])   (no reference counterpart)

`project` is host logic (numpy), exactly as in the reference including its use of np.random for
the directions of removed points; everything else is one call into the C ABI.
"""
import logging

import numpy as np
import torch

from .. import engine
from .bo import BO


class OEI(BO):
    """The Optimistic Expected Improvement acquisition function."""

    def __init__(self, options):
        super(OEI, self).__init__(options)

    # ---- oei.py:19-39 -------------------------------------------------------------------------
    def acquisition(self, x):
        x = np.asarray(x, dtype=np.float64)
        X = x.reshape((-1, self.dim))
        X_, V = self.project(X)
        if len(X_) > 0:
            value, grad, status, _ = self.device_model().value_grad(X_[None], want_flags=False)
            self._warn(status)
            value = float(value.item())
            gradient_ = grad[0].cpu().numpy()
            gradient = V.T.dot(gradient_)
        else:
            value = 0
            gradient = np.random.rand(x.size)
        return np.array([value]), gradient.flatten()

    # ---- oei.py:41-64 -------------------------------------------------------------------------
    def acquisition_hessian(self, x):
        x = np.asarray(x, dtype=np.float64)
        X = x.reshape((-1, self.dim))
        X_, _ = self.project(X)
        if len(X) != len(X_):
            return np.zeros((x.size, x.size))
        H, status = self.device_model().hessian(X[None])
        self._warn(status)
        return H[0].cpu().numpy()

    # ---- batched evaluation: B candidate batches at once ---------------------------------------
    def acquisition_batched(self, X):
        """X [B, q, d] -> (values [B], gradients [B, q*d]).  Candidates that `project` would
        shrink (flagged on the GPU) are re-evaluated one by one through `acquisition`."""
        X = np.asarray(X, dtype=np.float64)
        B = X.shape[0]
        value, grad, status, flags = self.device_model().value_grad(X)
        self._warn(status)
        value = value.cpu().numpy()
        grad = grad.cpu().numpy().reshape(B, -1)
        if float(self.likelihood.variance.value) <= 1e-4:
            for b in np.nonzero(flags.cpu().numpy())[0]:
                v, g = self.acquisition(X[b])
                value[b], grad[b] = v[0], g
        return value, grad

    # ---- oei.py:98-103, 125-135 ----------------------------------------------------------------
    def omega(self, X):
        X = np.asarray(X, dtype=np.float64).reshape((-1, self.dim))
        return self.device_model().omega(X[None])[0].cpu().numpy()

    def domega(self, x):
        X = np.asarray(x, dtype=np.float64).reshape((-1, self.dim))
        return self.device_model().domega(X[None])[0].cpu().numpy()

    # ---- oei.py:157-199 ------------------------------------------------------------------------
    def project(self, X):
        """Removes points that duplicate the dataset or each other when the kernel is (nearly)
        noiseless; V maps the gradient of the reduced problem back and gives removed points a
        random unit direction."""
        if float(self.likelihood.variance.value) > 1e-4:
            return X, np.eye(X.shape[0])
        l = self.kern.lengthscales.value
        near = np.all(np.abs((X / l)[:, None, :] - self.X.value / l) < 1e-2, axis=2)
        idx = np.where(near)[0]
        keep = np.setdiff1d(np.arange(X.shape[0]), idx)
        X_u = X[keep]
        V = np.eye(X.shape[0])[keep]
        random_direction = np.random.rand(V.shape[0], len(idx))
        if V.shape[0] > 0 and len(idx) > 0:
            V[:, idx] = random_direction / np.linalg.norm(random_direction, axis=0, keepdims=True)
        if X_u.shape[0] > 0:
            _, first = np.unique(np.round(X_u / l, decimals=2), return_index=True, axis=0)
            X_u = X_u[first]
            V = V[first]
        if len(X_u) == len(X):
            X_u = X
            V = np.eye(X.shape[0])
        return X_u, V

    @staticmethod
    def _warn(status):
        bad = int((status != 0).sum().item())
        if bad:
            logging.getLogger('opt').warning('SDP solution status: %d of %d candidates not solved'
                                             % (bad, status.numel()))
