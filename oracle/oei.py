"""Oracle (test infrastructure): OEI acquisition value / gradient / Hessian orchestration.

Follows methods/oei.py of the reference:

* `acquisition`          oei.py:19-39
* `acquisition_hessian`  oei.py:41-64 (+ oei.py:105-123 for the final chain rule)
* `project`              oei.py:157-199

The SDP is solved with `sdp_ref.sdp_exact` (the reference's formulation taken to machine
precision) unless `first_order=True`, which stops at the reference's eps=1e-5 (sdp.py:51) and is
what the CPU baseline times.
"""
import numpy as np

from . import sdp_ref


class OEIOracle:
    def __init__(self, model):
        self.model = model
        self.dim = model.dim

    # oei.py:157-199
    def project(self, X, rng=np.random):
        m = self.model
        if m.noise > 1e-4:
            return X, np.eye(X.shape[0])
        l = m.kernel.lengthscales
        idx = np.where(np.all(np.abs((X / l)[:, None, :] - m.X / l) < 1e-2, axis=2))[0]
        idx = np.unique(idx)
        idx_ = np.setdiff1d(np.arange(X.shape[0]), idx)
        X_u = X[idx_]
        V = np.eye(X.shape[0])[idx_]
        random_direction = rng.rand(V.shape[0], len(idx))
        if V.shape[0] > 0 and len(idx) > 0:
            V[:, idx] = random_direction / np.linalg.norm(random_direction, axis=0, keepdims=True)
        if X_u.shape[0] > 0:
            _, uidx = np.unique(np.round(X_u / l, decimals=2), return_index=True, axis=0)
            X_u = X_u[uidx]
            V = V[uidx]
        if len(X_u) == len(X):
            X_u = X
            V = np.eye(X.shape[0])
        return X_u, V

    def _sdp(self, omega, first_order):
        if first_order:
            return sdp_ref.sdp_first_order(omega, self.model.fmin, eps=1e-5)
        return sdp_ref.sdp_exact(omega, self.model.fmin)

    # oei.py:19-39
    def acquisition(self, x, first_order=False, rng=np.random):
        X = np.asarray(x, dtype=np.float64).reshape(-1, self.dim)
        X_, V = self.project(X, rng)
        if len(X_) > 0:
            value, M_, _, _ = self._sdp(self.model.omega(X_), first_order)
            _, gradient_ = self.model.inner_and_grad(X_, M_)
            gradient = V.T.dot(gradient_)
        else:
            value = 0
            gradient = rng.rand(X.size)
        return np.array([value]), gradient.flatten()

    # oei.py:41-64
    def acquisition_hessian(self, x, first_order=False, rng=np.random):
        x = np.asarray(x, dtype=np.float64)
        X = x.reshape(-1, self.dim)
        X_, _ = self.project(X, rng)
        if len(X) != len(X_):
            return np.zeros((x.size, x.size))
        _, M, Y, C = self._sdp(self.model.omega(X), first_order)
        domega = self.model.domega(X)
        dM = sdp_ref.solution_derivative(M, Y, C, domega)
        d2f_a = self.model.inner_hessian(X, M)                         # oei.py:120
        d2f_b = np.tensordot(np.transpose(dM), domega, axes=2)         # oei.py:121
        return d2f_a + d2f_b
