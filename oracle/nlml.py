"""CPU restatement (test infrastructure, never shipped) of the objective behind the reference's
hyper-parameter fit: the negative log marginal likelihood of the GP regression model that gpflow's
GPR builds (`bo.py:194-221` optimises it through `self._objective`), and its gradient with respect
to (variance, lengthscale(s), noise).  Checked in the CPU tests against finite differences and
against scikit-learn's `log_marginal_likelihood` for the same kernels."""
import numpy as np

from . import gp


def nlml(X, y, kind, variance, lengthscales, noise, floor_mode=None, want_grad=True):
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    n, d = X.shape
    ls = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64).reshape(-1), (d,)) \
        if np.size(lengthscales) in (1, d) else None
    ard = np.size(lengthscales) == d and d > 1
    kern = gp.Kernel(kind, variance, lengthscales, d, floor_mode)
    K = kern.K(X, X) + noise * np.eye(n)
    L = np.linalg.cholesky(K)
    alpha = np.linalg.solve(L.T, np.linalg.solve(L, y))
    value = 0.5 * y.dot(alpha) + np.log(np.diag(L)).sum() + 0.5 * n * np.log(2 * np.pi)
    if not want_grad:
        return value, None
    Kinv = np.linalg.inv(K)
    Q = Kinv - np.outer(alpha, alpha)
    diff = X[:, None, :] - X[None, :, :]
    tl = diff ** 2 / ls ** 2                       # [n, n, d]
    t = tl.sum(axis=2)
    k0, k1 = kern.kappa(t, order=1)
    g = [0.5 * np.sum(Q * k0)]
    if ard:
        for l in range(d):
            g.append(0.5 * np.sum(Q * (-2.0 * variance * k1 * tl[:, :, l] / ls[l])))
    else:
        g.append(0.5 * np.sum(Q * (-2.0 * variance * k1 * (tl / ls).sum(axis=2))))
    g.append(0.5 * np.trace(Q))
    return value, np.array(g)
