"""Oracle (test infrastructure): GP posterior, Omega, and closed-form derivatives.

Restates in numpy what the reference obtains from gpflow 0.5 `GPR` and
TensorFlow autodiff:

* kernels:            run.py:19-23 (SafeMatern32 distance floor), run.py:79-98,
                      tests/create_model.py:28-30 (stock Matern32, non-ARD)
* posterior:          oei.py:86 (`build_predict(X, full_cov=True)`),
                      oei.py:27,49 (`predict_f` at the training inputs -> fmin)
* Omega:              oei.py:79-96 (noise added to the covariance, oei.py:87)
* d<Omega,M>/dX:      oei.py:66-77 (`tf.gradients` with M held constant)
* dOmega/dx:          oei.py:125-155 (one `tf.gradients` per entry of Omega)
* d2<Omega,M>/dx2:    oei.py:119-120 (`tf.hessians` with M held constant)

The derivatives are written in closed form here and are checked against
central finite differences in tests/test_oracle_gp.py.
"""
import numpy as np
import scipy.linalg as sla

KERNEL_RBF = 0
KERNEL_MATERN32 = 1

# distance conventions for Matern kernels
FLOOR_NONE = 0    # r = sqrt(r2)                      (RBF never takes a root)
FLOOR_SAFE = 1    # r = sqrt(max(r2, 1e-40))          run.py:19-23
FLOOR_STOCK = 2   # r = sqrt(r2 + 1e-12)              stock gpflow 0.5 Matern32

SQRT3 = np.sqrt(3.0)


class Kernel:
    """Stationary kernel k(x, x') = variance * kappa(r2), r2 = sum(((x-x')/l)^2)."""

    def __init__(self, kind, variance, lengthscales, dim, floor_mode=None):
        self.kind = int(kind)
        self.variance = float(variance)
        ls = np.atleast_1d(np.asarray(lengthscales, dtype=np.float64))
        if ls.size == 1:
            ls = np.full(dim, ls[0])
        assert ls.size == dim
        self.lengthscales = ls
        self.dim = dim
        if floor_mode is None:
            floor_mode = FLOOR_NONE if kind == KERNEL_RBF else FLOOR_SAFE
        self.floor_mode = int(floor_mode)

    # kappa and its first two derivatives with respect to t = r2
    def kappa(self, t, order=0):
        t = np.asarray(t, dtype=np.float64)
        if self.kind == KERNEL_RBF:
            e = np.exp(-0.5 * t)
            out = [e]
            if order >= 1:
                out.append(-0.5 * e)
            if order >= 2:
                out.append(0.25 * e)
            return out
        if self.kind == KERNEL_MATERN32:
            if self.floor_mode == FLOOR_STOCK:
                r = np.sqrt(t + 1e-12)
                live = np.ones_like(t, dtype=bool)
            elif self.floor_mode == FLOOR_SAFE:
                r = np.sqrt(np.maximum(t, 1e-40))
                live = t > 1e-40
            else:
                r = np.sqrt(t)
                live = t > 0
            e = np.exp(-SQRT3 * r)
            out = [(1.0 + SQRT3 * r) * e]
            if order >= 1:
                # d kappa/dr = -3 r e ; dr/dt = 1/(2r)  =>  -1.5 e   (0 where floored)
                out.append(np.where(live, -1.5 * e, 0.0))
            if order >= 2:
                with np.errstate(divide='ignore', invalid='ignore'):
                    k2 = np.where(live, 0.75 * SQRT3 * e / r, 0.0)
                out.append(k2)
            return out
        raise ValueError('unknown kernel kind')

    def K(self, X, X2=None):
        X = np.asarray(X, dtype=np.float64)
        X2 = X if X2 is None else np.asarray(X2, dtype=np.float64)
        diff = (X[:, None, :] - X2[None, :, :]) / self.lengthscales
        t = np.sum(diff * diff, axis=2)
        return self.variance * self.kappa(t)[0]


class GPModel:
    """GP regression posterior with a fixed Gaussian noise variance.

    Standard GPR formulas as used by gpflow 0.5 `GPR.build_predict`:
    K = k(X,X) + noise*I, L = chol(K), A = L^-1 k(X,X*), v = L^-1 y,
    mean = A^T v, cov = k(X*,X*) - A^T A.
    """

    def __init__(self, X, y, kernel, noise):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64).reshape(-1)
        self.kernel = kernel
        self.noise = float(noise)
        self.n, self.dim = self.X.shape
        K = kernel.K(self.X) + self.noise * np.eye(self.n)
        self.L = np.linalg.cholesky(K)
        self.v = sla.solve_triangular(self.L, self.y, lower=True)
        self.alpha = sla.solve_triangular(self.L.T, self.v, lower=False)
        # oei.py:27  fmin = min of the posterior MEAN at the training inputs
        mean_train = self.predict_f(self.X)[0]
        self.fmin = float(np.min(mean_train))

    def predict_full(self, Xs):
        Xs = np.asarray(Xs, dtype=np.float64).reshape(-1, self.dim)
        Ks = self.kernel.K(self.X, Xs)                      # [n, q]
        A = sla.solve_triangular(self.L, Ks, lower=True)    # [n, q]
        mean = A.T @ self.v
        cov = self.kernel.K(Xs) - A.T @ A
        return mean, cov

    def predict_f(self, Xs):
        mean, cov = self.predict_full(Xs)
        return mean[:, None], np.diag(cov)[:, None].copy()

    def mean_and_var(self, Xs):
        """mean [q], var [q,q] = posterior covariance + noise*I (oei.py:86-87)."""
        mean, cov = self.predict_full(Xs)
        return mean, cov + self.noise * np.eye(cov.shape[0])

    def omega(self, Xs):
        """oei.py:79-96."""
        mean, var = self.mean_and_var(Xs)
        q = mean.shape[0]
        om = np.empty((q + 1, q + 1))
        om[:q, :q] = var + np.outer(mean, mean)
        om[:q, q] = mean
        om[q, :q] = mean
        om[q, q] = 1.0
        return om

    # ---- derivative building blocks -------------------------------------------------
    def _cross(self, Xs):
        """k(X, x_a), its Jacobian J_a = d k(X,x_a)/d x_a [q, n, d] and the
        second-derivative tensor [q, n, d, d]."""
        k = self.kernel
        diff = (Xs[:, None, :] - self.X[None, :, :])            # [q, n, d]  x_a - X_j
        sd = diff / k.lengthscales ** 2                         # (x-x')/l^2
        t = np.sum(diff * diff / k.lengthscales ** 2, axis=2)   # [q, n]
        k0, k1, k2 = k.kappa(t, order=2)
        Ks = k.variance * k0                                    # [q, n]
        J = k.variance * (2.0 * k1)[:, :, None] * sd            # [q, n, d]
        H = k.variance * ((4.0 * k2)[:, :, None, None] * sd[:, :, :, None] * sd[:, :, None, :]
                          + (2.0 * k1)[:, :, None, None] * np.diag(1.0 / k.lengthscales ** 2))
        return Ks, J, H

    def _self(self, Xs):
        """k(x_a,x_b) with first derivative wrt x_a [q,q,d] and the second derivatives
        d2k/dx_a dx_a^T [q,q,d,d] (d2k/dx_a dx_b^T is its negative)."""
        k = self.kernel
        diff = (Xs[:, None, :] - Xs[None, :, :])
        sd = diff / k.lengthscales ** 2
        t = np.sum(diff * diff / k.lengthscales ** 2, axis=2)
        k0, k1, k2 = k.kappa(t, order=2)
        G = k.variance * (2.0 * k1)[:, :, None] * sd
        H = k.variance * ((4.0 * k2)[:, :, None, None] * sd[:, :, :, None] * sd[:, :, None, :]
                          + (2.0 * k1)[:, :, None, None] * np.diag(1.0 / k.lengthscales ** 2))
        return k.variance * k0, G, H

    def inner_and_grad(self, Xs, M):
        """f = <Omega(Xs), M> and df/dXs with M constant (oei.py:66-77)."""
        Xs = np.asarray(Xs, dtype=np.float64).reshape(-1, self.dim)
        q = Xs.shape[0]
        M = np.asarray(M, dtype=np.float64)
        M11, m = M[:q, :q], M[:q, q]
        om = self.omega(Xs)
        f = float(np.sum(om * M))
        Ks, J, _ = self._cross(Xs)                      # Ks [q,n]
        mean = Ks @ self.alpha
        _, G, _ = self._self(Xs)
        # W = K^-1 K_* M11   [n, q]
        Z = sla.solve_triangular(self.L, Ks.T, lower=True) @ M11
        W = sla.solve_triangular(self.L.T, Z, lower=False)
        gmu = 2.0 * (M11 @ mean) + 2.0 * m              # d f / d mean_a
        grad = np.zeros((q, self.dim))
        for a in range(q):
            grad[a] = J[a].T @ (gmu[a] * self.alpha - 2.0 * W[:, a])
            for b in range(q):
                if b != a:
                    grad[a] += 2.0 * M11[a, b] * G[a, b]
        return f, grad

    def domega(self, Xs):
        """dOmega/dx as [q+1, q+1, q*d] (oei.py:125-135)."""
        Xs = np.asarray(Xs, dtype=np.float64).reshape(-1, self.dim)
        q, d = Xs.shape
        Ks, J, _ = self._cross(Xs)
        _, G, _ = self._self(Xs)
        mean = Ks @ self.alpha
        KiKs = sla.cho_solve((self.L, True), Ks.T)      # [n, q]
        out = np.zeros((q + 1, q + 1, q * d))
        for a in range(q):
            dmu = J[a].T @ self.alpha                   # [d]
            dcov = G[a] - (J[a].T @ KiKs).T             # [q, d]: d Sigma_ab / d x_a, b != a
            dcov[a] = -2.0 * (J[a].T @ KiKs[:, a])      # d Sigma_aa / d x_a
            for l in range(d):
                j = a * d + l
                col = dcov[:, l] + dmu[l] * mean        # d Omega11[a, b], b != a
                col[a] = dcov[a, l] + 2.0 * mean[a] * dmu[l]
                out[a, :q, j] = col
                out[:q, a, j] = col
                out[a, q, j] = dmu[l]
                out[q, a, j] = dmu[l]
        return out

    def inner_hessian(self, Xs, M):
        """d2 <Omega(x), M> / dx2 with M constant, [q*d, q*d] (oei.py:119-120)."""
        Xs = np.asarray(Xs, dtype=np.float64).reshape(-1, self.dim)
        q, d = Xs.shape
        M = np.asarray(M, dtype=np.float64)
        M11, m = M[:q, :q], M[:q, q]
        Ks, J, Hc = self._cross(Xs)
        _, _, Hs = self._self(Xs)
        mean = Ks @ self.alpha
        gmu = 2.0 * (M11 @ mean) + 2.0 * m
        dmu = np.einsum('anl,n->al', J, self.alpha)     # [q, d]
        Z = sla.solve_triangular(self.L, Ks.T, lower=True) @ M11
        W = sla.solve_triangular(self.L.T, Z, lower=False)           # [n, q]
        T = sla.solve_triangular(self.L, J.transpose(1, 0, 2).reshape(self.n, q * d), lower=True)
        TT = T.T @ T                                    # J_a^T K^-1 J_b blocks
        H = np.zeros((q * d, q * d))
        for a in range(q):
            sa = slice(a * d, (a + 1) * d)
            for b in range(q):
                sb = slice(b * d, (b + 1) * d)
                blk = 2.0 * M11[a, b] * np.outer(dmu[a], dmu[b]) - 2.0 * M11[a, b] * TT[sa, sb]
                if a != b:
                    blk = blk - 2.0 * M11[a, b] * Hs[a, b]      # d2k/dx_a dx_b^T = -Hs
                H[sa, sb] += blk
            diag = np.einsum('n,nlm->lm', gmu[a] * self.alpha - 2.0 * W[:, a], Hc[a])
            for c in range(q):
                if c != a:
                    diag += 2.0 * M11[a, c] * Hs[a, c]
            H[sa, sa] += diag
        return H
