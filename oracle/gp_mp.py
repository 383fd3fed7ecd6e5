"""Oracle (test infrastructure): the GP part of the path in extended precision.

Same quantities and the same reference lines as `oracle/gp.py` -- posterior (oei.py:86,27,49),
Omega (oei.py:79-96), d<Omega,M>/dX with M constant (oei.py:66-77), dOmega/dx (oei.py:125-155),
d2<Omega,M>/dx2 (oei.py:119-120) -- evaluated with about 75 significant digits, so that it can
arbitrate between the fp64 restatement and the device path where cond(K) eats the digits of both
(RBF kernels at noise 1e-6: cond(K) ~ 1e12).

Arithmetic: every number is a Python integer holding value * 2**FRAC (fixed point, 256 fractional
bits, unbounded integer part), matrices are numpy object arrays of such integers, so sums, dot
products, Cholesky and the triangular solves are exact integer operations up to one truncation
per product.  The transcendental functions of the kernel (exp, sqrt) come from mpmath at a working
precision of FRAC + 64 bits.  fp64 inputs are dyadic rationals and enter exactly; results are
rounded once, on the way out.

Only used to generate the committed fixtures under tests/golden/ (it needs minutes at n = 500).
"""
import math

import mpmath
import numpy as np

from .gp import KERNEL_RBF, KERNEL_MATERN32, FLOOR_NONE, FLOOR_SAFE, FLOOR_STOCK

FRAC = 256
ONE = 1 << FRAC
_PREC = FRAC + 64


def to_fx(a):
    """fp64 array (or scalar) -> fixed point, exactly."""
    a = np.asarray(a, dtype=np.float64)
    out = np.empty(a.shape, dtype=object)
    flat = out.reshape(-1)
    for i, x in enumerate(a.reshape(-1)):
        num, den = float(x).as_integer_ratio()      # den is a power of two <= 2**1074
        flat[i] = (num << FRAC) // den
    return out if a.ndim else out.reshape(())[()]


def to_float(a):
    """fixed point -> fp64, correctly rounded (integer true division)."""
    if isinstance(a, np.ndarray):
        out = np.empty(a.shape, dtype=np.float64)
        of = out.reshape(-1)
        for i, x in enumerate(a.reshape(-1)):
            of[i] = x / ONE
        return out
    return a / ONE


def fmul(a, b):
    return (a * b) >> FRAC


def fdiv(a, b):
    return (a << FRAC) // b


def fdot(a, b):
    return np.dot(a, b) >> FRAC


def fsqrt(x):
    return math.isqrt(int(x) << FRAC)


def _mp(x):
    return mpmath.ldexp(mpmath.mpf(int(x)), -FRAC)


def _fx(x):
    return int(mpmath.floor(mpmath.ldexp(x, FRAC)))


def chol(K):
    """Lower Cholesky factor of a fixed-point symmetric positive definite matrix."""
    n = K.shape[0]
    L = np.zeros((n, n), dtype=object)
    L[:, :] = 0
    for j in range(n):
        s = K[j, j] - (fdot(L[j, :j], L[j, :j]) if j else 0)
        if s <= 0:
            raise ValueError('matrix not positive definite at column %d' % j)
        L[j, j] = fsqrt(s)
        if j + 1 < n:
            col = K[j + 1:, j] - (fdot(L[j + 1:, :j], L[j, :j]) if j else 0)
            L[j + 1:, j] = (col << FRAC) // L[j, j]
    return L


def solve_lower(L, B):
    """L^-1 B, B [n] or [n,m]."""
    n = L.shape[0]
    Y = np.empty(B.shape, dtype=object)
    for i in range(n):
        r = B[i] - (fdot(L[i, :i], Y[:i]) if i else 0)
        Y[i] = (r << FRAC) // L[i, i]
    return Y


def solve_upper_t(L, B):
    """L^-T B."""
    n = L.shape[0]
    Y = np.empty(B.shape, dtype=object)
    for i in range(n - 1, -1, -1):
        r = B[i] - (fdot(L[i + 1:, i], Y[i + 1:]) if i + 1 < n else 0)
        Y[i] = (r << FRAC) // L[i, i]
    return Y


class KernelMP:
    """variance * kappa(t), t = sum(((x-x')/l)^2): kappa and its first two derivatives in t."""

    def __init__(self, kind, variance, lengthscales, dim, floor_mode=None):
        self.kind = int(kind)
        ls = np.atleast_1d(np.asarray(lengthscales, dtype=np.float64))
        if ls.size == 1:
            ls = np.full(dim, ls[0])
        self.lengthscales = ls
        self.dim = dim
        if floor_mode is None:
            floor_mode = FLOOR_NONE if kind == KERNEL_RBF else FLOOR_SAFE
        self.floor_mode = int(floor_mode)
        self.variance_fx = to_fx(float(variance))
        l_fx = to_fx(ls)
        self.inv_l2 = np.array([fdiv(ONE, fmul(l, l)) for l in l_fx], dtype=object)

    def kappa(self, t, order=0):
        """t: object array of fixed-point numbers -> list of arrays kappa, kappa', kappa''."""
        shape = t.shape
        outs = [np.empty(t.size, dtype=object) for _ in range(order + 1)]
        with mpmath.workprec(_PREC):
            sqrt3 = mpmath.sqrt(3)
            for i, ti in enumerate(t.reshape(-1)):
                x = _mp(ti)
                if self.kind == KERNEL_RBF:
                    e = mpmath.exp(-x / 2)
                    vals = (e, -e / 2, e / 4)
                else:
                    assert self.kind == KERNEL_MATERN32
                    if self.floor_mode == FLOOR_STOCK:
                        r, live = mpmath.sqrt(x + mpmath.mpf('1e-12')), True
                    elif self.floor_mode == FLOOR_SAFE:
                        fl = mpmath.mpf('1e-40')
                        r, live = mpmath.sqrt(max(x, fl)), x > fl
                    else:
                        r, live = mpmath.sqrt(x), x > 0
                    e = mpmath.exp(-sqrt3 * r)
                    vals = ((1 + sqrt3 * r) * e,
                            -1.5 * e if live else mpmath.mpf(0),
                            0.75 * sqrt3 * e / r if live else mpmath.mpf(0))
                for o in range(order + 1):
                    outs[o][i] = _fx(vals[o])
        return [o.reshape(shape) for o in outs]

    def sq_dist(self, X, X2):
        """diff [a,b,d], diff/l^2 [a,b,d], t [a,b] for fixed-point coordinate arrays."""
        diff = X[:, None, :] - X2[None, :, :]
        sd = fmul(diff, self.inv_l2)
        t = fmul(diff, sd).sum(axis=2)
        return diff, sd, t


class GPModelMP:
    """Drop-in for `oracle.gp.GPModel` (same attributes and methods, fp64 in and out) whose
    arithmetic runs in fixed point with 256 fractional bits."""

    def __init__(self, X, y, kernel, noise):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64).reshape(-1)
        self.kernel = kernel
        self.noise = float(noise)
        self.n, self.dim = self.X.shape
        self.Xf = to_fx(self.X)
        self.noise_fx = to_fx(self.noise)
        yf = to_fx(self.y)
        _, _, t = kernel.sq_dist(self.Xf, self.Xf)
        K = fmul(kernel.kappa(t)[0], kernel.variance_fx)
        for i in range(self.n):
            K[i, i] += self.noise_fx
        self.Lf = chol(K)
        self.vf = solve_lower(self.Lf, yf)
        self.alphaf = solve_upper_t(self.Lf, self.vf)
        # oei.py:27: minimum posterior mean at the training inputs; k(X,X) alpha = y - noise alpha
        self.fmin = float(np.min(to_float(yf - fmul(self.alphaf, self.noise_fx))))

    # ---- building blocks (fixed point) ------------------------------------------------
    def _cross(self, Xsf, order):
        k = self.kernel
        _, sd, t = k.sq_dist(Xsf, self.Xf)                       # [q,n,d], [q,n]
        ks = k.kappa(t, order)
        var = k.variance_fx
        out = [fmul(ks[0], var)]
        if order >= 1:
            out.append(fmul(fmul(2 * ks[1], var)[:, :, None], sd))            # J [q,n,d]
        if order >= 2:
            d = self.dim
            eye_l = np.zeros((d, d), dtype=object)
            eye_l[:, :] = 0
            for l in range(d):
                eye_l[l, l] = k.inv_l2[l]
            c2 = fmul(4 * ks[2], var)
            c1 = fmul(2 * ks[1], var)
            H = fmul(c2[:, :, None, None], fmul(sd[:, :, :, None], sd[:, :, None, :])) \
                + fmul(c1[:, :, None, None], eye_l)
            out.append(H)
        return out

    def _self(self, Xsf, order):
        k = self.kernel
        _, sd, t = k.sq_dist(Xsf, Xsf)
        ks = k.kappa(t, order)
        var = k.variance_fx
        out = [fmul(ks[0], var)]
        if order >= 1:
            out.append(fmul(fmul(2 * ks[1], var)[:, :, None], sd))
        if order >= 2:
            d = self.dim
            eye_l = np.zeros((d, d), dtype=object)
            eye_l[:, :] = 0
            for l in range(d):
                eye_l[l, l] = k.inv_l2[l]
            H = fmul(fmul(4 * ks[2], var)[:, :, None, None],
                     fmul(sd[:, :, :, None], sd[:, :, None, :])) \
                + fmul(fmul(2 * ks[1], var)[:, :, None, None], eye_l)
            out.append(H)
        return out

    def _omega_fx(self, Xsf):
        q = Xsf.shape[0]
        Ks = self._cross(Xsf, 0)[0]                                # [q,n]
        A = solve_lower(self.Lf, Ks.T.copy())                      # [n,q]
        mean = fdot(Ks, self.alphaf)
        cov = self._self(Xsf, 0)[0] - fdot(A.T, A)
        om = np.empty((q + 1, q + 1), dtype=object)
        om[:q, :q] = cov + fmul(mean[:, None], mean[None, :])
        for a in range(q):
            om[a, a] += self.noise_fx
        om[:q, q] = mean
        om[q, :q] = mean
        om[q, q] = ONE
        return om, mean, Ks

    # ---- the interface of oracle.gp.GPModel -------------------------------------------
    def _xs(self, Xs):
        return to_fx(np.asarray(Xs, dtype=np.float64).reshape(-1, self.dim))

    def omega(self, Xs):
        return to_float(self._omega_fx(self._xs(Xs))[0])

    def mean_and_var(self, Xs):
        om = self.omega(Xs)
        mean = om[:-1, -1]
        return mean, om[:-1, :-1] - np.outer(mean, mean)

    def inner_and_grad(self, Xs, M):
        Xsf = self._xs(Xs)
        q, d = Xsf.shape
        Mf = to_fx(np.asarray(M, dtype=np.float64))
        M11, m = Mf[:q, :q], Mf[:q, q]
        om, mean, _ = self._omega_fx(Xsf)
        f = fmul(om, Mf).sum()
        Ks, J = self._cross(Xsf, 1)
        G = self._self(Xsf, 1)[1]
        KiKs = solve_upper_t(self.Lf, solve_lower(self.Lf, Ks.T.copy()))     # K^-1 Ks^T [n,q]
        W = fdot(KiKs, M11)
        gmu = 2 * (fdot(M11, mean) + m)
        grad = np.empty((q, d), dtype=object)
        for a in range(q):
            coef = fmul(self.alphaf, gmu[a]) - 2 * W[:, a]
            grad[a] = fdot(J[a].T, coef)
            for b in range(q):
                if b != a:
                    grad[a] = grad[a] + 2 * fmul(G[a, b], M11[a, b])
        return to_float(f), to_float(grad)

    def domega(self, Xs):
        Xsf = self._xs(Xs)
        q, d = Xsf.shape
        Ks, J = self._cross(Xsf, 1)
        G = self._self(Xsf, 1)[1]
        mean = fdot(Ks, self.alphaf)
        KiKs = solve_upper_t(self.Lf, solve_lower(self.Lf, Ks.T.copy()))
        out = np.zeros((q + 1, q + 1, q * d), dtype=object)
        out[:, :, :] = 0
        for a in range(q):
            dmu = fdot(J[a].T, self.alphaf)
            dcov = G[a] - fdot(J[a].T, KiKs).T
            dcov[a] = -2 * fdot(J[a].T, KiKs[:, a])
            for l in range(d):
                j = a * d + l
                col = dcov[:, l] + fmul(mean, dmu[l])
                col[a] = dcov[a, l] + 2 * fmul(mean[a], dmu[l])
                out[a, :q, j] = col
                out[:q, a, j] = col
                out[a, q, j] = dmu[l]
                out[q, a, j] = dmu[l]
        return to_float(out)

    def inner_hessian(self, Xs, M):
        Xsf = self._xs(Xs)
        q, d = Xsf.shape
        Mf = to_fx(np.asarray(M, dtype=np.float64))
        M11, m = Mf[:q, :q], Mf[:q, q]
        Ks, J, Hc = self._cross(Xsf, 2)
        Hs = self._self(Xsf, 2)[2]
        mean = fdot(Ks, self.alphaf)
        gmu = 2 * (fdot(M11, mean) + m)
        dmu = np.empty((q, d), dtype=object)
        for a in range(q):
            dmu[a] = fdot(J[a].T, self.alphaf)
        KiKs = solve_upper_t(self.Lf, solve_lower(self.Lf, Ks.T.copy()))
        W = fdot(KiKs, M11)
        Jn = J.transpose(1, 0, 2).reshape(self.n, q * d).copy()
        T = solve_lower(self.Lf, Jn)
        TT = fdot(T.T, T)
        H = np.zeros((q * d, q * d), dtype=object)
        H[:, :] = 0
        for a in range(q):
            sa = slice(a * d, (a + 1) * d)
            for b in range(q):
                sb = slice(b * d, (b + 1) * d)
                blk = fmul(dmu[a][:, None], dmu[b][None, :]) - TT[sa, sb]
                if a != b:
                    blk = blk - Hs[a, b]
                H[sa, sb] = H[sa, sb] + 2 * fmul(blk, M11[a, b])
            coef = fmul(self.alphaf, gmu[a]) - 2 * W[:, a]
            diag = fmul(Hc[a], coef[:, None, None]).sum(axis=0)
            for c in range(q):
                if c != a:
                    diag = diag + 2 * fmul(Hs[a, c], M11[a, c])
            H[sa, sa] = H[sa, sa] + diag
        return to_float(H)
