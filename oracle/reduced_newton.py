"""Oracle (test infrastructure): numpy model of the reduced dual that the HIP solver iterates on.

Not a restatement of a reference file: the reference hands the full cone program to SCS
(sdp.py:51).  This module documents, in executable form, the exact reduction the HIP kernel
uses, so that each stage of the kernel can be checked in isolation; the reference-shaped
solver in `sdp_ref` stays the arbiter of correctness.

With Omega = w0 * [[S + mu mu^T, mu], [mu^T, 1]], S > 0 and delta = mu - fmin, eliminating the
Schur complements of the k_ constraints C_i - M >= 0 leaves

    -value / w0 = max over p in the open simplex (p_i > 0, sum p < 1) of
                  g(p) = tr (S^1/2 D(p) S^1/2)^1/2 - p^T delta,   D(p) = diag(p) - p p^T,

a smooth concave problem in q = k_ - 1 variables.  With D = W^T W, W = (I - gamma s s^T) diag(s),
s = sqrt(p), gamma = 1/(1 + sqrt(p0)), p0 = 1 - sum p, and X = W S W^T = V diag(lam) V^T:

    B = 2 W^-1 X^1/2 W^-T,   N = B^-1 = 1/2 W^T X^-1/2 W,   c = mu + B p / 2,
    M = -[[N, -N c], [-(N c)^T, c^T N c]],   y_0 = sqrt(p0) [c; 1],   y_i = sqrt(p_i) [c - B e_i / 2; 1],
    grad g = diag(B)/4 - B p/2 - delta.
"""
import numpy as np


def split_omega(omega):
    omega = (omega + omega.T) / 2
    w0 = omega[-1, -1]
    mu = omega[:-1, -1] / w0
    S = omega[:-1, :-1] / w0 - np.outer(mu, mu)
    return w0, mu, S


def state(p, mu, S, fmin):
    q = len(p)
    p0 = 1.0 - p.sum()
    s = np.sqrt(p)
    gam = 1.0 / (1.0 + np.sqrt(p0))
    W = (np.eye(q) - gam * np.outer(s, s)) * s[None, :]
    X = W @ S @ W.T
    lam, V = np.linalg.eigh((X + X.T) / 2)
    lam = np.maximum(lam, 1e-300)
    sq = np.sqrt(lam)
    Winv = (np.eye(q) + (gam / np.sqrt(p0)) * np.outer(s, s)) / s[:, None]
    R = Winv @ V                      # B = 2 R diag(sq) R^T
    B = 2.0 * (R * sq) @ R.T
    U = W.T @ V                       # N = 1/2 U diag(1/sq) U^T
    N = 0.5 * (U / sq) @ U.T
    g = sq.sum() - p @ (mu - fmin)
    grad = 0.25 * np.diag(B) - 0.5 * (B @ p) - (mu - fmin)
    return dict(p0=p0, lam=lam, V=V, sq=sq, R=R, U=U, B=B, N=N, g=g, grad=grad, W=W, Winv=Winv)


def hessian(p, st):
    """Hessian of g.  In the eigenbasis of X, dX/dp_i = w_i w_i^T - m m^T with u_i = sq * R[i, :],
    m = sum p_i u_i, w_i = u_i - m, and the second differential of tr X^1/2 is the Loewner kernel
    K_ab = 1 / (sq_a sq_b (sq_a + sq_b)) (Daleckii-Krein):
        H_ij = -1/2 sum_ab K_ab E_i,ab E_j,ab - sum_a u_ia u_ja / sq_a."""
    sq = st['sq']
    Ut = st['R'] * sq
    m = p @ Ut
    Wv = Ut - m[None, :]
    Kab = 1.0 / (sq[:, None] * sq[None, :] * (sq[:, None] + sq[None, :]))
    E = np.einsum('ia,ib->iab', Wv, Wv) - np.outer(m, m)[None]
    return -0.5 * np.einsum('iab,jab,ab->ij', E, E, Kab) - (Ut / sq) @ Ut.T


def initial_p(mu, S, fmin):
    d = mu - fmin
    sig2 = np.diag(S)
    p1 = 0.5 * (1.0 - d / np.sqrt(sig2 + d * d))
    p1 = np.maximum(p1, 1e-300)
    return p1 / (1.0 + p1.sum())


def solve(omega, fmin, tol=1e-7, max_iter=100, verbose=False):
    """Damped Newton with a fraction-to-the-boundary rule and Armijo backtracking on g.  The
    iteration stops after a full step whose relative size max|dp_i|/p_i is below `tol`
    (quadratic convergence then leaves an error near tol^2)."""
    w0, mu, S = split_omega(np.asarray(omega, dtype=np.float64))
    fmin = float(np.asarray(fmin).reshape(-1)[0])
    p = initial_p(mu, S, fmin)
    st = state(p, mu, S, fmin)
    its = 0
    for its in range(1, max_iter + 1):
        H = hessian(p, st)
        step = np.linalg.solve(H, -st['grad'])
        rel = max(np.max(np.abs(step) / p), abs(step.sum()) / st['p0'])
        t = 1.0 if rel < 0.9 else 0.9 / rel          # stay inside the simplex
        dec = st['grad'] @ step
        slack = 1e-14 * (1.0 + abs(st['g']))
        while True:
            pn = p + t * step
            stn = state(pn, mu, S, fmin)
            if stn['g'] >= st['g'] + 1e-4 * t * dec - slack or t < 1e-12:
                break
            t *= 0.5
        if verbose:
            print(its, 't=%.3g' % t, 'g=%.15g' % stn['g'], 'rel=%.3g' % rel)
        p, st = pn, stn
        if t == 1.0 and rel < tol:
            break
    return assemble(p, st, mu, S, fmin, w0) + (its,)


def assemble(p, st, mu, S, fmin, w0=1.0):
    q = len(p)
    B, N = st['B'], st['N']
    c = mu + 0.5 * B @ p
    n = -N @ c
    nu = c @ N @ c
    M = np.zeros((q + 1, q + 1))
    M[:q, :q] = -N
    M[:q, q] = -n
    M[q, :q] = -n
    M[q, q] = -nu
    ys = [np.sqrt(st['p0'] * w0) * np.append(c, 1.0)]
    for i in range(q):
        ys.append(np.sqrt(p[i] * w0) * np.append(c - 0.5 * B[:, i], 1.0))
    Y = [np.outer(y, y) for y in ys]
    value = -w0 * st['g']
    return value, M, Y, p


def sensitivity(p, st, mu, S, fmin, domegas, w0=1.0, return_dp=False):
    """dM for perturbations dOmega[:, :, j] through the reduced optimality conditions
    N S N = D/4 and diag(B) - 2 B p = 4 (mu - fmin):  with dN = U Z U^T,
    Z_ab = 2 (R^T (dD/4 - N dS N) R)_ab / (sq_a + sq_b), dB = -4 R (sq Z sq) R^T, and dp from the
    q x q Newton matrix (the Hessian of g)."""
    q = len(p)
    domegas = np.asarray(domegas, dtype=np.float64)
    if domegas.ndim == 2:
        domegas = domegas[:, :, None]
    K = domegas.shape[2]
    sq, R, U, B, N = st['sq'], st['R'], st['U'], st['B'], st['N']
    H = hessian(p, st)
    c = mu + 0.5 * B @ p
    Phi = 2.0 / (sq[:, None] + sq[None, :])
    dM = np.zeros((q + 1, q + 1, K))
    dps = np.zeros((q, K))
    om11 = S + np.outer(mu, mu)
    for j in range(K):
        dO = 0.5 * (domegas[:, :, j] + domegas[:, :, j].T)
        dw0 = dO[q, q]
        dmu = (dO[:q, q] - mu * dw0) / w0
        dS = (dO[:q, :q] - om11 * dw0) / w0 - np.outer(dmu, mu) - np.outer(mu, dmu)
        # part with dp = 0
        ZS = -0.25 * ((U.T @ dS @ U) / sq[:, None] / sq[None, :]) * Phi
        dBS = -4.0 * (R * sq) @ ZS @ (R * sq).T
        rhs = dmu - 0.25 * (np.diag(dBS) - 2.0 * dBS @ p)
        dp = np.linalg.solve(H, rhs)
        dps[:, j] = dp
        dD = np.diag(dp) - np.outer(dp, p) - np.outer(p, dp)
        Z = ZS + 0.25 * (R.T @ dD @ R) * Phi
        dN = U @ Z @ U.T
        dB = -4.0 * (R * sq) @ Z @ (R * sq).T
        dc = dmu + 0.5 * (dB @ p + B @ dp)
        dn = -dN @ c - N @ dc
        dnu = 2.0 * c @ N @ dc + c @ dN @ c
        dM[:q, :q, j] = -dN
        dM[:q, q, j] = -dn
        dM[q, :q, j] = -dn
        dM[q, q, j] = -dnu
    return (dM, dps) if return_dp else dM
