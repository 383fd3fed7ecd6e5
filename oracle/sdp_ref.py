"""Oracle (test infrastructure): the OEI semidefinite program as the reference poses it.

    value = max_M <Omega, M>   s.t.  C_i - M >= 0 (PSD),  i = 0..k_-1
    dual:   min sum_i <Y_i, C_i>   s.t.  sum_i Y_i = Omega,  Y_i >= 0

Follows methods/sdp.py of the reference:

* `pack` / `unpack`            sdp.py:249-281 (lower triangle, column-major, sqrt(2) off-diagonals)
* `create_C`, `create_scs_data` sdp.py:123-157
* `sdp_first_order`            sdp.py:11-67: the problem handed to SCS and how (value, M, Y) are read
                               back.  SCS (>=2.0.2, not vendored, absent here) is restated from its
                               published algorithm: operator splitting on the homogeneous self-dual
                               embedding (O'Donoghue, Chu, Parikh, Boyd, JOTA 2016) with relaxation
                               1.5, rho_x 1e-3, the primal/dual/gap stopping rule at eps, and the PSD
                               cone projection by symmetric eigendecomposition.  Data equilibration is
                               omitted: every row of A has norm 1 and every column norm sqrt(k_).
* `kkt_matrix`, `solution_derivative`  sdp.py:159-234, 283-303 (dense numpy solve for Pardiso)
* `kkt_polish`                 Newton's method on the same KKT equations (their Jacobian is the
                               matrix of `create_matrix`), used to take a first-order solution to
                               machine precision so that parity is not limited by solver noise.
* `kkt_certificate`            solver-independent optimality check (dual form of tests/test_sdp.py:5-42).
"""
import numpy as np


# ----------------------------------------------------------------------------- packing
def pack(Z, n=None):
    """sdp.py:249-264."""
    Z = np.asarray(Z, dtype=np.float64)
    n = Z.shape[0] if n is None else n
    iu = np.triu_indices(n)
    rows, cols = iu[1], iu[0]               # lower triangle walked column by column
    z = Z[rows, cols] * np.sqrt(2.0)
    z[rows == cols] /= np.sqrt(2.0)
    return z


def unpack(z, n):
    """sdp.py:266-281."""
    iu = np.triu_indices(n)
    rows, cols = iu[1], iu[0]
    Z = np.zeros((n, n))
    Z[rows, cols] = z
    Z = (Z + Z.T) / np.sqrt(2.0)
    Z[np.diag_indices(n)] /= np.sqrt(2.0)
    return Z


def create_C(k_, fmin):
    """sdp.py:143-152: C_0 = 0; C_i has 1/2 at (last, i-1),(i-1, last) and -fmin at (last,last)."""
    fmin = float(np.asarray(fmin).reshape(-1)[0])
    C = [np.zeros((k_, k_))]
    for i in range(1, k_):
        Ci = np.zeros((k_, k_))
        Ci[-1, i - 1] = 0.5
        Ci[i - 1, -1] = 0.5
        Ci[-1, -1] = -fmin
        C.append(Ci)
    return C


def create_scs_data(omega, fmin):
    """sdp.py:123-157.  A is k_ stacked identities, so it is never materialised: A x stacks k_
    copies of x and A^T y sums the k_ blocks of y."""
    k_ = omega.shape[0]
    C = create_C(k_, fmin)
    b = np.concatenate([pack(Ci) for Ci in C])
    c = -pack(omega)
    return {'b': b, 'c': c, 'C': C, 'k_': k_}


# ----------------------------------------------------------------------------- first-order solver
def _proj_psd_blocks(V, k_):
    """Project each packed k_ x k_ block (rows of V) onto the PSD cone."""
    nb = V.shape[0]
    iu = np.triu_indices(k_)
    rows, cols = iu[1], iu[0]
    Z = np.zeros((nb, k_, k_))
    Z[:, rows, cols] = V
    Z = (Z + Z.transpose(0, 2, 1)) / np.sqrt(2.0)
    idx = np.arange(k_)
    Z[:, idx, idx] /= np.sqrt(2.0)
    w, Q = np.linalg.eigh(Z)
    w = np.maximum(w, 0.0)
    Z = (Q * w[:, None, :]) @ Q.transpose(0, 2, 1)
    out = Z[:, rows, cols] * np.sqrt(2.0)
    out[:, rows == cols] /= np.sqrt(2.0)
    return out


def sdp_first_order(omega, fmin, eps=1e-5, max_iters=5000, alpha=1.5, rho_x=1e-3,
                    return_info=False, warm=None):
    """SCS-style solve of the reference's problem; returns (objective, M, Y, C) like sdp.py:67.
    `warm` = (u, v) of an earlier call continues that run (the role of sdp.py:77-121)."""
    omega = np.asarray(omega, dtype=np.float64)
    omega = (omega + omega.T) / 2                      # sdp.py:34
    assert not np.isnan(omega).any()                   # sdp.py:35
    data = create_scs_data(omega, fmin)
    k_, b, c = data['k_'], data['b'], data['c']
    n = k_ * (k_ + 1) // 2
    m = k_ * n
    B = b.reshape(k_, n)

    # One step solves (R + Q) ut = R w with Q = [[0, A^T, c], [-A, 0, b], [-c^T, -b^T, 0]] and
    # R = diag(rho_x I, I, 1).  Because A^T A = k_ I the inner system has a closed form:
    #   [[rho_x I, A^T], [-A, I]] [x; y] = [rx; ry]  ->  x = (rx - A^T ry)/(rho_x + k_), y = ry + A x
    def solve_inner(rx, ry):
        x = (rx - ry.reshape(k_, n).sum(axis=0)) / (rho_x + k_)
        y = ry + np.tile(x, k_)
        return np.concatenate([x, y])

    h = np.concatenate([c, b])
    g = solve_inner(c, b)
    gTh = g @ h

    def project_lin(w):
        p = solve_inner(rho_x * w[:n], w[n:n + m])
        tau = (w[-1] + h @ p) / (1.0 + gTh)
        return np.concatenate([p - g * tau, [tau]])

    u = np.zeros(n + m + 1)
    v = np.zeros(n + m + 1)
    u[-1] = np.sqrt(n + m + 1)
    v[-1] = np.sqrt(n + m + 1)
    if warm is not None:
        u, v = warm[0].copy(), warm[1].copy()
    nb, nc = np.linalg.norm(b), np.linalg.norm(c)
    status = 'Unfinished'
    it = 0
    for it in range(1, max_iters + 1):
        ut = project_lin(u + v)
        relaxed = alpha * ut + (1 - alpha) * u
        w = relaxed - v
        u_new = w.copy()                                     # x block: free cone
        u_new[n:n + m] = _proj_psd_blocks(w[n:n + m].reshape(k_, n), k_).reshape(-1)
        u_new[-1] = max(w[-1], 0.0)
        v = v - relaxed + u_new
        u = u_new
        tau = u[-1]
        if tau > 1e-12:
            x = u[:n] / tau
            y = u[n:n + m] / tau
            s = v[n:n + m] / tau
            pres = np.linalg.norm(np.tile(x, k_) + s - b) / (1 + nb)
            dres = np.linalg.norm(y.reshape(k_, n).sum(axis=0) + c) / (1 + nc)
            pobj, dobj = c @ x, -(b @ y)
            gap = abs(pobj - dobj) / (1 + abs(pobj) + abs(dobj))
            if max(pres, dres, gap) < eps:
                status = 'Solved'
                break
    tau = max(u[-1], 1e-300)
    x = u[:n] / tau
    y = u[n:n + m] / tau
    M = unpack(x, k_)                                      # sdp.py:236-247
    Y = [unpack(y[i * n:(i + 1) * n], k_) for i in range(k_)]
    objective = -(c @ x)                                   # sdp.py:60
    if return_info:
        return objective, M, Y, data['C'], {'status': status, 'iters': it, 'warm': (u, v)}
    return objective, M, Y, data['C']


# ----------------------------------------------------------------------------- KKT system
def _top_factor(Yi):
    """sdp.py:214-217: rank-one factor y_i = sqrt(lambda_max) * v_max."""
    w, Q = np.linalg.eigh(Yi)
    return np.sqrt(max(w[-1], 0.0)) * Q[:, -1]


def get_P_(k):
    """sdp.py:283-295: expands the packed (row-major upper triangle) vector to vec of the
    symmetric matrix; shape [k*k, k(k+1)/2]."""
    iu = np.triu_indices(k)
    P_ = np.zeros((k * k, len(iu[0])))
    for col, (i, j) in enumerate(zip(*iu)):
        P_[i * k + j, col] = 1.0
        P_[j * k + i, col] = 1.0
    return P_


def get_P(k):
    """sdp.py:297-303: selects the upper triangle (row-major) from vec; shape [k(k+1)/2, k*k]."""
    iu = np.triu_indices(k)
    P = np.zeros((len(iu[0]), k * k))
    for row, (i, j) in enumerate(zip(*iu)):
        P[row, i * k + j] = 1.0
    return P


def kkt_matrix(M, ys, C):
    """sdp.py:208-233 without the factorisation: H = [[blockdiag(M-C_i), H2], [H3, 0]]."""
    k = len(ys)
    n_tri = k * (k + 1) // 2
    P, P_ = get_P(k), get_P_(k)
    I = np.eye(k)
    N = k * k + n_tri
    H = np.zeros((N, N))
    for i in range(k):
        y = ys[i].reshape(-1, 1)
        H[i * k:(i + 1) * k, i * k:(i + 1) * k] = M - C[i]
        H[i * k:(i + 1) * k, k * k:] = np.kron(y.T, I) @ P_
        H[k * k:, i * k:(i + 1) * k] = P @ (np.kron(y, I) + np.kron(I, y))
    return H


def solution_derivative(M, Y, C, domegas, return_dY=False):
    """sdp.py:159-206 with a dense solve in place of Pardiso."""
    ys = [_top_factor(Yi) for Yi in Y]
    H = kkt_matrix(M, ys, C)
    domegas = np.asarray(domegas, dtype=np.float64)
    assert domegas.ndim in (2, 3)
    if domegas.ndim == 2:
        domegas = domegas[:, :, None]
    n, k = domegas.shape[0], domegas.shape[-1]
    n_tri = n * (n + 1) // 2
    z = np.zeros((n * n + n_tri, k))
    z[-n_tri:, :] = domegas[np.triu_indices(n)]
    x = np.linalg.solve(H, z)
    G = np.zeros((n, n, k))
    G[np.triu_indices(n)] = x[-n_tri:, :]
    G = G + np.transpose(G, [1, 0, 2])
    G[np.diag_indices(n)] = G[np.diag_indices(n)] / 2
    dM = G
    if return_dY:
        assert k == 1
        dy = x[:n * n, 0].reshape(n, n)
        dY = [np.outer(ys[i], dy[i]) + np.outer(dy[i], ys[i]) for i in range(n)]
        return dM, dY
    return dM


def kkt_polish(omega, fmin, M, Y, iters=20, tol=1e-13):
    """Newton on F(y, M) = [(M - C_i) y_i ; triu(sum y_i y_i^T - Omega)] = 0 from a first-order
    solution.  Returns (objective, M, Y, ys, residual)."""
    omega = (omega + omega.T) / 2
    k_ = omega.shape[0]
    C = create_C(k_, fmin)
    ys = [_top_factor(Yi) for Yi in Y]
    iu = np.triu_indices(k_)
    n_tri = len(iu[0])
    M = M.copy()
    res = np.inf
    for _ in range(iters):
        F = np.concatenate([(M - C[i]) @ ys[i] for i in range(k_)] +
                           [(sum(np.outer(y, y) for y in ys) - omega)[iu]])
        res = np.linalg.norm(F)
        if res < tol * (1 + np.linalg.norm(omega)):
            break
        H = kkt_matrix(M, ys, C)
        step = np.linalg.solve(H, -F)
        # damped step: keep the residual from increasing
        t = 1.0
        for _ls in range(30):
            ys_n = [ys[i] + t * step[i * k_:(i + 1) * k_] for i in range(k_)]
            dM = np.zeros((k_, k_))
            dM[iu] = step[-n_tri:]
            dM = dM + dM.T - np.diag(np.diag(dM))
            M_n = M + t * dM
            F_n = np.concatenate([(M_n - C[i]) @ ys_n[i] for i in range(k_)] +
                                 [(sum(np.outer(y, y) for y in ys_n) - omega)[iu]])
            if np.linalg.norm(F_n) < res:
                break
            t *= 0.5
        ys, M = ys_n, M_n
    Y = [np.outer(y, y) for y in ys]
    return float(np.sum(omega * M)), M, Y, ys, res


def _cert_max(cert):
    return max(v for k, v in cert.items() if k not in ('pobj', 'dobj'))


def sdp_exact(omega, fmin, tol=1e-10, checkpoints=(100, 200, 400, 800, 1600, 3200, 6400, 12800)):
    """Reference formulation solved to machine precision: run the first-order solver and, at
    growing iteration counts, try the KKT-Newton polish; accept the first polished point whose
    KKT certificate is below `tol`.  Returns (objective, M, Y, C)."""
    omega = np.asarray(omega, dtype=np.float64)
    omega = (omega + omega.T) / 2
    warm, done = None, 0
    best = None
    for cp in checkpoints:
        _, M, Y, C, info = sdp_first_order(omega, fmin, eps=1e-9, max_iters=cp - done,
                                           return_info=True, warm=warm)
        warm, done = info['warm'], cp
        try:
            obj, Mp, Yp, ys, res = kkt_polish(omega, fmin, M, Y, iters=30)
        except np.linalg.LinAlgError:      # singular KKT matrix at this iterate: iterate further
            continue
        if np.all(np.isfinite(Mp)):
            cm = _cert_max(kkt_certificate(omega, fmin, Mp, Yp))
            if best is None or cm < best[0]:
                best = (cm, obj, Mp, Yp)
            if cm < tol:
                break
    if best is None or best[0] > 1e-6:
        raise RuntimeError('sdp_exact: no certified solution (best certificate %r)'
                           % (None if best is None else best[0]))
    return best[1], best[2], best[3], C


def kkt_certificate(omega, fmin, M, Y):
    """Solver-independent optimality certificate.  Returns a dict of residuals, each scaled so that
    0 means exact: primal feasibility (most negative eigenvalue of C_i - M), dual feasibility
    (most negative eigenvalue of Y_i; ||sum Y_i - Omega||), complementarity and duality gap."""
    omega = (omega + omega.T) / 2
    k_ = omega.shape[0]
    C = create_C(k_, fmin)
    scale = 1.0 + np.linalg.norm(omega)
    pfeas = min(np.linalg.eigvalsh(Ci - M)[0] for Ci in C)
    dfeas = min(np.linalg.eigvalsh((Yi + Yi.T) / 2)[0] for Yi in Y)
    deq = np.linalg.norm(sum(Y) - omega)
    comp = max(abs(np.sum(Yi * (Ci - M))) for Yi, Ci in zip(Y, C))
    pobj = float(np.sum(omega * M))
    dobj = float(sum(np.sum(Yi * Ci) for Yi, Ci in zip(Y, C)))
    return {'primal_infeas': max(0.0, -pfeas) / (1.0 + np.linalg.norm(M)),
            'dual_infeas': max(0.0, -dfeas) / scale,
            'dual_eq': deq / scale,
            'complementarity': comp / scale,
            'gap': abs(pobj - dobj) / (1.0 + abs(pobj)),
            'pobj': pobj, 'dobj': dobj}
