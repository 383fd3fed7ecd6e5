"""Drop-in for the reference's methods/sdp.py, served by the HIP library.

Same call signatures and return conventions:

    sdp(omega, fmin, warm_start=True) -> (objective, M, Y, C)            sdp.py:11-67
    solution_derivative(M, Y, C, domegas, return_dY=False) -> dM[, dY]   sdp.py:159-206
    reset_warm_starting()                                                sdp.py:69-75
    pack(Z, n), unpack(z, n)                                             sdp.py:249-281

numpy float64 arrays in and out; inputs are never retained or mutated.  The SDP is solved on the
GPU to machine precision by a Newton iteration on its reduced dual (csrc/oei_sdp.cuh), so the
cross-call warm start that the reference needs for SCS (sdp.py:77-121) has nothing to speed up:
`warm_start` is accepted and ignored, `reset_warm_starting()` clears an (unused) history so that
callers such as bo.py:146 keep working.  As in the reference a failed solve never raises: a warning
goes to the 'opt' logger (sdp.py:53-56); NaN in omega raises AssertionError (sdp.py:35).  A
positive semidefinite but singular omega is solved on omega + eps I (status 'Iteration limit' in
the warning: the result is that of a neighbouring problem); a matrix with a negative eigenvalue
gives NaN.
"""
import collections
import logging

import numpy as np

from .. import engine

OUTPUT_LEVEL = 0

past_omegas = collections.deque(maxlen=20)
past_solutions = collections.deque(maxlen=20)


def reset_warm_starting():
    """Clears the list of previous solutions (kept for API compatibility, sdp.py:69-75)."""
    global past_omegas, past_solutions
    past_omegas = collections.deque(maxlen=20)
    past_solutions = collections.deque(maxlen=20)


def create_C(k_, fmin):
    """The constraint matrices of sdp.py:143-152."""
    fmin = float(np.asarray(fmin).reshape(-1)[0])
    C = [np.zeros((k_, k_))]
    for i in range(1, k_):
        Ci = np.zeros((k_, k_))
        Ci[-1, i - 1] = 1 / 2
        Ci[i - 1, -1] = 1 / 2
        Ci[-1, -1] = -fmin
        C.append(Ci)
    return C


def sdp(omega, fmin, warm_start=True):
    omega = np.asarray(omega, dtype=np.float64)
    omega = (omega + omega.T) / 2
    assert not np.isnan(omega).any()
    k_ = omega.shape[0]
    fmin_s = float(np.asarray(fmin).reshape(-1)[0])
    value, M, Y, status, _ = engine.sdp_batched(omega[None], np.array([fmin_s]))
    st = int(status.item())
    if st != 0:
        # The device solver needs a positive definite covariance block; SCS in the reference takes
        # any symmetric matrix (sdp.py:34-67).  A positive semidefinite but singular Omega has a
        # finite optimum that depends continuously on Omega: solve Omega + eps I, eps from
        # 1e-12 tr(Omega)/k_ upwards, and report the first that converges, with a warning.
        tr = float(np.trace(omega)) / k_
        for eps in (1e-12, 1e-10, 1e-8):
            if not (tr > 0):
                break
            v2, M2, Y2, s2, _ = engine.sdp_batched((omega + eps * tr * np.eye(k_))[None], np.array([fmin_s]))
            if int(s2.item()) == 0:
                value, M, Y, st = v2, M2, Y2, 1
                logging.getLogger('opt').warning('SDP: Omega is singular to working precision, solved on the '
                                                 'regularised matrix Omega + %.0e tr(Omega)/k I' % eps)
                break
    if st != 0:
        logging.getLogger('opt').warning('SDP solution status:' + engine.STATUS_TEXT.get(st, str(st)))
    M = M[0].cpu().numpy()
    Yall = Y[0].cpu().numpy()
    Y = [Yall[i].copy() for i in range(k_)]
    objective = float(value.item())
    if warm_start:
        past_omegas.append(omega)
    return objective, M, Y, create_C(k_, fmin_s)


def sdp_batched(omegas, fmins):
    """B independent problems at once: omegas [B,k_,k_], fmins [B] or scalar ->
    (objective [B], M [B,k_,k_], Y [B,k_,k_,k_], status [B]) as numpy arrays."""
    omegas = np.asarray(omegas, dtype=np.float64)
    assert not np.isnan(omegas).any()
    value, M, Y, status, _ = engine.sdp_batched(omegas, np.asarray(fmins, dtype=np.float64))
    return value.cpu().numpy(), M.cpu().numpy(), Y.cpu().numpy(), status.cpu().numpy()


def _fmin_from_C(C):
    # C_1[-1,-1] = -fmin (sdp.py:151); with k_ = 1 there is no such block and fmin is irrelevant
    return -float(C[1][-1, -1]) if len(C) > 1 else 0.0


def solution_derivative(M, Y, C, domegas, return_dY=False):
    M = np.asarray(M, dtype=np.float64)
    domegas = np.asarray(domegas, dtype=np.float64)
    assert len(domegas.shape) == 2 or len(domegas.shape) == 3
    if len(domegas.shape) == 2:
        domegas = domegas[:, :, None]
    k = domegas.shape[-1]
    Yarr = np.stack([np.asarray(Yi, dtype=np.float64) for Yi in Y])
    if return_dY:
        assert k == 1  # as in the reference (sdp.py:196)
    dM, dY, status = engine.solution_derivative_batched(M[None], Yarr[None], np.array([_fmin_from_C(C)]),
                                                        domegas[None], want_dY=return_dY)
    if int(status.item()) != 0:
        logging.getLogger('opt').warning('SDP sensitivity: numerical breakdown')
    dM = dM[0].cpu().numpy()
    if return_dY:
        dYall = dY[0].cpu().numpy()
        return dM, [dYall[i].copy() for i in range(Yarr.shape[0])]
    return dM


def _tri_walk(n):
    """Row/column indices of the lower triangle walked column by column (SCS ordering)."""
    cols, rows = np.triu_indices(n)
    return rows, cols


def pack(Z, n):
    """Matrix -> SCS vector format (sdp.py:249-264): lower triangle column by column with the
    off-diagonal entries scaled by sqrt(2)."""
    rows, cols = _tri_walk(n)
    z = np.asarray(Z, dtype=np.float64)[rows, cols].copy()
    z[rows != cols] *= np.sqrt(2.0)
    return z


def unpack(z, n):
    """SCS vector format -> symmetric matrix (sdp.py:266-281)."""
    rows, cols = _tri_walk(n)
    vals = np.asarray(z, dtype=np.float64).copy()
    vals[rows != cols] /= np.sqrt(2.0)
    Z = np.zeros((n, n))
    Z[rows, cols] = vals
    Z[cols, rows] = vals
    return Z
