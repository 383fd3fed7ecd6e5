"""GPU parity of the 256-thread ("wide", q > 32) instances of the Hessian path: the stand-alone
sensitivity solve, dOmega/dx and the acquisition Hessian, up to the largest shape of the
benchmark table (q = 50, d = 10, n = 1000) and the largest q the library accepts.
Task bar: 1e-3 on Hessian entries; the assertions are tighter and say so."""
import os

import numpy as np
import pytest
import torch

from oracle import gp as ogp, hart6, oei as ooei, sdp_ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _mk(n, d=6, lengthscale=0.5):
    from bayesian_optimization_amd import DeviceGP
    om = hart6.make_model(n, d=d, lengthscale=lengthscale)
    dm = DeviceGP(om.X, om.y, ogp.KERNEL_MATERN32, om.kernel.variance, om.kernel.lengthscales,
                  om.noise, floor_mode=om.kernel.floor_mode)
    return om, dm


@pytest.mark.parametrize('n,q,d,ls', [(60, 33, 6, 0.5), (200, 50, 10, 0.8)])
def test_solution_derivative_wide_vs_dense_kkt(cuda, n, q, d, ls):
    """k_soldiff<256>: a few dOmega/dx directions and a generic one against the dense solve of the
    reference's KKT system (sdp.py:159-234; 3927 unknowns at q = 50)."""
    from bayesian_optimization_amd import sdp_batched, solution_derivative_batched
    om, dm = _mk(n, d=d, lengthscale=ls)
    rs = np.random.RandomState(29)
    Xs = rs.uniform(-.5, .5, (q, d))
    omega = om.omega(Xs)
    value, M, Y, status, _ = [t.cpu().numpy() for t in sdp_batched(omega, om.fmin)]
    assert status[0] == 0
    C = sdp_ref.create_C(q + 1, om.fmin)
    dom = om.domega(Xs)[:, :, [0, 7, q * d // 2, q * d - 1]]
    D = rs.randn(q + 1, q + 1)
    dom = np.concatenate([dom, (D + D.T)[:, :, None]], axis=2)
    dM_ref = sdp_ref.solution_derivative(M[0], list(Y[0]), C, dom)
    dM, _, st = solution_derivative_batched(M, Y, om.fmin, dom[None])
    assert st.cpu().numpy()[0] == 0
    scale = np.abs(dM_ref).max(axis=(0, 1), keepdims=True)
    assert (np.abs(dM.cpu().numpy()[0] - dM_ref) / scale).max() < 1e-6


@pytest.mark.parametrize('n,q,d,ls', [(60, 33, 6, 0.5), (200, 50, 10, 0.8), (40, 64, 3, 0.4)])
def test_domega_wide(cuda, n, q, d, ls):
    """k_domega<256> against the oracle's closed-form dOmega/dx."""
    om, dm = _mk(n, d=d, lengthscale=ls)
    Xs = np.random.RandomState(37).uniform(-.5, .5, (2, q, d))
    got = dm.domega(Xs).cpu().numpy()
    for b in range(2):
        ref = om.domega(Xs[b])
        assert np.abs(got[b] - ref).max() <= 1e-8 * np.abs(ref).max()


def test_hessian_wide_vs_oracle(cuda):
    """acquisition_hessian for q > 32 (tables in global memory) against the oracle at q = 40,
    d = 6, a shape the LDS-table version rejected."""
    n, q, d, ls = 50, 40, 6, 0.5
    om, dm = _mk(n, d=d, lengthscale=ls)
    oracle = ooei.OEIOracle(om)
    Xs = np.random.RandomState(47).uniform(-.5, .5, (2, q, d))
    H, status = dm.hessian(Xs)
    assert (status.cpu().numpy() == 0).all()
    H = H.cpu().numpy()
    Href = oracle.acquisition_hessian(Xs[1].ravel())
    scale = np.abs(Href).max()
    assert np.abs(H[1] - H[1].T).max() <= 1e-8 * scale
    assert np.linalg.norm(H[1] - Href) <= 1e-6 * np.linalg.norm(Href)
    assert np.abs(H[1] - Href).max() <= 1e-6 * scale                          # task bar: 1e-3
    # a candidate's Hessian does not depend on its neighbours in the batch
    H1, _ = dm.hessian(Xs[1:])
    assert np.array_equal(H1.cpu().numpy()[0], H[1])


@pytest.mark.parametrize('n,q,d,ls', [(30, 64, 2, 0.4), (80, 57, 12, 0.9)])
def test_hessian_largest_q_finite_differences(cuda, n, q, d, ls):
    """The largest q the library accepts (and the largest d it serves there): rows of the Hessian
    against central differences of the device gradient, itself checked against the oracle in the
    size-class tests (the oracle's dense sensitivity solve needs many minutes at q = 64)."""
    om, dm = _mk(n, d=d, lengthscale=ls)
    X = np.random.RandomState(53).uniform(-.5, .5, (q, d))
    H, status = dm.hessian(X[None])
    assert int(status[0]) == 0
    H = H[0].cpu().numpy()
    scale = np.abs(H).max()
    assert np.abs(H - H.T).max() <= 1e-8 * scale
    rows = [0, 1, q * d // 3, q * d - 1]
    h = 1e-5
    Xp = np.repeat(X.ravel()[None], 2 * len(rows), axis=0)
    for k, r in enumerate(rows):
        Xp[2 * k, r] += h
        Xp[2 * k + 1, r] -= h
    _, g, st, _ = dm.value_grad(Xp.reshape(-1, q, d))
    assert int((st != 0).sum()) == 0
    g = g.cpu().numpy().reshape(2 * len(rows), -1)
    for k, r in enumerate(rows):
        fd = (g[2 * k] - g[2 * k + 1]) / (2 * h)
        assert np.abs(fd - H[r]).max() <= 2e-5 * scale, (r, np.abs(fd - H[r]).max() / scale)


def test_hessian_cfg3_shape(cuda):
    """n = 1000, q = 50, d = 10: every tenth row of the 500 x 500 Hessian, its diagonal, value and
    gradient against the committed oracle fixture (the oracle needs minutes at this size)."""
    g = np.load(os.path.join(GOLD, 'hess_n1000_q50_d10.npz'))
    om, dm = _mk(int(g['n']), d=int(g['d']), lengthscale=float(g['lengthscale']))
    X = g['X']
    value, grad, status, _ = [t.cpu().numpy() for t in dm.value_grad(X[None])]
    assert status[0] == 0
    assert abs(value[0] - float(g['value'])) <= 1e-9 * abs(float(g['value']))           # task bar: 1e-5
    assert np.abs(grad[0].ravel() - g['gradient']).max() <= 1e-7 * np.abs(g['gradient']).max()
    H, hst = dm.hessian(X[None])
    assert int(hst[0]) == 0
    H = H[0].cpu().numpy()
    scale = float(g['H_absmax'])
    assert np.abs(H - H.T).max() <= 1e-8 * scale
    assert np.abs(H[g['rows']] - g['H_rows']).max() <= 1e-6 * scale                      # task bar: 1e-3
    assert np.abs(np.diag(H) - g['H_diag']).max() <= 1e-6 * scale
    assert np.linalg.norm(H[g['rows']] - g['H_rows']) <= 1e-6 * np.linalg.norm(g['H_rows'])


@pytest.mark.parametrize('n,q,d', [(40, 31, 9), (30, 32, 12), (50, 28, 12)])
def test_hessian_register_sizes_with_tables_beyond_lds(cuda, n, q, d):
    """q <= 32 whose per-candidate tables (q d q doubles) no longer fit in LDS beside a wave's
    scratch matrices: the wide kernel with tables in global memory takes over.  Checked against
    central differences of the device gradient along random directions and for symmetry."""
    from bayesian_optimization_amd import DeviceGP
    om = hart6.make_model(n, d=d, lengthscale=0.5 + 0.05 * d)
    dm = DeviceGP(om.X, om.y, ogp.KERNEL_MATERN32, om.kernel.variance, om.kernel.lengthscales, om.noise,
                  floor_mode=om.kernel.floor_mode)
    rs = np.random.RandomState(q + d)
    Xs = rs.uniform(-.5, .5, (2, q, d))
    H, st = dm.hessian(Xs)
    assert int((st != 0).sum()) == 0
    x0 = torch.as_tensor(Xs, device=cuda).reshape(2, -1)
    for _ in range(3):
        v = torch.as_tensor(rs.randn(2, q * d), device=cuda)
        v = v / v.norm(dim=1, keepdim=True)
        h = 1e-5
        gp_ = dm.value_grad((x0 + h * v).reshape(2, q, d))[1]
        gm_ = dm.value_grad((x0 - h * v).reshape(2, q, d))[1]
        fd = (gp_ - gm_).reshape(2, -1) / (2 * h)
        hv = torch.einsum('bij,bj->bi', H, v)
        assert float(((hv - fd).abs().amax(dim=1) / fd.abs().amax(dim=1)).max()) <= 1e-5
    assert float((H - H.transpose(1, 2)).abs().max()) <= 1e-9 * float(H.abs().max())
