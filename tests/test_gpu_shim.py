"""GPU tests of the drop-in layer: the reference's call signatures (methods/sdp.py, methods/oei.py)
served by the HIP library, checked against golden fixtures and the oracle.  They read like the
reference's own tests (tests/test_sdp.py, tests/test_derivatives.py) with the absent third-party
comparison tools (MOSEK, numdifftools) replaced by certified fixtures and plain finite differences."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import gp as ogp, hart6, oei as ooei, sdp_ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class _Objective:
    def __init__(self, d=6):
        self.bounds = np.tile(np.array([[-0.5, 0.5]]), (d, 1))
        self.fmin = hart6.HART6_FMIN

    def f(self, X):
        return hart6.hart6_scaled(X)


def _make_oei(n, q, lengthscale=0.5, noise=1e-6, nl_solver='bfgs'):
    from bayesian_optimization_amd.methods.oei import OEI
    from bayesian_optimization_amd.kernels import SafeMatern32
    model = hart6.make_model(n)
    options = dict(objective=_Objective(), kernel=SafeMatern32(6, variance=1.0, lengthscales=lengthscale),
                   noise=noise, batch_size=q, initial_size=n, iterations=1, model_restarts=0,
                   opt_restarts=2, normalize_Y=1, nl_solver=nl_solver, hessian=1, job_name='test')
    bo = OEI(options)
    bo.X = model.X
    bo.Y = model.y[:, None]
    return bo, model


def test_sdp_matches_golden(cuda):
    """methods.sdp.sdp on the reference's test inputs (tests/test_sdp.py:44-74): the reference
    accepts rtol 1e-4 on the value and 1e-2 normwise on M against MOSEK; here 1e-10 / 1e-7."""
    from bayesian_optimization_amd.methods import sdp
    g = np.load(os.path.join(GOLD, 'sdp_reference_inputs.npz'))
    for b in range(10):
        val, M, Y, C = sdp.sdp(g['omega'][b], np.array([g['fmin'][b]]), warm_start=False)
        assert isinstance(val, float) and M.shape == (6, 6) and len(Y) == 6 and len(C) == 6
        np.testing.assert_allclose(val, g['value'][b], rtol=1e-10)
        assert np.linalg.norm(M - g['M'][b]) / np.linalg.norm(g['M'][b]) < 1e-7
        assert np.abs(np.stack(Y) - g['Y'][b]).max() < 1e-6 * np.abs(g['Y'][b]).max()
        assert C[3][-1, 2] == 0.5 and C[3][-1, -1] == -g['fmin'][b]


def test_sdp_nan_raises_and_failure_warns(cuda, caplog):
    from bayesian_optimization_amd.methods import sdp
    om = np.eye(3)
    om[0, 1] = np.nan
    with pytest.raises(AssertionError):
        sdp.sdp(om, 0.0)
    bad = -np.eye(3)                      # not positive definite: warn and continue (sdp.py:53-56)
    bad[2, 2] = 1.0
    with caplog.at_level('WARNING', logger='opt'):
        val, M, Y, C = sdp.sdp(bad, 0.0)
    assert 'SDP solution status' in caplog.text and np.isnan(val)


def test_sdp_singular_psd_omega_is_served(cuda, caplog):
    """The reference hands any symmetric Omega to SCS (sdp.py:34-67); a positive semidefinite but
    singular one (two identical candidate points, no noise) has a finite optimum.  The device
    solver needs a positive definite covariance block, so the shim retries on Omega + eps I and
    warns: finite value, within 1e-5 of the first-order solve of the singular problem."""
    from bayesian_optimization_amd.methods import sdp
    rs = np.random.RandomState(5)
    A = rs.randn(4, 3)
    S = A @ A.T                                     # rank 3, 4 x 4
    mu = rs.randn(4)
    om = np.zeros((5, 5))
    om[:4, :4] = S + np.outer(mu, mu)
    om[:4, 4] = om[4, :4] = mu
    om[4, 4] = 1.0
    assert np.linalg.eigvalsh(om)[0] > -1e-12 and np.linalg.matrix_rank(om) == 4
    with caplog.at_level('WARNING', logger='opt'):
        val, M, Y, C = sdp.sdp(om, -0.3)
    assert np.isfinite(val) and np.isfinite(M).all() and 'regularised' in caplog.text
    # the reference's route: the singular matrix itself through the first-order cone solver
    ref, _, _, _, info = sdp_ref.sdp_first_order(om, -0.3, eps=1e-7, max_iters=50000, return_info=True)
    assert info['status'] == 'Solved' and abs(val - ref) <= 1e-5 * abs(ref)
    # a matrix with a negative eigenvalue still warns and gives NaN
    bad = om - 0.5 * np.eye(5)
    with caplog.at_level('WARNING', logger='opt'):
        val2 = sdp.sdp(bad, -0.3)[0]
    assert np.isnan(val2)


def test_sensitivity_like_reference_test(cuda):
    """tests/test_derivatives.py:82-112 with finite differences of the GPU solve itself."""
    from bayesian_optimization_amd.methods.sdp import sdp, solution_derivative
    K = 3
    bo, model = _make_oei(25, K)
    rs = np.random.RandomState(21)          # not the training design's seed
    X = rs.uniform(-.5, .5, (K, 6))
    omega = bo.omega(X)
    mu = omega[0:K, -1][:, None]
    D_s = rs.rand(K, K)
    D_s = D_s.dot(D_s.T)
    D_m = rs.rand(K, 1)
    D = np.zeros((K + 1, K + 1))
    D[0:K, 0:K] = D_s + mu.dot(D_m.T) + D_m.dot(mu.T)
    D[-1, 0:K] = D_m.flatten()
    D[0:K, -1] = D_m.flatten()
    D = 1e-3 * (D + D.T) / 2
    fmin = np.min(bo.predict_f(bo.X.value)[0])
    _, M, Y, C = sdp(omega, fmin)
    d_opt_val = np.trace(M.T.dot(D))
    dM, dY = solution_derivative(M, Y, C, D, return_dY=True)
    h = 1e-4
    vp, Mp, Yp, _ = sdp(omega + h * D, fmin)
    vm, Mm, Ym, _ = sdp(omega - h * D, fmin)
    np.testing.assert_allclose(d_opt_val, (vp - vm) / (2 * h), rtol=1e-6)       # reference: 1e-2
    sol = np.concatenate([dM.flatten()] + [y.flatten() for y in dY])
    sol_n = np.concatenate([((Mp - Mm) / (2 * h)).flatten()] + [((a - b) / (2 * h)).flatten() for a, b in zip(Yp, Ym)])
    assert np.abs(sol - sol_n).max() < 1e-5 * np.abs(sol_n).max()               # reference: rtol 3e-1


@pytest.mark.parametrize('name', sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, 'oei_*.npz'))))
def test_oei_matches_golden(cuda, name):
    g = np.load(os.path.join(GOLD, name))
    n, q = int(g['n']), int(g['q'])
    bo, model = _make_oei(n, q)
    x = g['X'].ravel()
    value, gradient = bo.acquisition(x)
    assert value.shape == (1,) and gradient.shape == (q * 6,)
    assert abs(value[0] - float(g['value'])) <= 1e-9 * abs(float(g['value']))      # task bar 1e-5
    assert np.abs(gradient - g['gradient']).max() <= 1e-7 * np.abs(g['gradient']).max()   # 1e-4
    v2, _ = bo.acquisition(g['X'])                      # 2-D input as bo.py:150 passes it
    assert v2[0] == value[0]
    H = bo.acquisition_hessian(x)
    assert H.shape == (q * 6, q * 6)
    assert np.abs(H - g['hessian']).max() <= 1e-6 * np.abs(g['hessian']).max()      # task bar 1e-3
    assert np.allclose(bo.omega(g['X']), g['omega'], rtol=1e-9, atol=1e-11)
    if 'domega' in g:
        assert np.abs(bo.domega(x) - g['domega']).max() <= 1e-8 * np.abs(g['domega']).max()


def test_derivatives_like_reference_test(cuda):
    """tests/test_derivatives.py:115-133 (K = 3) with central differences; the reference accepts
    1e-2 normwise on the gradient and 2e-2 on the Hessian."""
    K = 3
    bo, model = _make_oei(25, K)
    X = np.random.RandomState(1).uniform(-.5, .5, (K, 6))
    x = X.flatten()
    gradient, hessian = bo.acquisition(x)[1], bo.acquisition_hessian(x)
    h = 1e-5
    gn = np.array([(bo.acquisition(x + h * e)[0][0] - bo.acquisition(x - h * e)[0][0]) / (2 * h) for e in np.eye(x.size)])
    Hn = np.array([(bo.acquisition(x + h * e)[1] - bo.acquisition(x - h * e)[1]) / (2 * h) for e in np.eye(x.size)])
    assert np.linalg.norm(gradient - gn) / np.linalg.norm(gradient) < 1e-7
    assert np.linalg.norm(hessian - Hn) / np.linalg.norm(hessian) < 1e-6


def test_project_branches(cuda):
    """oei.py:30-37,53-55: reduced problem + V^T gradient; zero Hessian; all points removed."""
    bo, model = _make_oei(25, 4)
    rs = np.random.RandomState(9)
    X = rs.uniform(-.5, .5, (4, 6))
    X[2] = model.X[3] + 1e-4
    np.random.seed(3)
    value, grad = bo.acquisition(X.ravel())
    keep = [0, 1, 3]
    v_red, g_red = bo.acquisition(X[keep].ravel())
    assert value[0] == v_red[0]
    G = grad.reshape(4, 6)
    assert np.allclose(G[keep], g_red.reshape(3, 6), rtol=1e-12, atol=0)
    assert np.array_equal(bo.acquisition_hessian(X.ravel()), np.zeros((24, 24)))
    value0, grad0 = bo.acquisition((model.X[:4] + 1e-5).ravel())
    assert value0[0] == 0 and grad0.shape == (24,)
    # batched path re-evaluates flagged candidates through the same logic
    Xb = np.stack([rs.uniform(-.5, .5, (4, 6)), X])
    vals, grads = bo.acquisition_batched(Xb)
    assert vals[1] == value[0] and grads.shape == (2, 24)


def test_noise_above_threshold_skips_project(cuda):
    bo, model = _make_oei(25, 3, noise=1e-3)
    X = model.X[:3] + 1e-5
    Xu, V = bo.project(X)
    assert Xu is X and np.array_equal(V, np.eye(3))
    value, grad = bo.acquisition(X.ravel())
    assert value[0] < 0 and np.isfinite(grad).all()


def test_get_suggestion_improves(cuda):
    """bo.py:129-192 with the bounded quasi-Newton solver: the suggestion is no worse than its starts."""
    bo, model = _make_oei(25, 3)
    np.random.seed(5)
    X = bo.get_suggestion(3)
    assert X.shape == (3, 6) and (np.abs(X) <= 0.5 + 1e-12).all()
    np.random.seed(5)
    start = bo.random_sample(bo.bounds, 3)
    assert bo.acquisition(X)[0][0] <= bo.acquisition(start)[0][0] + 1e-12


def test_newton_solver_consumes_hessian(cuda):
    from bayesian_optimization_amd.methods.solvers import solve
    bo, model = _make_oei(25, 2, nl_solver='newton')
    np.random.seed(2)
    X0 = bo.random_sample(bo.bounds, 2)
    X, y, status = solve(X0, np.tile(bo.bounds, (2, 1)), True, bo, 'newton')
    assert y <= bo.acquisition(X0)[0][0] + 1e-12 and X.shape == (2, 6)
    with pytest.raises(NameError):
        solve(X0, np.tile(bo.bounds, (2, 1)), True, bo, 'knitro')


def test_multistart_select_on_gpu(cuda):
    """Sharded evaluation on one rank equals the plain arg-min; the winner's X is bit-identical."""
    from bayesian_optimization_amd import DeviceGP, multistart
    model = hart6.make_model(70)
    gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
    X = torch.as_tensor(np.random.RandomState(4).uniform(-.5, .5, (300, 4, 6)), device=cuda)
    value, grad, status, (bv, bi, bx) = multistart.evaluate_and_select(gp, X, index_base=1000)
    k = int(torch.argmin(value))
    assert int(bi) == 1000 + k and float(bv) == float(value[k]) and torch.equal(bx, X[k])
    # two half shards, selected separately then merged by hand, give the same winner
    r0 = multistart.evaluate_and_select(gp, X[:150], index_base=1000)[3]
    r1 = multistart.evaluate_and_select(gp, X[150:], index_base=1150)[3]
    win = r0 if float(r0[0]) <= float(r1[0]) else r1
    assert int(win[1]) == int(bi) and torch.equal(win[2], bx)
