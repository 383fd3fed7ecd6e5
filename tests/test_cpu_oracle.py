"""CPU tests of the oracle: golden vectors, solver-independent certificates, finite differences.
None of these needs a GPU."""
import glob
import os

import numpy as np
import pytest

from oracle import gp as ogp, hart6, oei as ooei, reduced_newton, sdp_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _cert_max(c):
    return max(v for k, v in c.items() if k not in ('pobj', 'dobj'))


def test_pack_unpack_roundtrip():
    rs = np.random.RandomState(0)
    Z = rs.randn(6, 6)
    Z = Z + Z.T
    z = sdp_ref.pack(Z)
    assert z.shape == (21,)
    assert np.allclose(sdp_ref.unpack(z, 6), Z)
    # inner products are preserved by the sqrt(2) scaling
    W = rs.randn(6, 6)
    W = W + W.T
    assert np.isclose(z @ sdp_ref.pack(W), np.sum(Z * W))


def test_golden_sdp_reference_inputs():
    """The reference's own SDP test inputs (tests/test_sdp.py:45-62): stored optimum vs a fresh
    reference-form solve, the KKT certificate, and the reference's own acceptance bars for the
    first-order solve at eps = 1e-5 (tests/test_sdp.py:69-72)."""
    g = np.load(os.path.join(GOLD, 'sdp_reference_inputs.npz'))
    for b in range(0, 10, 3):
        om, fm = g['omega'][b], g['fmin'][b]
        cert = sdp_ref.kkt_certificate(om, fm, g['M'][b], list(g['Y'][b]))
        assert _cert_max(cert) < 1e-9
        v, M, Y, _ = sdp_ref.sdp_first_order(om, fm, eps=1e-5)
        assert np.isclose(v, g['value'][b], rtol=1e-3)
        assert np.linalg.norm(M - g['M'][b]) / np.linalg.norm(g['M'][b]) < 5e-2
        v2, M2, _, _, _ = reduced_newton.solve(om, fm)
        assert abs(v2 - g['value'][b]) < 1e-11 * abs(g['value'][b])
        assert np.abs(M2 - g['M'][b]).max() < 1e-7 * np.abs(g['M'][b]).max()


def test_q1_closed_form():
    rs = np.random.RandomState(3)
    for _ in range(5):
        mu, s2, fm = rs.randn(), rs.uniform(0.05, 2.0), rs.randn()
        om = np.array([[s2 + mu * mu, mu], [mu, 1.0]])
        v = sdp_ref.sdp_exact(om, fm)[0]
        assert np.isclose(v, -0.5 * (np.sqrt(s2 + (mu - fm) ** 2) - (mu - fm)), rtol=1e-10)


@pytest.mark.parametrize('name', sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, 'oei_*.npz'))))
def test_golden_oei(name):
    g = np.load(os.path.join(GOLD, name))
    n, q = int(g['n']), int(g['q'])
    model = hart6.make_model(n)
    assert np.array_equal(model.X, g['X_train']) and np.allclose(model.y, g['y_train'], atol=1e-14)
    assert np.isclose(model.fmin, float(g['fmin']), rtol=1e-10)
    X = g['X']
    assert np.allclose(model.omega(X), g['omega'], rtol=1e-10, atol=1e-12)
    cert = sdp_ref.kkt_certificate(g['omega'], model.fmin, g['M'],
                                   reduced_newton.solve(g['omega'], model.fmin)[2])
    assert _cert_max(cert) < 1e-8
    v, M, _, _, _ = reduced_newton.solve(g['omega'], model.fmin)
    assert abs(v - float(g['value'])) < 1e-10 * abs(float(g['value']))
    _, grad = model.inner_and_grad(X, g['M'])
    assert np.allclose(grad.ravel(), g['gradient'], rtol=1e-8, atol=1e-11)
    if q <= 3:
        H = ooei.OEIOracle(model).acquisition_hessian(X.ravel())
        assert np.abs(H - g['hessian']).max() < 1e-7 * np.abs(g['hessian']).max()


def test_sensitivity_reduced_vs_dense():
    model = hart6.make_model(25)
    rs = np.random.RandomState(5)
    X = rs.uniform(-.5, .5, (4, 6))
    om = model.omega(X)
    v, M, Y, p, _ = reduced_newton.solve(om, model.fmin)
    C = sdp_ref.create_C(5, model.fmin)
    D = rs.randn(5, 5)
    D = D + D.T
    dM_ref = sdp_ref.solution_derivative(M, Y, C, D)
    w0, mu, S = reduced_newton.split_omega(om)
    st = reduced_newton.state(p, mu, S, model.fmin)
    dM = reduced_newton.sensitivity(p, st, mu, S, model.fmin, D, w0)
    assert np.abs(dM - dM_ref).max() < 1e-9 * np.abs(dM_ref).max()


@pytest.mark.parametrize('kind,fm', [(ogp.KERNEL_MATERN32, 1), (ogp.KERNEL_MATERN32, 2), (ogp.KERNEL_RBF, 0)])
def test_gp_derivatives_fd(kind, fm):
    m = hart6.make_model(25, kind=kind, floor_mode=fm)
    q = 3
    X = np.random.RandomState(2).uniform(-.5, .5, (q, 6))
    M = np.random.RandomState(3).randn(q + 1, q + 1)
    M = M + M.T
    x0, h = X.ravel(), 1e-6
    F = lambda x: np.sum(m.omega(x.reshape(q, 6)) * M)
    G = lambda x: m.inner_and_grad(x.reshape(q, 6), M)[1].ravel()
    _, g = m.inner_and_grad(X, M)
    gn = np.array([(F(x0 + h * e) - F(x0 - h * e)) / (2 * h) for e in np.eye(q * 6)])
    assert np.linalg.norm(gn - g.ravel()) < 1e-7 * np.linalg.norm(gn)
    Hn = np.array([(G(x0 + h * e) - G(x0 - h * e)) / (2 * h) for e in np.eye(q * 6)])
    H = m.inner_hessian(X, M)
    assert np.linalg.norm(H - Hn) < 1e-7 * np.linalg.norm(Hn)
    dom = m.domega(X)
    domn = np.stack([(m.omega((x0 + h * e).reshape(q, 6)) - m.omega((x0 - h * e).reshape(q, 6))) / (2 * h)
                     for e in np.eye(q * 6)], axis=2)
    assert np.linalg.norm(dom - domn) < 1e-7 * np.linalg.norm(domn)


def test_gp_against_sklearn():
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import Matern, ConstantKernel
    m = hart6.make_model(25, floor_mode=0)
    gpr = GaussianProcessRegressor(kernel=ConstantKernel(1.0, 'fixed') * Matern(length_scale=0.5, length_scale_bounds='fixed', nu=1.5),
                                   alpha=1e-6, optimizer=None).fit(m.X, m.y)
    Xs = np.random.RandomState(2).uniform(-.5, .5, (5, 6))
    mu, cov = gpr.predict(Xs, return_cov=True)
    mu2, cov2 = m.predict_full(Xs)
    assert np.allclose(mu, mu2, atol=1e-10) and np.allclose(cov, cov2, atol=1e-10)


def test_project_cases():
    """oei.py:157-199: no duplicates / duplicate of the dataset / mutual duplicates / all removed."""
    m = hart6.make_model(25)
    o = ooei.OEIOracle(m)
    rs = np.random.RandomState(9)
    X = rs.uniform(-.5, .5, (4, 6))
    Xu, V = o.project(X, rs)
    assert Xu is X and np.array_equal(V, np.eye(4))
    X1 = X.copy()
    X1[2] = m.X[3] + 1e-4
    Xu, V = o.project(X1, rs)
    assert Xu.shape == (3, 6) and V.shape == (3, 4) and np.isclose(np.linalg.norm(V[:, 2]), 1.0)
    X2 = X.copy()
    X2[3] = X2[0] + 1e-5
    Xu, V = o.project(X2, rs)
    assert Xu.shape == (3, 6)
    Xu, V = o.project(m.X[:4] + 1e-5, rs)
    assert Xu.shape == (0, 6)
    v, g = o.acquisition((m.X[:4] + 1e-5).ravel(), rng=rs)
    assert v[0] == 0 and g.shape == (24,)
    assert np.array_equal(o.acquisition_hessian(X1.ravel(), rng=rs), np.zeros((24, 24)))


def test_hart6_matches_published_minimum():
    xmin = np.array([0.20169, 0.150011, 0.476874, 0.275332, 0.311652, 0.6573])
    assert abs(hart6.hart6_unit(xmin)[0, 0] - hart6.HART6_FMIN) < 1e-4


def test_objective_against_reference_run():
    """hart6 / scale_function of the oracle and of the product driver against outputs of the
    reference's own benchmark_functions.py (tests/golden/make_hart6_fixture.py ran it; the
    fixture holds inputs and outputs only)."""
    from bayesian_optimization_amd import benchmarks
    g = np.load(os.path.join(GOLD, 'hart6_reference.npz'))
    assert np.abs(hart6.hart6_unit(g['X_unit']) - g['f_unit']).max() < 1e-14
    assert np.abs(hart6.hart6_scaled(g['X_scaled']) - g['f_scaled']).max() < 1e-14
    prod = benchmarks.scale_function(benchmarks.hart6())
    assert np.abs(benchmarks.hart6().f(g['X_unit']) - g['f_unit']).max() < 1e-14
    assert np.abs(prod.f(g['X_scaled']) - g['f_scaled']).max() < 1e-14
    assert np.abs(prod.restore(g['X_scaled']) - g['restored']).max() < 1e-15
    assert np.abs(prod.scale(g['X_unit']) - g['rescaled']).max() < 1e-15
    assert np.array_equal(prod.bounds, g['bounds']) and prod.fmin == float(g['fmin'])
