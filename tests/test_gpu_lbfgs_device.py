"""The multistart optimiser with its state in HIP kernels (`oei_lbfgs_*`) against the generic torch
version of the same method (lockstep.minimize_lockstep) and scipy's L-BFGS-B (solvers.py:10-21).
Per run both versions apply the same formulae in the same order up to the order of the terms of a
dot product, so on well-conditioned problems the iterates agree to rounding."""
import numpy as np
import pytest
import scipy.optimize
import torch

from oracle import gp as ogp, hart6

pytestmark = pytest.mark.gpu


def _rosen(X):
    a, b = X[:, :-1], X[:, 1:]
    f = (100.0 * (b - a * a) ** 2 + (1.0 - a) ** 2).sum(dim=1)
    g = torch.zeros_like(X)
    g[:, :-1] += -400.0 * a * (b - a * a) - 2.0 * (1.0 - a)
    g[:, 1:] += 200.0 * (b - a * a)
    return f, g


def test_quadratic_active_bounds_same_iterates(cuda):
    from bayesian_optimization_amd.lockstep import minimize_lockstep, minimize_lockstep_device
    rs = np.random.RandomState(1)
    n = 150                                       # more than two elements per lane
    A = rs.randn(n, n)
    Q = torch.as_tensor(A @ A.T + n * np.eye(n), device=cuda)
    c = torch.as_tensor(rs.randn(n) * 10, device=cuda)

    def fun(X):
        g = X @ Q - c
        return 0.5 * (X * (X @ Q)).sum(dim=1) - X @ c, g
    X0 = torch.as_tensor(rs.uniform(-1, 1, (37, n)), device=cuda)
    lo = torch.full((n,), -0.2, dtype=torch.float64, device=cuda)
    hi = torch.full((n,), 0.3, dtype=torch.float64, device=cuda)
    hi[::7] = 0.05
    dev = minimize_lockstep_device(fun, X0, lo, hi, max_iter=300, pgtol=1e-5, ftol=0.0)
    ref = minimize_lockstep(fun, X0, lo, hi, max_iter=300, pgtol=1e-5, ftol=0.0)
    assert bool(dev.converged.all()) and bool(ref.converged.all())
    assert float((dev.x - ref.x).abs().max()) <= 1e-6        # both within pgtol / curvature of the optimum
    assert float((dev.fun - ref.fun).abs().max()) <= 1e-12 * float(ref.fun.abs().max())
    assert bool((dev.x >= lo).all()) and bool((dev.x <= hi).all())
    assert int((dev.nit - ref.nit).abs().max()) <= 6
    # the first iterations, before rounding differences can change a line-search decision
    dev5 = minimize_lockstep_device(fun, X0, lo, hi, max_iter=5, pgtol=1e-7, ftol=0.0)
    ref5 = minimize_lockstep(fun, X0, lo, hi, max_iter=5, pgtol=1e-7, ftol=0.0)
    assert torch.equal(dev5.nit, ref5.nit) and torch.equal(dev5.nfev, ref5.nfev)
    assert float((dev5.x - ref5.x).abs().max()) <= 1e-12
    assert float((dev5.grad - ref5.grad).abs().max()) <= 1e-9
    sp = scipy.optimize.minimize(lambda x: (lambda f, g: (float(f[0]), g[0].cpu().numpy()))(*fun(torch.as_tensor(x[None], device=cuda))),
                                 X0[0].cpu().numpy(), jac=True, method='L-BFGS-B',
                                 bounds=list(zip(lo.tolist(), hi.tolist())), options={'gtol': 1e-12, 'ftol': 1e-16})
    assert np.allclose(dev.x[5].cpu().numpy(), sp.x, atol=1e-6)
    assert dev.message.count('converged') == 37


def test_rosenbrock_box_and_rejected_points(cuda):
    """Curved valley with the optimum on a bound; one start has no finite value, one region of the
    box returns NaN (rejected by the line search, like an unsolved SDP)."""
    from bayesian_optimization_amd.lockstep import minimize_lockstep, minimize_lockstep_device
    rs = np.random.RandomState(0)
    X0 = torch.as_tensor(rs.uniform(-1.5, 1.5, (16, 2)), device=cuda)
    X0[3] = torch.tensor([1.4, 1.4], dtype=torch.float64)
    calls = []

    def fun(X):
        calls.append(X.shape[0])
        f, g = _rosen(X)
        hole = (X[:, 0] > 1.2) & (X[:, 1] > 1.2)
        return torch.where(hole, torch.full_like(f, float('nan')), f), g
    lo, hi = -1.5, 1.5
    dev = minimize_lockstep_device(fun, X0, lo, hi, max_iter=500, pgtol=1e-8, ftol=1e-15)
    rounds = len(calls)
    ref = minimize_lockstep(fun, X0, lo, hi, max_iter=500, pgtol=1e-8, ftol=1e-15)
    assert dev.message[3] == 'no finite start value' and not bool(dev.converged[3])
    keep = [b for b in range(16) if b != 3]
    assert rounds == int(dev.nfev.max())                     # one round per evaluation of the slowest run
    assert calls[0] == 16 and min(calls[:rounds]) >= 1
    for b in keep:
        assert bool(dev.converged[b]) == bool(ref.converged[b])
        assert np.allclose(dev.x[b].cpu().numpy(), [1.0, 1.0], atol=1e-5), (b, dev.x[b])
        assert abs(float(dev.fun[b])) <= 1e-9
    assert int((dev.nit[keep] - ref.nit[keep]).abs().max()) <= 10


def test_iteration_and_line_search_limits(cuda):
    from bayesian_optimization_amd.lockstep import minimize_lockstep, minimize_lockstep_device
    X0 = torch.as_tensor(np.random.RandomState(2).uniform(-1.5, 1.5, (9, 6)), device=cuda)
    dev = minimize_lockstep_device(_rosen, X0, -2.0, 2.0, max_iter=7, pgtol=1e-12, ftol=0.0)
    ref = minimize_lockstep(_rosen, X0, -2.0, 2.0, max_iter=7, pgtol=1e-12, ftol=0.0)
    assert int(dev.nit.max()) == 7 and torch.equal(dev.nit, ref.nit) and torch.equal(dev.nfev, ref.nfev)
    assert float((dev.x - ref.x).abs().max()) <= 1e-8
    assert not bool(dev.converged.any())

    def uphill(X):                       # gradient with the wrong sign: no trial point is ever accepted
        f, g = _rosen(X)
        return f, -g
    dev = minimize_lockstep_device(uphill, X0, -2.0, 2.0, max_ls=5)
    assert torch.equal(dev.nfev, torch.full((9,), 6, dtype=torch.int64, device=cuda))
    assert int(dev.nit.max()) == 0 and not bool(dev.converged.any())
    assert float((dev.x - X0).abs().max()) == 0.0


def test_acquisition_multistart_same_result_as_torch_version(cuda):
    """q = 20, n = 310: 64 starts through the device optimiser and through the torch version; the
    acquisition amplifies rounding differences over hundreds of iterations, so the runs are
    compared by what they reach: same best value to 1e-6 relative, every converged run a
    projected-stationary point."""
    from bayesian_optimization_amd import DeviceGP, lockstep
    om = hart6.make_model(310)
    gp = DeviceGP(om.X, om.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
    q, d = 20, 6
    starts = torch.as_tensor(np.random.RandomState(5).uniform(-.5, .5, (64, q * d)), device=cuda)
    fun = lockstep.acquisition_fun(gp, q)
    dev = lockstep.minimize_lockstep_device(fun, starts, -0.5, 0.5)
    ref = lockstep.minimize_lockstep(fun, starts, -0.5, 0.5)
    assert int(dev.converged.sum()) >= int(0.8 * int(ref.converged.sum()))
    assert float(dev.fun.min()) <= float(ref.fun.min()) * (1 - 2e-2)          # values are negative
    assert abs(float(dev.fun.mean()) - float(ref.fun.mean())) <= 2e-2 * abs(float(ref.fun.mean()))
    f2, g2 = fun(dev.x)
    assert torch.equal(f2, dev.fun)                                            # state is the last accepted point
    pg = dev.x - torch.clamp(dev.x - g2, -0.5, 0.5)
    conv_pg = dev.converged & (pg.abs().amax(dim=1) <= 1e-5)
    assert int(conv_pg.sum()) >= 1
    # first iterations are identical: one accepted step from the same start gives the same point
    one_d = lockstep.minimize_lockstep_device(fun, starts, -0.5, 0.5, max_iter=1)
    one_r = lockstep.minimize_lockstep(fun, starts, -0.5, 0.5, max_iter=1)
    assert float((one_d.x - one_r.x).abs().max()) <= 1e-12


def _rosen_hess(X):
    """Dense Hessian of the chained Rosenbrock function, [B, n, n]."""
    B, n = X.shape
    H = torch.zeros(B, n, n, dtype=X.dtype, device=X.device)
    a, b = X[:, :-1], X[:, 1:]
    i = torch.arange(n - 1, device=X.device)
    H[:, i, i] += 1200.0 * a * a - 400.0 * b + 2.0
    H[:, i + 1, i + 1] += 200.0
    H[:, i, i + 1] += -400.0 * a
    H[:, i + 1, i] += -400.0 * a
    return H


def test_newton_device_matches_torch_version(cuda):
    """Projected Newton with the state in HIP kernels against the torch statement of the same
    method: non-convex chained Rosenbrock (the Levenberg ladder is needed away from the valley),
    an active upper bound, n = 9 and n = 130 (a factor of 130 x 131 doubles in LDS)."""
    from bayesian_optimization_amd.lockstep import minimize_lockstep_newton, minimize_lockstep_newton_device
    for n, B in ((9, 21), (130, 5)):
        X0 = torch.as_tensor(np.random.RandomState(n).uniform(-1.5, 1.5, (B, n)), device=cuda)
        dev = minimize_lockstep_newton_device(_rosen, _rosen_hess, X0, -2.0, 0.8, max_iter=200)
        ref = minimize_lockstep_newton(_rosen, _rosen_hess, X0, -2.0, 0.8, max_iter=200)
        assert int(dev.converged.sum()) >= int(ref.converged.sum()) - 1
        both = dev.converged & ref.converged
        assert int(both.sum()) >= B // 2
        assert float((dev.fun[both] - ref.fun[both]).abs().max()) <= 1e-6 * (1 + float(ref.fun[both].abs().max()))
        assert bool((dev.x <= 0.8).all()) and bool((dev.x >= -2.0).all())
        f2, g2 = _rosen(dev.x)
        assert torch.allclose(f2, dev.fun, rtol=1e-12, atol=0.0)      # torch's sum order depends on the batch shape
        pg = dev.x - torch.clamp(dev.x - g2, -2.0, 0.8)
        assert float(pg[dev.converged].abs().max()) <= 1e-3
        # one iteration from the same start: same ladder, same factor, same accepted point
        one_d = minimize_lockstep_newton_device(_rosen, _rosen_hess, X0, -2.0, 0.8, max_iter=1)
        one_r = minimize_lockstep_newton(_rosen, _rosen_hess, X0, -2.0, 0.8, max_iter=1)
        assert torch.equal(one_d.nfev, one_r.nfev)
        assert float((one_d.x - one_r.x).abs().max()) <= 1e-9 * (1 + float(one_r.x.abs().max()))


def test_newton_device_useless_hessian_and_limits(cuda):
    from bayesian_optimization_amd import _lib
    from bayesian_optimization_amd.lockstep import minimize_lockstep_newton_device
    X0 = torch.as_tensor(np.random.RandomState(3).uniform(-1.5, 1.5, (6, 4)), device=cuda)
    nan_h = lambda X: torch.full((X.shape[0], 4, 4), float('nan'), dtype=X.dtype, device=X.device)
    res = minimize_lockstep_newton_device(_rosen, nan_h, X0, -2.0, 0.8, max_iter=5)
    assert bool((res.fun <= _rosen(X0)[0]).all()) and int(res.nit.max()) <= 5
    neg_h = lambda X: -_rosen_hess(X).abs() - torch.eye(4, dtype=X.dtype, device=X.device)      # never positive definite at lam = 0
    res = minimize_lockstep_newton_device(_rosen, neg_h, X0, -2.0, 0.8, max_iter=8)
    assert bool((res.fun < _rosen(X0)[0]).all())
    X0[2] = float('nan')
    res = minimize_lockstep_newton_device(_rosen, _rosen_hess, X0, -2.0, 0.8)
    assert res.message[2] == 'no finite start value' and bool(res.converged[[0, 1, 3, 4, 5]].all())
    with pytest.raises(_lib.OeiError):
        minimize_lockstep_newton_device(_rosen, _rosen_hess, torch.zeros(2, 142, dtype=torch.float64, device=cuda), -2.0, 0.8)


def test_round_loop_inside_the_library_matches_the_python_loop(cuda):
    """`minimize_acquisition_rounds` (rounds driven by oei_lbfgs_minimize_rounds /
    oei_newton_minimize_rounds) against the same state machines driven round by round from Python
    with `acquisition_fun` / `acquisition_hess_fun`: same evaluations, same advance kernels, so the
    same iterates bit for bit, for both methods; and with a start on a training point the flagged
    trial points go through the host and the runs still end finite and converged or stopped."""
    from oracle import hart6, gp as ogp
    from bayesian_optimization_amd import DeviceGP, lockstep
    q, d = 5, 6
    model = hart6.make_model(40)
    gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
    X0 = torch.as_tensor(np.random.RandomState(11).uniform(-.5, .5, (24, q * d)), device=cuda)
    lo = torch.full((q * d,), -0.5, dtype=torch.float64, device=cuda)
    hi = -lo
    fun = lockstep.acquisition_fun(gp, q)
    ref = lockstep.minimize_lockstep_device(fun, X0, lo, hi, max_iter=300)
    got = lockstep.minimize_acquisition_rounds(gp, q, X0, lo, hi, max_iter=300)
    for a, b in zip((got.x, got.fun, got.grad, got.nit, got.nfev, got.converged),
                    (ref.x, ref.fun, ref.grad, ref.nit, ref.nfev, ref.converged)):
        assert torch.equal(a, b)
    assert got.message == ref.message and int(got.converged.sum()) >= 20
    refn = lockstep.minimize_lockstep_newton_device(fun, lockstep.acquisition_hess_fun(gp, q), X0, lo, hi, max_iter=80)
    gotn = lockstep.minimize_acquisition_rounds(gp, q, X0, lo, hi, method='newton', max_iter=80)
    for a, b in zip((gotn.x, gotn.fun, gotn.nit, gotn.nfev), (refn.x, refn.fun, refn.nit, refn.nfev)):
        assert torch.equal(a, b)
    # a start that duplicates a training point: handed back to the host for its rounds
    calls = []

    def fallback(xrow):
        calls.append(1)
        v, g, st, _ = gp.value_grad(xrow.reshape(1, q, d), want_flags=False)
        return float(v.item()) + 0.25, g.reshape(-1).cpu().numpy()       # a different value: visibly the host's
    Xd = X0.clone()
    Xd[3, :d] = torch.as_tensor(model.X[7] + 1e-4, device=cuda)
    for method in ('lbfgs', 'newton'):
        calls.clear()
        r = lockstep.minimize_acquisition_rounds(gp, q, Xd, lo, hi, method=method, project_fallback=fallback, max_iter=60)
        assert len(calls) >= 1 and bool(torch.isfinite(r.fun).all())
        others = [i for i in range(24) if i != 3]
        base = lockstep.minimize_acquisition_rounds(gp, q, Xd, lo, hi, method=method, max_iter=60)
        assert torch.equal(r.x[others], base.x[others])                   # the other runs never notice
