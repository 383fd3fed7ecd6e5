"""Lock-step bounded quasi-Newton (host logic, CPU tensors): against scipy's L-BFGS-B, which the
reference uses for the same job (solvers.py:10-21)."""
import numpy as np
import scipy.optimize
import torch

from bayesian_optimization_amd import lockstep
from bayesian_optimization_amd.lockstep import minimize_lockstep


def _rosen(X):
    a, b = X[:, :-1], X[:, 1:]
    f = (100.0 * (b - a * a) ** 2 + (1.0 - a) ** 2).sum(dim=1)
    g = torch.zeros_like(X)
    g[:, :-1] += -400.0 * a * (b - a * a) - 2.0 * (1.0 - a)
    g[:, 1:] += 200.0 * (b - a * a)
    return f, g


def _rosen_np(x):
    f, g = _rosen(torch.as_tensor(x[None]))
    return float(f[0]), g[0].numpy()


def test_rosenbrock_box_matches_lbfgsb():
    """Curved valley with the optimum on a bound (two variables: a single local minimum, so every
    start must reach the point scipy reaches)."""
    rs = np.random.RandomState(0)
    X0 = rs.uniform(-1.5, 1.5, (16, 2))
    lo, hi = -1.5, 0.8                       # the unconstrained optimum (1, 1) is outside
    res = minimize_lockstep(_rosen, torch.as_tensor(X0), lo, hi, max_iter=500, pgtol=1e-8, ftol=1e-15)
    ref = scipy.optimize.minimize(_rosen_np, X0[0], jac=True, method='L-BFGS-B', bounds=[(lo, hi)] * 2,
                                  options={'gtol': 1e-10, 'ftol': 1e-16, 'maxiter': 500})
    assert np.allclose(ref.x, [0.8, 0.64], atol=1e-6)
    for b in range(16):
        assert abs(res.fun[b].item() - ref.fun) <= 1e-9
        assert np.allclose(res.x[b].numpy(), ref.x, atol=1e-5)
        assert (res.x[b] >= lo).all() and (res.x[b] <= hi).all()


def test_quadratic_active_bounds_exact():
    rs = np.random.RandomState(1)
    n = 12
    A = rs.randn(n, n)
    Q = torch.as_tensor(A @ A.T + n * np.eye(n))
    c = torch.as_tensor(rs.randn(n) * 10)

    def fun(X):
        g = X @ Q - c
        return 0.5 * (X * (X @ Q)).sum(dim=1) - X @ c, g
    X0 = torch.as_tensor(rs.uniform(-1, 1, (8, n)))
    res = minimize_lockstep(fun, X0, -0.2, 0.3, max_iter=300, pgtol=1e-10, ftol=0.0)
    ref = scipy.optimize.minimize(lambda x: (lambda f, g: (float(f[0]), g[0].numpy()))(*fun(torch.as_tensor(x[None]))),
                                  X0[0].numpy(), jac=True, method='L-BFGS-B', bounds=[(-0.2, 0.3)] * n,
                                  options={'gtol': 1e-12, 'ftol': 1e-16})
    assert res.converged.all()
    for b in range(8):            # strictly convex: every start reaches the same constrained optimum
        assert np.allclose(res.x[b].numpy(), ref.x, atol=1e-7)


def test_rejects_non_finite_and_keeps_others():
    def fun(X):
        f = (X * X).sum(dim=1)
        f = torch.where(X[:, 0] > 0.5, torch.full_like(f, float('inf')), f)
        return f, 2 * X
    X0 = torch.tensor([[0.9, 0.9], [0.4, -0.3], [-0.2, 0.1]], dtype=torch.float64)
    res = minimize_lockstep(fun, X0, -1.0, 1.0)
    assert not res.converged[0] and res.message[0] == 'no finite start value'
    assert res.converged[1] and res.converged[2]
    assert torch.allclose(res.x[1:], torch.zeros(2, 2, dtype=torch.float64), atol=1e-5)


def test_newton_lockstep_rosenbrock_active_bound():
    """Projected Newton in lock-step (the batched consumer of exact Hessians): Rosenbrock with
    the upper bound cutting off the unconstrained minimum; every run ends at (0.8, 0.64)."""
    def hess(X):
        x, y = X[:, 0], X[:, 1]
        H = torch.zeros(X.shape[0], 2, 2, dtype=X.dtype)
        H[:, 0, 0] = 2 - 400 * (y - x * x) + 800 * x * x
        H[:, 0, 1] = H[:, 1, 0] = -400 * x
        H[:, 1, 1] = 200
        return H

    def fun(X):
        x, y = X[:, 0], X[:, 1]
        f = (1 - x) ** 2 + 100 * (y - x * x) ** 2
        return f, torch.stack([-2 * (1 - x) - 400 * x * (y - x * x), 200 * (y - x * x)], 1)
    X0 = torch.tensor(np.random.RandomState(0).uniform(-1.5, 1.5, (16, 2)))
    res = lockstep.minimize_lockstep_newton(fun, hess, X0, -2.0, 0.8)
    assert bool(res.converged.all())
    assert torch.allclose(res.x, torch.tensor([0.8, 0.64], dtype=torch.float64).expand(16, 2), atol=1e-6)
    assert int(res.nfev.max()) <= 60
    # a Hessian that is useless (zeros, NaN) degrades to gradient steps, never to a failure
    res2 = lockstep.minimize_lockstep_newton(fun, lambda X: torch.full((X.shape[0], 2, 2), float('nan'), dtype=X.dtype),
                                             X0[:4], -2.0, 0.8, max_iter=5)
    assert bool((res2.fun <= fun(X0[:4])[0]).all())
