"""Oracle (test infrastructure): objectives and synthetic GP models.

Hartmann-6 on the unit cube, rescaled to [-1/2, 1/2]^6 as the reference does
before building its GP (benchmark_functions.py:5-59 `scale_function`,
benchmark_functions.py:62-114 `hart6`), the output normalisation of
bo.py:241-255 (population standard deviation), and the model recipes used by
the fixtures and the benchmark.  Written vectorised; constants are the
published Hartmann-6 parameters.
"""
import numpy as np

from . import gp

_ALPHA = np.array([1.0, 1.2, 3.0, 3.2])
_A = np.array([[10, 3, 17, 3.5, 1.7, 8],
               [0.05, 10, 17, 0.1, 8, 14],
               [3, 3.5, 1.7, 10, 17, 8],
               [17, 8, 0.05, 10, 0.1, 14]], dtype=np.float64)
_P = 1e-4 * np.array([[1312, 1696, 5569, 124, 8283, 5886],
                      [2329, 4135, 8307, 3736, 1004, 9991],
                      [2348, 1451, 3522, 2883, 3047, 6650],
                      [4047, 8828, 8732, 5743, 1091, 381]], dtype=np.float64)
HART6_FMIN = -3.32237


def hart6_unit(X):
    """Hartmann-6 on [0,1]^6, X [m,6] -> [m,1]."""
    X = np.asarray(X, dtype=np.float64).reshape(-1, 6)
    inner = np.einsum('ij,mij->mi', _A, (X[:, None, :] - _P[None]) ** 2)
    return -(np.exp(-inner) @ _ALPHA)[:, None]


def hart6_scaled(X):
    """The objective on [-1/2,1/2]^6: restore to the unit cube, then evaluate."""
    return hart6_unit(np.asarray(X, dtype=np.float64) + 0.5)


def smooth10(X):
    """Synthetic smooth 10-d objective on [-1/2,1/2]^10 (a fixed sum of Gaussians).
    The reference ships no 10-d function; this only shapes a GP model."""
    X = np.asarray(X, dtype=np.float64).reshape(-1, 10)
    rs = np.random.RandomState(77)
    centres = rs.uniform(-0.5, 0.5, (6, 10))
    widths = rs.uniform(0.3, 0.8, 6)
    amps = rs.uniform(0.5, 2.0, 6)
    d2 = np.sum((X[:, None, :] - centres[None]) ** 2, axis=2)
    return -(np.exp(-d2 / (2 * widths ** 2)) @ amps)[:, None]


def normalize(Y):
    """bo.py:241-255 with normalize_Y=1."""
    Y = np.asarray(Y, dtype=np.float64).copy()
    s = np.std(Y[:, 0])
    if s > 0:
        Y[:, 0] = (Y[:, 0] - np.mean(Y[:, 0])) / s
    return Y


def make_model(n, d=6, train_seed=0, kind=gp.KERNEL_MATERN32, variance=1.0,
               lengthscale=0.5, noise=1e-6, floor_mode=None):
    """Synthetic model recipe: uniform design in [-1/2,1/2]^d, hart6 (d=6) or the
    smooth 10-d objective, normalised outputs, fixed hyper-parameters."""
    rs = np.random.RandomState(train_seed)
    X = rs.uniform(-0.5, 0.5, (n, d))
    y = hart6_scaled(X) if d == 6 else smooth10(X) if d == 10 else \
        np.sin(3.0 * X).sum(axis=1, keepdims=True)
    y = normalize(y)
    kern = gp.Kernel(kind, variance, lengthscale, d, floor_mode)
    return gp.GPModel(X, y[:, 0], kern, noise)
