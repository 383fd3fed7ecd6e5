"""Objective functions for the driver, with the interface the reference's optimisation loop expects
from `benchmark_functions.py`: an object with `bounds [d, 2]`, optional `fmin`, and
`f(X [m, d]) -> [m, 1]`; `scale_function` presents any of them on [-1/2, 1/2]^d
(benchmark_functions.py:5-59).  Hartmann-6 uses the published constants."""
import numpy as np


class hart6:
    """Hartmann 6-dimensional function on the unit cube; global minimum -3.32237."""
    _alpha = np.array([1.0, 1.2, 3.0, 3.2])
    _A = np.array([[10, 3, 17, 3.5, 1.7, 8],
                   [0.05, 10, 17, 0.1, 8, 14],
                   [3, 3.5, 1.7, 10, 17, 8],
                   [17, 8, 0.05, 10, 0.1, 14]], dtype=np.float64)
    _P = 1e-4 * np.array([[1312, 1696, 5569, 124, 8283, 5886],
                          [2329, 4135, 8307, 3736, 1004, 9991],
                          [2348, 1451, 3522, 2883, 3047, 6650],
                          [4047, 8828, 8732, 5743, 1091, 381]], dtype=np.float64)

    def __init__(self):
        self.bounds = np.array([[0.0, 1.0]] * 6)
        self.fmin = -3.32237
        self.min = np.array([0.20169, 0.150011, 0.476874, 0.275332, 0.311652, 0.6573])

    def f(self, X):
        X = np.asarray(X, dtype=np.float64).reshape(-1, 6)
        inner = np.einsum('ij,mij->mi', self._A, (X[:, None, :] - self._P[None]) ** 2)
        return -(np.exp(-inner) @ self._alpha)[:, None]


class scale_function:
    """Presents `function` on the box [-1/2, 1/2]^d."""

    def __init__(self, function):
        self.function = function
        self._lo = np.asarray(function.bounds, dtype=np.float64)[:, 0].copy()
        self._len = np.asarray(function.bounds, dtype=np.float64)[:, 1] - self._lo
        self.bounds = np.tile(np.array([[-0.5, 0.5]]), (len(self._lo), 1))
        if hasattr(function, 'fmin'):
            self.fmin = function.fmin

    def restore(self, X):
        return (np.asarray(X, dtype=np.float64) + 0.5) * self._len + self._lo

    def scale(self, X):
        return (np.asarray(X, dtype=np.float64) - self._lo) / self._len - 0.5

    def f(self, X):
        ret = self.function.f(self.restore(X))
        if isinstance(ret, tuple):
            return ret[0], self.scale(ret[1])
        return ret
