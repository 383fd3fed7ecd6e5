"""Kernel descriptions with the attribute names the reference reads from gpflow 0.5 kernel objects
(`kern.variance.value`, `kern.lengthscales.value`, `ARD`; run.py:79-98, oei.py:177).  They carry
hyper-parameters only; all arithmetic happens on the GPU."""
import numpy as np

from .engine import KERNEL_RBF, KERNEL_MATERN32, FLOOR_NONE, FLOOR_SAFE, FLOOR_STOCK


class Param:
    """Minimal stand-in for a gpflow Param: a value and a `fixed` flag."""

    def __init__(self, value):
        self.value = np.asarray(value, dtype=np.float64)
        self.fixed = False

    def __float__(self):
        return float(self.value)

    def __repr__(self):
        return 'Param(%r)' % (self.value,)


class _Stationary:
    kind = None
    floor_mode = FLOOR_NONE

    def __init__(self, input_dim, variance=1.0, lengthscales=None, ARD=False):
        self.input_dim = int(input_dim)
        self.ARD = bool(ARD)
        if lengthscales is None:
            lengthscales = np.ones(self.input_dim) if self.ARD else 1.0
        self.variance = Param(variance)
        self.lengthscales = Param(lengthscales)

    def lengthscale_vector(self):
        ls = np.atleast_1d(self.lengthscales.value).astype(np.float64)
        return np.full(self.input_dim, ls[0]) if ls.size == 1 else ls


class RBF(_Stationary):
    """variance * exp(-r^2 / 2)   (run.py:80)."""
    kind = KERNEL_RBF
    floor_mode = FLOOR_NONE


class Matern32(_Stationary):
    """variance * (1 + sqrt3 r) exp(-sqrt3 r) with r = sqrt(r^2 + 1e-12), the distance convention
    of the stock gpflow 0.5 kernel used by tests/create_model.py:28."""
    kind = KERNEL_MATERN32
    floor_mode = FLOOR_STOCK


class SafeMatern32(Matern32):
    """Matern-3/2 with r = sqrt(max(r^2, 1e-40))   (run.py:19-23)."""
    floor_mode = FLOOR_SAFE
