"""Wall time of the hyper-parameter fit the driver runs before every batch (20 restarts, n = 310,
d = 6, fixed noise): objective evaluations per second through one re-factorised device model."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_optimization_amd import hyperfit          # noqa: E402
from bayesian_optimization_amd.engine import KERNEL_MATERN32          # noqa: E402


def hart6ish(X):
    return np.sin(3 * X).sum(axis=1) + 0.3 * (X ** 2).sum(axis=1)


def main():
    for n in (110, 310, 1000):
        rs = np.random.RandomState(1)
        X = rs.uniform(-.5, .5, (n, 6))
        y = hart6ish(X)
        y = (y - y.mean()) / y.std()
        calls = [0]
        orig = hyperfit.objective

        def counted(*a, **k):
            calls[0] += 1
            if '--fresh' in sys.argv:       # a model built and released per evaluation
                k['gp'] = None
            return orig(*a, **k)
        hyperfit.objective = counted
        hyperfit.fit_sequential(X, y, KERNEL_MATERN32, 1.0, 1.0, 1e-6, floor_mode=1, restarts=1, rng=np.random.RandomState(0))
        calls[0] = 0
        t0 = time.perf_counter()
        res = hyperfit.fit_sequential(X, y, KERNEL_MATERN32, 1.0, 1.0, 1e-6, floor_mode=1, restarts=20, rng=np.random.RandomState(0))
        dt = time.perf_counter() - t0
        hyperfit.objective = orig
        print('n=%d: sequential fit with 20 restarts %.3f s, %d objective evaluations, %.2f ms each, nlml %.6f'
              % (n, dt, calls[0], 1e3 * dt / calls[0], res['nlml']), flush=True)
        hyperfit.fit(X, y, KERNEL_MATERN32, 1.0, 1.0, 1e-6, floor_mode=1, restarts=2, rng=np.random.RandomState(0))
        t0 = time.perf_counter()
        res = hyperfit.fit(X, y, KERNEL_MATERN32, 1.0, 1.0, 1e-6, floor_mode=1, restarts=20, rng=np.random.RandomState(0))
        dt = time.perf_counter() - t0
        print('n=%d: batched fit with 20 restarts %.3f s, nlml %.6f' % (n, dt, res['nlml']), flush=True)


if __name__ == '__main__':
    main()
