"""GPU tests against the extended-precision fixtures tests/golden/illcond_rbf_*.npz.

The cases are RBF problems at noise 1e-6 .. 5e-6 (cond(K) 1e8 .. 1e9) from a randomised sweep on
which the device gradient and the fp64 CPU restatement differed by up to 1.3e-4: more than the
task's gradient bar, with neither side known to be right.  The fixtures hold Omega, the gradient
and (one case) the Hessian with every GP quantity computed in 256-bit fixed point
(oracle/gp_mp.py; tests/golden/make_illcond_fixture.py regenerates them), and the fp64
restatement's own distance to that.  Bars are the task's: 1e-5 value, 1e-4 gradient, 1e-3 Hessian
entries, all relative to the largest entry.
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FIXTURES = sorted(glob.glob(os.path.join(GOLD, 'illcond_rbf_*.npz')))


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def _model(g):
    from bayesian_optimization_amd import DeviceGP
    return DeviceGP(g['X_train'], g['y'], int(g['kind']), float(g['variance']), g['lengthscales'],
                    float(g['noise']), floor_mode=int(g['floor_mode']))


def test_fixtures_present():
    assert len(FIXTURES) >= 4, FIXTURES


@pytest.mark.parametrize('path', FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_gradient_against_extended_precision(cuda, path):
    g = np.load(path)
    dm = _model(g)
    X = g['X']
    value, grad, status, _ = [t.cpu().numpy() for t in dm.value_grad(X[None])]
    assert status[0] == 0
    assert abs(dm.fmin - float(g['fmin_mp'])) <= 1e-7 * abs(float(g['fmin_mp']))
    om = dm.omega(X[None])[0].cpu().numpy()
    e_om = _rel(om, g['omega_mp'])
    e_val = abs(value[0] - float(g['value'])) / abs(float(g['value']))
    e_grad = _rel(grad[0], g['grad_mp'])
    print('%s: device vs extended precision: omega %.1e value %.1e gradient %.1e | fp64 restatement: gradient %.1e'
          % (os.path.basename(path), e_om, e_val, e_grad, float(g['grad_fp64_err'])))
    assert e_val <= 1e-5                                   # task bar
    assert e_grad <= 1e-4                                  # task bar
    dm.close()


def test_hessian_against_extended_precision(cuda):
    paths = [p for p in FIXTURES if 'hess_mp' in np.load(p).files]
    assert paths
    for path in paths:
        g = np.load(path)
        dm = _model(g)
        H, st = dm.hessian(g['X'][None])
        assert int(st[0]) == 0
        H = H[0].cpu().numpy()
        e = _rel(H, g['hess_mp'])
        print('%s: device Hessian vs extended precision %.1e | fp64 restatement %.1e'
              % (os.path.basename(path), e, float(g['hess_fp64_err'])))
        assert e <= 1e-3                                   # task bar
        assert np.abs(H - H.T).max() <= 1e-9 * np.abs(H).max()
        dm.close()
