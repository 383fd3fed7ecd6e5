"""Hyper-parameter objective on the GPU against the CPU restatement, the restarted fit, and a short
end-to-end run of the driver (flags and output files of the reference's run.py)."""
import os
import pickle

import numpy as np
import pytest

from oracle import gp as ogp, hart6, nlml as onlml

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,d,kind,fm,ard', [(40, 6, ogp.KERNEL_MATERN32, 1, False),
                                             (150, 6, ogp.KERNEL_MATERN32, 2, False),
                                             (70, 4, ogp.KERNEL_RBF, 0, False),
                                             (70, 3, ogp.KERNEL_MATERN32, 1, True)])
def test_nlml_and_gradient_vs_oracle(cuda, n, d, kind, fm, ard):
    from bayesian_optimization_amd import DeviceGP
    rs = np.random.RandomState(3)
    X = rs.uniform(-.5, .5, (n, d))
    y = hart6.normalize(np.sin(3 * X).sum(axis=1, keepdims=True))[:, 0]
    var, noise = 1.4, 1e-3
    ls = rs.uniform(0.3, 0.8, d) if ard else np.array([0.45])
    gp = DeviceGP(X, y, kind, var, ls, noise, floor_mode=fm)
    val, grad = gp.nlml(ard=ard)
    ref, gref = onlml.nlml(X, y, kind, var, ls, noise, floor_mode=fm)
    assert abs(val - ref) <= 1e-9 * (1 + abs(ref))
    assert np.abs(grad - gref).max() <= 1e-7 * (1 + np.abs(gref).max())


def test_refactor_in_place_equals_a_new_model(cuda):
    """New hyper-parameters through oei_model_refactor give the bits a freshly created model gives
    (factor, fmin, likelihood and its gradient, acquisition value and gradient), also after a trial
    point whose covariance is not positive definite."""
    import torch
    from bayesian_optimization_amd import DeviceGP, _lib
    rs = np.random.RandomState(8)
    X = rs.uniform(-.5, .5, (130, 6))
    y = hart6.normalize(hart6.hart6_scaled(X))[:, 0]
    Xc = rs.uniform(-.5, .5, (5, 8, 6))
    gp = DeviceGP(X, y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
    gp.nlml()
    dup = np.vstack([X[:1], X])                                     # duplicate row, zero noise, RBF: singular
    for var, ls, noise in ((2.3, [0.31], 1e-4), (0.7, [0.9], 1e-2), (1.0, [0.5], 1e-6)):
        gp.refactor(var, ls, noise)
        new = DeviceGP(X, y, ogp.KERNEL_MATERN32, var, ls, noise, floor_mode=1)
        assert gp.fmin == new.fmin and np.array_equal(gp.factor(), new.factor())
        a, b = gp.nlml(), new.nlml()
        assert a[0] == b[0] and np.array_equal(a[1], b[1])
        for s, t in zip(gp.value_grad(Xc), new.value_grad(Xc)):
            assert torch.equal(s, t)
        new.close()
    sing = DeviceGP(dup, np.concatenate([y[:1], y]), ogp.KERNEL_RBF, 1.0, [0.5], 1e-3, floor_mode=0)
    with pytest.raises(_lib.OeiError):
        sing.refactor(1.0, [0.5], 0.0)
    sing.refactor(1.0, [0.5], 1e-3)                                 # the handle recovers
    ref = onlml.nlml(dup, np.concatenate([y[:1], y]), ogp.KERNEL_RBF, 1.0, [0.5], 1e-3, floor_mode=0, want_grad=False)[0]
    assert abs(sing.nlml(want_grad=False)[0] - ref) <= 1e-9 * (1 + abs(ref))
    with pytest.raises(_lib.OeiError):
        gp.refactor(-1.0, [0.5], 1e-6)
    sing.close()
    gp.close()


def test_fit_recovers_better_likelihood(cuda):
    from bayesian_optimization_amd import hyperfit
    rs = np.random.RandomState(4)
    X = rs.uniform(-.5, .5, (60, 6))
    y = hart6.normalize(hart6.hart6_scaled(X))[:, 0]
    start = onlml.nlml(X, y, ogp.KERNEL_MATERN32, 1.0, 1.0, 1e-6, floor_mode=1, want_grad=False)[0]
    res = hyperfit.fit(X, y, ogp.KERNEL_MATERN32, 1.0, 1.0, 1e-6, floor_mode=1, restarts=4,
                       rng=np.random.RandomState(0))
    assert res['nlml'] <= start + 1e-9 and res['noise'] == 1e-6
    # stationarity in the log-parameters at the returned point (checked with the oracle's gradient)
    _, g = onlml.nlml(X, y, ogp.KERNEL_MATERN32, res['variance'], res['lengthscales'], 1e-6, floor_mode=1)
    glog = g[:2] * np.array([res['variance'], res['lengthscales'][0]])
    assert np.abs(glog).max() < 1e-2 * (1 + abs(res['nlml']))
    ref = onlml.nlml(X, y, ogp.KERNEL_MATERN32, res['variance'], res['lengthscales'], 1e-6,
                     floor_mode=1, want_grad=False)[0]
    assert abs(ref - res['nlml']) <= 1e-8 * (1 + abs(ref))


def test_driver_end_to_end(cuda, tmp_path, monkeypatch):
    from bayesian_optimization_amd import run
    monkeypatch.chdir(tmp_path)
    args = run.create_parser().parse_args(['--seed=7', '--batch_size=3', '--initial_size=8',
                                           '--iterations=2', '--noise=1e-6', '--model_restarts=2',
                                           '--opt_restarts=4'])
    run.main(args)
    out = np.load(os.path.join('out', 'hart6_OEI', '7.npz'))
    X, Y = out['X'], out['Y']
    assert X.shape == (8 + 2 * 3, 6) and Y.shape == (14, 1) and (np.abs(X) <= 0.5 + 1e-12).all()
    assert np.allclose(Y, hart6.hart6_scaled(X))
    assert abs(np.loadtxt(os.path.join('out', 'hart6_OEI', 'fmin.txt')) + 3.32237) < 1e-9
    with open(os.path.join('out', 'hart6_OEI', 'arguments.pkl'), 'rb') as fh:
        assert pickle.load(fh).seed == 7
    for name in ('evals', 'model', 'opt'):
        assert os.path.getsize(os.path.join('log', 'hart6_OEI', '7', name + '.log')) > 0
    # same seed, same run
    run.main(args)
    assert np.array_equal(np.load(os.path.join('out', 'hart6_OEI', '7.npz'))['X'], X)
