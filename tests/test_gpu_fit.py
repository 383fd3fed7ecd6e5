"""Hyper-parameter objective on the GPU against the CPU restatement, the restarted fit, and a short
end-to-end run of the driver (flags and output files of the reference's run.py)."""
import os
import pickle

import numpy as np
import pytest
import torch

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


@pytest.mark.parametrize('n,d,kind,fm,ard,free_noise,priors', [(40, 6, ogp.KERNEL_MATERN32, 1, False, False, False),
                                                                (150, 6, ogp.KERNEL_MATERN32, 2, False, True, True),
                                                                (70, 4, ogp.KERNEL_RBF, 0, True, False, True),
                                                                (200, 3, ogp.KERNEL_MATERN32, 1, True, True, False)])
def test_batched_objective_vs_oracle(cuda, n, d, kind, fm, ard, free_noise, priors):
    """R parameter vectors in one pass: value and gradient with respect to the log-parameters
    against the CPU restatement, item by item; an item whose covariance is not positive definite
    gives +inf and does not disturb its neighbours; a sub-batch gives the same bits."""
    import torch
    from bayesian_optimization_amd import hyperfit
    rs = np.random.RandomState(11)
    X = rs.uniform(-.5, .5, (n, d))
    if kind == ogp.KERNEL_RBF:
        X[1] = X[0]                                   # duplicate row: singular without noise
    y = hart6.normalize(np.sin(3 * X).sum(axis=1, keepdims=True))[:, 0]
    nls = d if ard else 1
    fixed = None if free_noise else 1e-4
    nt = 1 + nls + (1 if free_noise else 0)
    R = 7
    theta = rs.uniform(-1.0, 0.7, (R, nt))
    if free_noise:
        theta[:, -1] = rs.uniform(-9, -3, R)
    obj = hyperfit.BatchedObjective(X, y, kind, fm, ard, fixed, priors, max_batch=R)
    value, grad = [t.cpu().numpy() for t in obj(torch.as_tensor(theta, device=cuda))]
    for z in range(R):
        var, ls = np.exp(theta[z, 0]), np.exp(theta[z, 1:1 + nls])
        noise = np.exp(theta[z, -1]) if free_noise else fixed
        ref, gref = onlml.nlml(X, y, kind, var, ls, noise, floor_mode=fm)
        pv, dvar, dls = hyperfit._neg_log_prior(var, ls, priors)
        gt = np.concatenate([[(gref[0] + dvar) * var], (gref[1:1 + nls] + dls) * ls,
                             [gref[1 + nls] * noise] if free_noise else []])
        assert abs(value[z] - (ref + pv)) <= 1e-9 * (1 + abs(ref)), z
        assert np.abs(grad[z] - gt).max() <= 1e-7 * (1 + np.abs(gt).max()), z
        single = hyperfit.objective(theta[z], X, y, kind, fm, ard, fixed, priors)
        assert abs(single[0] - value[z]) <= 1e-10 * (1 + abs(value[z]))
    v2, g2 = obj(torch.as_tensor(theta[2:5], device=cuda))
    assert np.array_equal(v2.cpu().numpy(), value[2:5]) and np.array_equal(g2.cpu().numpy(), grad[2:5])
    if kind == ogp.KERNEL_RBF:
        obj0 = hyperfit.BatchedObjective(X, y, kind, fm, ard, 0.0, priors, max_batch=R)
        v0, g0 = obj0(torch.as_tensor(theta, device=cuda))
        # singular in exact arithmetic; rounding decides per item whether a pivot turns non-positive
        bad = torch.isinf(v0)
        assert bool(bad.any()) and float(g0[bad].abs().max()) == 0.0 and bool((v0[bad] > 0).all())
        assert bool(torch.isfinite(v0[~bad]).all())
        obj0.close()
    with pytest.raises(Exception):
        obj(torch.as_tensor(np.zeros((R + 1, nt)), device=cuda))
    obj.close()


def test_batched_fit_matches_sequential_fit(cuda):
    """All restarts in lock-step on the device against one scipy L-BFGS-B run per restart from the
    same random starts: the same best likelihood (the optimisers differ, the optimum does not)."""
    from bayesian_optimization_amd import hyperfit
    rs = np.random.RandomState(4)
    X = rs.uniform(-.5, .5, (110, 6))
    y = hart6.normalize(hart6.hart6_scaled(X))[:, 0]
    for kw in (dict(fixed_noise=True, ard=False, priors=False), dict(fixed_noise=False, ard=True, priors=True)):
        a = hyperfit.fit(X, y, ogp.KERNEL_MATERN32, 1.0, 1.0, 1e-6, floor_mode=1, restarts=8,
                         rng=np.random.RandomState(0), **kw)
        b = hyperfit.fit_sequential(X, y, ogp.KERNEL_MATERN32, 1.0, 1.0, 1e-6, floor_mode=1, restarts=8,
                                    rng=np.random.RandomState(0), **kw)
        assert abs(a['nlml'] - b['nlml']) <= 1e-5 * (1 + abs(b['nlml'])), (a, b)
        assert abs(a['variance'] - b['variance']) <= 2e-2 * b['variance']
        assert np.abs(a['lengthscales'] - b['lengthscales']).max() <= 2e-2 * b['lengthscales'].max()


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


def test_refactor_waits_for_evaluations_in_flight(cuda):
    """oei_model_refactor rewrites the factors that evaluations read.  A large evaluation enqueued
    on a non-default (non-blocking with respect to the null stream) torch stream right before a
    refactorisation must still see the old factors: same bits as an evaluation that was waited
    for, and the next evaluation sees the new ones."""
    from bayesian_optimization_amd import DeviceGP
    om = hart6.make_model(310)
    dm = DeviceGP(om.X, om.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
    Xd = torch.as_tensor(np.random.RandomState(4).uniform(-.5, .5, (4096, 20, 6)), device=cuda)
    ref_old = dm.value_grad(Xd)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        out = dm.value_grad(Xd)                 # asynchronous: about a millisecond of device work
    dm.refactor(1.3, [0.7], 1e-6)               # must not overtake it
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref_old[0]) and torch.equal(out[1], ref_old[1])
    new = dm.value_grad(Xd)
    fresh = DeviceGP(om.X, om.y, ogp.KERNEL_MATERN32, 1.3, [0.7], 1e-6, floor_mode=1)
    ref_new = fresh.value_grad(Xd)
    assert torch.equal(new[0], ref_new[0]) and torch.equal(new[1], ref_new[1])
