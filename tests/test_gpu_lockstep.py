"""Lock-step multistart descent on the GPU against the reference's sequential multistart
(bo.py:129-192 with scipy L-BFGS-B, solvers.py:10-21) from the same starts."""
import numpy as np
import pytest
import scipy.optimize
import torch

from oracle import gp as ogp, hart6

pytestmark = pytest.mark.gpu


class _Objective:
    bounds = np.array([[-0.5, 0.5]] * 6)

    def f(self, X):
        return hart6.hart6_scaled(X)


def _make_oei(n, q, nl_solver, opt_restarts):
    from bayesian_optimization_amd.methods.oei import OEI
    from bayesian_optimization_amd.kernels import SafeMatern32
    model = hart6.make_model(n)
    options = dict(objective=_Objective(), kernel=SafeMatern32(6, variance=1.0, lengthscales=0.5),
                   noise=1e-6, batch_size=q, initial_size=n, iterations=1, model_restarts=0,
                   opt_restarts=opt_restarts, normalize_Y=1, nl_solver=nl_solver, hessian=0,
                   job_name='test')
    bo = OEI(options)
    bo.X = model.X
    bo.Y = model.y[:, None]
    return bo, model


def test_lockstep_matches_sequential_lbfgsb(cuda):
    from bayesian_optimization_amd import DeviceGP, lockstep
    q, B = 3, 12
    model = hart6.make_model(25)
    gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
    starts = np.random.RandomState(7).uniform(-.5, .5, (B, q, 6))
    res, (bv, bi, bx) = lockstep.optimize_multistart(gp, starts, [-0.5] * 6, [0.5] * 6)
    fun = res.fun.cpu().numpy()
    x = res.x.cpu().numpy()
    assert res.converged.all() and (np.abs(x) <= 0.5).all()
    # every final point is a KKT point of the box-constrained problem (projected gradient ~ 0)
    v, g, st, _ = gp.value_grad(x.reshape(B, q, 6))
    assert np.allclose(v.cpu().numpy(), fun, rtol=1e-12)
    g = g.cpu().numpy().reshape(B, -1)
    pg = x - np.clip(x - g, -0.5, 0.5)
    assert np.abs(pg).max() < 1e-3          # runs may also stop on the relative-decrease rule, as scipy does

    def f1(z):
        vv, gg, _, _ = gp.value_grad(z.reshape(1, q, 6))
        return float(vv.item()), gg.cpu().numpy().ravel()
    seq = []
    for b in range(B):
        r = scipy.optimize.minimize(f1, starts[b].ravel(), jac=True, method='L-BFGS-B',
                                    bounds=[(-0.5, 0.5)] * (q * 6))
        seq.append(r.fun)
    seq = np.array(seq)
    # the acquisition surface is multimodal: single runs may land in different basins, the
    # multistart result (the quantity bo.py keeps) must agree, and most runs individually too
    assert float(bv) <= seq.min() + 1e-6 * abs(seq.min())
    assert (np.abs(fun - seq) <= 1e-4 * np.abs(seq)).mean() >= 0.5
    assert int(bi) == int(np.argmin(fun)) and np.array_equal(bx.cpu().numpy().ravel(), x[int(bi)])
    # lock-step saves calls: evaluations per run stay comparable, launches are shared
    assert res.nfev.max().item() < 400


def test_get_suggestion_lockstep(cuda):
    bo, model = _make_oei(25, 3, 'lockstep', 6)
    np.random.seed(5)
    X = bo.get_suggestion(3)
    assert X.shape == (3, 6) and (np.abs(X) <= 0.5 + 1e-12).all()
    np.random.seed(5)
    starts = [bo.random_sample(bo.bounds, 3) for _ in range(6)]
    best_start = min(bo.acquisition(s)[0][0] for s in starts)
    assert bo.acquisition(X)[0][0] <= best_start + 1e-12
    # same stream use as the sequential loop: the sequential solver from the same seed must not
    # beat the lock-step result by more than a different-basin margin
    bo2, _ = _make_oei(25, 3, 'bfgs', 6)
    np.random.seed(5)
    X2 = bo2.get_suggestion(3)
    assert bo.acquisition(X)[0][0] <= bo2.acquisition(X2)[0][0] + 0.05 * abs(bo2.acquisition(X2)[0][0])


def test_lockstep_large_batch_device_resident(cuda):
    """512 runs of q = 20 in lock-step: runs, history and line search stay on the device."""
    from bayesian_optimization_amd import DeviceGP, lockstep
    model = hart6.make_model(310)
    gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
    starts = torch.as_tensor(np.random.RandomState(9).uniform(-.5, .5, (512, 20, 6)), device=cuda)
    v0 = gp.value_grad(starts)[0]
    res, best = lockstep.optimize_multistart(gp, starts, [-0.5] * 6, [0.5] * 6, max_iter=30)
    assert res.x.device.type == 'cuda'
    assert bool((res.fun <= v0 + 1e-12).all())
    assert float(best[0]) == float(res.fun.min())


def test_warm_started_solve_matches_cold(cuda):
    """oei_value_grad_warm_batched: a solve started from the weights of a nearby evaluation
    (the role of sdp.py:77-121) returns the cold-start result to 1e-10 and needs fewer Newton
    iterations; weights outside the simplex or NaN fall back to the closed-form start."""
    from bayesian_optimization_amd import DeviceGP
    model = hart6.make_model(310)
    gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
    rs = np.random.RandomState(13)
    X = torch.as_tensor(rs.uniform(-.45, .45, (256, 20, 6)), device=cuda)
    v0, g0, st0, fl0, p0, it0 = gp.value_grad_warm(X, want_iters=True)
    cold = gp.value_grad(X)
    assert torch.equal(v0, cold[0]) and torch.equal(g0, cold[1])            # no start = the ordinary call
    X1 = X + 1e-3 * torch.as_tensor(rs.randn(256, 20, 6), device=cuda)       # a line-search-sized move
    vc, gc, stc, _, pc, itc = gp.value_grad_warm(X1, want_iters=True)
    vw, gw, stw, _, pw, itw = gp.value_grad_warm(X1, p_init=p0, want_iters=True)
    assert int((stc != 0).sum()) == 0 and int((stw != 0).sum()) == 0
    assert float(((vw - vc).abs() / vc.abs()).max()) <= 1e-10
    assert float((gw - gc).abs().max() / gc.abs().max()) <= 2e-9         # task bar: 1e-4
    assert float((pw - pc).abs().max()) <= 1e-7                          # states (weight ratios)
    nc, nw = (itc & 0xffff).double().mean().item(), (itw & 0xffff).double().mean().item()
    assert nc >= 3.0 and nw <= 2.3, (nc, nw)
    # unusable starts: NaN, outside (0, 1), sum >= 1
    bad = p0.clone()
    bad[0] = float('nan')
    bad[1, 3] = -0.1
    bad[2] = 0.06
    vb = gp.value_grad_warm(X, p_init=bad)[0]
    assert float(((vb - v0).abs() / v0.abs()).max()) <= 1e-10


def test_lockstep_with_warm_start(cuda):
    """The lock-step optimiser can thread every run's solver state through its evaluations: same
    quality of optima as with cold starts, and no more Newton iterations per evaluation (the
    quasi-Newton steps are mostly too long for the state to beat the closed-form start; the gain
    shows for short moves, see test_warm_started_solve_matches_cold)."""
    from bayesian_optimization_amd import DeviceGP, lockstep
    model = hart6.make_model(310)
    gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
    starts = torch.as_tensor(np.random.RandomState(9).uniform(-.5, .5, (256, 20, 6)), device=cuda)
    sw, sc = {}, {}
    rw, bw = lockstep.optimize_multistart(gp, starts, [-0.5] * 6, [0.5] * 6, max_iter=30, warm_start=True, stats=sw)
    rc, bc = lockstep.optimize_multistart(gp, starts, [-0.5] * 6, [0.5] * 6, max_iter=30, warm_start=False, stats=sc)
    assert sw['newton'] / sw['evals'] <= 4.0          # cold: 3.2; early, long steps cost a little, late ones gain
    # the iterates of the two runs differ at rounding level and the descent amplifies that:
    # compare what the multistart keeps, the best value
    assert abs(float(bw[0]) - float(bc[0])) <= 2e-2 * abs(float(bc[0]))
    assert float((rw.fun - rc.fun).abs().median()) <= 1e-2


def test_lockstep_newton_consumes_batched_hessian(cuda):
    """`method='newton'`: all restarts take projected Newton steps on oei_hessian_batched at once
    (the role of KNITRO's SQP in solvers.py:56-69).  From the same starts it reaches optima as
    good as the first-order lock-step (the surface is multimodal: compared are the best and the
    average over the runs) in fewer value+gradient evaluations."""
    from bayesian_optimization_amd import DeviceGP, lockstep
    q, B = 5, 24
    model = hart6.make_model(60)
    gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
    starts = np.random.RandomState(11).uniform(-.5, .5, (B, q, 6))
    s1, s2 = {}, {}
    r1, b1 = lockstep.optimize_multistart(gp, starts, [-0.5] * 6, [0.5] * 6, stats=s1)
    r2, b2 = lockstep.optimize_multistart(gp, starts, [-0.5] * 6, [0.5] * 6, method='newton', stats=s2)
    m1, m2 = float(r1.fun.mean()), float(r2.fun.mean())
    assert float(b2[0]) <= float(b1[0]) + 2e-2 * abs(float(b1[0])), (float(b1[0]), float(b2[0]))
    assert m2 <= m1 + 2e-2 * abs(m1), (m1, m2)
    assert s2['evals'] < s1['evals'], (s1, s2)
    x = r2.x.cpu().numpy()
    assert (np.abs(x) <= 0.5 + 1e-12).all()
    v, g, st, _ = gp.value_grad(x.reshape(B, q, 6))
    assert np.allclose(v.cpu().numpy(), r2.fun.cpu().numpy(), rtol=1e-12)
    pg = x - np.clip(x - g.cpu().numpy().reshape(B, -1), -0.5, 0.5)
    assert np.abs(pg).max() < 1e-3


def test_newton_path_zeroes_the_hessian_of_duplicate_candidates(cuda):
    """acquisition_hessian returns zeros when `project` would remove a point (oei.py:53-55).  The
    batched Hessian adapter of the Newton path does the same from the device's duplicate flags,
    and a multistart whose starts include a duplicate of a training point still ends finite."""
    from bayesian_optimization_amd import DeviceGP, lockstep
    q = 3
    model = hart6.make_model(25)
    gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
    starts = np.random.RandomState(3).uniform(-.5, .5, (6, q, 6))
    starts[2, 1] = model.X[4] + 1e-4               # within 1e-2 lengthscales of a training input
    starts[5, 2] = starts[5, 0] + 1e-6             # two equal points
    hess = lockstep.acquisition_hess_fun(gp, q)
    H = hess(torch.as_tensor(starts, device=gp.device).reshape(6, -1))
    zero = (H == 0).all(dim=2).all(dim=1).cpu().numpy()
    assert zero.tolist() == [False, False, True, False, False, True]
    raw, st = gp.hessian(starts)
    assert int((st != 0).sum()) == 0 and torch.equal(H[0], raw[0])

    bo, _ = _make_oei(25, q, 'lockstep', 6)
    res, best = lockstep.optimize_multistart(
        gp, starts, [-0.5] * 6, [0.5] * 6, method='newton',
        project_fallback=lambda x: (lambda r: (r[0][0], r[1]))(bo.acquisition(x)))
    assert bool(torch.isfinite(res.fun).all()) and bool(torch.isfinite(best[0]))
