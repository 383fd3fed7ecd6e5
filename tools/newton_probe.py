"""Diagnostic: the device projected-Newton multistart alone (no L-BFGS polish): stop reasons, time and
iteration counts against its iteration cap."""
import sys, time, collections
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP, lockstep

q, d = 20, 6
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
NAMES = {0: 'running', 1: 'conv_pg', 2: 'conv_f', 3: 'ls_failed', 4: 'stalled', 5: 'bad_start', 6: 'iter_limit'}
for B, seed in ((20, 3), (512, 4)):
    starts = np.random.RandomState(seed).uniform(-.5, .5, (B, q, d))
    X0 = torch.as_tensor(starts, device=gp.device).reshape(B, q * d)
    lo = torch.full((q * d,), -0.5, dtype=torch.float64, device=gp.device)
    hi = -lo
    fun = lockstep.acquisition_fun(gp, q)
    hess = lockstep.acquisition_hess_fun(gp, q)
    lockstep.minimize_lockstep_newton_device(fun, hess, X0[:4], lo, hi, max_iter=2)
    for max_iter in (60, 200, -200):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if max_iter < 0:          # the round loop inside the library
            max_iter = -max_iter
            res = lockstep.minimize_acquisition_rounds(gp, q, X0, lo, hi, method='newton', max_iter=max_iter)
        else:
            res = lockstep.minimize_lockstep_newton_device(fun, hess, X0, lo, hi, max_iter=max_iter)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        pg = (res.x - torch.clamp(res.x - res.grad, -0.5, 0.5)).abs().amax(dim=1)
        print('B=%d newton max_iter=%d: %.3f s, nit max %d mean %.1f, nfev max %d, converged %d, mean f %.6f best %.6f, pg median %.1e max %.1e, %s'
              % (B, max_iter, dt, int(res.nit.max()), float(res.nit.double().mean()), int(res.nfev.max()), int(res.converged.sum()),
                 float(res.fun.mean()), float(res.fun.min()), float(pg.median()), float(pg.max()),
                 dict(collections.Counter(res.message))), flush=True)
