"""Diagnostic: the device L-BFGS multistart at the batch sizes of a BO loop (20 starts) and of the
benchmark (2048 starts): time, rounds, converged runs and stop reasons against the iteration cap."""
import sys, time, collections
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP, lockstep

q = 20
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
for B, seed in ((20, 3), (2048, 4)):
    starts = np.random.RandomState(seed).uniform(-.5, .5, (B, q, 6))
    lockstep.optimize_multistart(gp, starts[:8], [-.5] * 6, [.5] * 6, max_iter=3)
    for max_iter in [int(a) for a in sys.argv[1:] if a.isdigit()] or [200, 1000, 3000]:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res, best = lockstep.optimize_multistart(gp, starts, [-.5] * 6, [.5] * 6, max_iter=max_iter, **({"stats": {}} if "--python" in sys.argv else {}))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        hist = collections.Counter(res.message)
        print('B=%d max_iter=%d: %.3f s, rounds %d, evaluations %d, converged %d, best %.6f, mean %.6f, %s'
              % (B, max_iter, dt, int(res.nfev.max()), int(res.nfev.sum()), int(res.converged.sum()), float(best[0]),
                 float(res.fun.mean()), dict(hist)), flush=True)
