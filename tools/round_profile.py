"""Diagnostic: one 20-start multistart (library round loop) for a kernel trace: wall time and rounds,
so that the per-kernel durations of `rocprofv3 --kernel-trace --stats` can be set against them."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP, lockstep
q, d = 20, 6
method = sys.argv[1] if len(sys.argv) > 1 else 'lbfgs'
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
X0 = torch.as_tensor(np.random.RandomState(3).uniform(-.5, .5, (20, q * d)), device=gp.device)
lo = torch.full((q * d,), -0.5, dtype=torch.float64, device=gp.device)
lockstep.minimize_acquisition_rounds(gp, q, X0[:4], lo, -lo, method=method, max_iter=3)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = lockstep.minimize_acquisition_rounds(gp, q, X0, lo, -lo, method=method)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('%s: %.3f s, max evaluations of a run %d, converged %d' % (method, dt, int(res.nfev.max()), int(res.converged.sum())))
