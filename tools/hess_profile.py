"""Diagnostic: 2048 Hessians per call (q = 20, d = 6, n = 310), a dozen calls, for a kernel trace."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
X = torch.as_tensor(np.random.RandomState(1).uniform(-.5, .5, (2048, 20, 6)), device='cuda')
for _ in range(5):
    gp.hessian(X)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    H, st = gp.hessian(X)
torch.cuda.synchronize()
print('%.3f ms per call' % ((time.perf_counter() - t0) / 10 * 1e3))
