"""Diagnostic: oei_hessian_batched at one batch size (q = 20, n = 310), for kernel-trace profiles."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
X = torch.as_tensor(np.random.RandomState(1).uniform(-.5, .5, (B, 20, 6)), device='cuda')
gp.hessian(X); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    H, st = gp.hessian(X)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print('B=%4d  %.3f ms per call  %.1f Hessians/s  unsolved %d' % (B, dt * 1e3, B / dt, int((st != 0).sum())))
