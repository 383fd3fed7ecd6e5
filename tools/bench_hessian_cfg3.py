"""Diagnostic: Hessian throughput at the largest configuration (n=1000, q=50, d=10)."""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

model = hart6.make_model(1000, d=10, lengthscale=0.8)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.8], 1e-6, floor_mode=1)
for B in (1, 16, 256):
    X = torch.as_tensor(np.random.RandomState(2).uniform(-.5, .5, (B, 50, 10)), device='cuda')
    gp.hessian(X); torch.cuda.synchronize()
    t0 = time.perf_counter()
    H, st = gp.hessian(X)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('B=%4d  %.1f ms per call  %.1f Hessians/s  unsolved %d' % (B, dt * 1e3, B / dt, int((st != 0).sum())))
