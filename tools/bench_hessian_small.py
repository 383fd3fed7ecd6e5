"""Latency of oei_hessian_batched at the batch sizes a BO loop produces (q = 20, n = 310)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
X = torch.as_tensor(np.random.RandomState(1).uniform(-.5, .5, (2048, 20, 6)), device='cuda')
for B in (1, 4, 20, 64, 128, 256, 2048):
    gp.hessian(X[:B]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gp.hessian(X[:B])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print('B=%4d  %.3f ms per call  %.0f Hessians/s' % (B, dt * 1e3, B / dt), flush=True)
