"""Diagnostic: value+gradient throughput at the largest configuration (n=1000, q=50, d=10)."""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

model = hart6.make_model(1000, d=10, lengthscale=0.8)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.8], 1e-6, floor_mode=1)
gp.set_profiling(True)
for B in (64, 512, 2048):
    X = torch.as_tensor(np.random.RandomState(2).uniform(-.5, .5, (B, 50, 10)), device='cuda')
    gp.value_grad(X); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        v, g, st, fl = gp.value_grad(X)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print('B=%5d  %.2f ms  %.0f evals/s  unsolved %d  stages %s' % (B, dt * 1e3, B / dt, int((st != 0).sum()),
          np.round(gp.stage_times_ms(), 2)))
from bayesian_optimization_amd import sdp_batched
omg = gp.omega(X[:256])
t0 = time.perf_counter()
_, _, _, st2, iters = sdp_batched(omg, gp.fmin, want_Y=False)
torch.cuda.synchronize()
print('sdp_batched 256: %.2f ms' % ((time.perf_counter() - t0) * 1e3))
iters = iters.cpu().numpy()
print('newton mean %.2f max %d  sweeps mean %.2f max %d' % ((iters & 0xffff).mean(), (iters & 0xffff).max(), (iters >> 16).mean(), (iters >> 16).max()))
