"""Latency of oei_value_grad_batched at small batches (q = 20, n = 310): wall time per call and the
stage split from the library's events."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
X = torch.as_tensor(np.random.RandomState(1).uniform(-.5, .5, (4096, 20, 6)), device='cuda')
for B in (1, 20, 64, 256, 1024, 4096):
    gp.value_grad(X[:B]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        gp.value_grad(X[:B])
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    gp.set_profiling(True)
    st = np.zeros(5)
    for _ in range(5):
        gp.value_grad(X[:B]); st += np.array(gp.stage_times_ms())
    gp.set_profiling(False)
    print('B=%4d  %.3f ms per synchronous call; stages (ms) gram %.3f gemm %.3f moments+sdp %.3f gemm+contraction %.3f finish %.3f'
          % ((B, dt * 1e3) + tuple(st / 5)), flush=True)
