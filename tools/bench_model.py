"""Diagnostic: one-time model factorisation (Gram of the training set, blocked Cholesky, triangular
inverse, L^-1 y, fmin) -- wall time of oei_model_create and the FLOP/s of its dense part
(n^3/3 for the Cholesky + n^3/3 for the triangular inverse)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

for n, d, ls in [(310, 6, 0.5), (1000, 10, 0.8), (4000, 10, 0.8)]:
    model = hart6.make_model(n, d=d, lengthscale=ls)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [ls], 1e-6, floor_mode=1)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        gp.close()
    t = min(ts[1:])
    flops = 2.0 * n ** 3 / 3.0
    print('n=%5d d=%2d  model create %.2f ms  (Cholesky + triangular inverse %.2f GFLOP -> %.2f TFLOP/s over the whole call)'
          % (n, d, t * 1e3, flops / 1e9, flops / t / 1e12))
