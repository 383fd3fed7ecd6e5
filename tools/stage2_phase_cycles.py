"""Diagnostic (timing build only, see tools/sdp_phase_cycles.sh): cycles one wave of k_stage2 spends in
the posterior moments, the SDP solve, the primal matrix and the outputs + Z, at the benchmark's shape
(or, with a second argument `cfg3`, one workgroup of the wide kernel at q = 50, d = 10, n = 1000)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_optimization_amd import DeviceGP
from oracle import hart6, gp as ogp
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
if len(sys.argv) > 2 and sys.argv[2] == 'cfg3':          # the largest shape: q = 50, d = 10, n = 1000 (wide path)
    model = hart6.make_model(1000, d=10, lengthscale=0.8)
    gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.8], 1e-6, floor_mode=1)
    X = np.random.RandomState(2).uniform(-.5, .5, (B, 50, 10))
else:
    model = hart6.make_model(310)
    gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
    X = np.random.RandomState(1).uniform(-.5, .5, (B, 20, 6))
gp.set_pipelining(False)
for _ in range(3):
    out = gp.value_grad_warm(X)
c = out[4].cpu().numpy()
names = ['moments', 'solve', 'primal', 'outputs + Z', ' eval set-up', ' jacobi', ' eval rest', ' newton dir', '  hess items', '  cholesky']
for nm, v in zip(names, c[:, :10].mean(axis=0)):
    print('%-14s %9.0f cycles' % (nm, v))
print('moments: after the row loop (store, kernel entries) %9.0f cycles' % c[:, 12].mean())
print('kernel total   %9.0f cycles per wave' % c[:, :4].sum(axis=1).mean())
