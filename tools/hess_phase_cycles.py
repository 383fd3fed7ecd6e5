"""Diagnostic: cycle stamps of one wave of k_hess_dirs (library built with -DOEI_HESS_TIMING, which
writes them over the first entries of candidate 0's Hessian).
    OEI_LIB_OVERRIDE=tools/_variants/hesstiming.so python tools/hess_phase_cycles.py"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
X = torch.as_tensor(np.random.RandomState(1).uniform(-.5, .5, (2048, 20, 6)), device='cuda')
for _ in range(3):
    H, st = gp.hessian(X)
torch.cuda.synchronize()
c = H[0].flatten()[:24].cpu().numpy()
names = ['ZS pass', 'M1 = ZS Rs^T, t2', 'rhs, two solves', 'rd, R^T dD R', 'Z pass', 'Y = U Z, t2', 'dc, w1, w2', 'dN mu, contraction']
for n_, v in zip(names, c[:8]):
    print('%-24s %9.0f cycles (10 directions of wave 0)' % (n_, v))
print('directions in all        %9.0f' % c[10])
print('prologue                 %9.0f  (loads + diagonal-block pre-pass %.0f, tables + L^-1 %.0f, table transform %.0f)' % (c[11], c[8], c[9], c[11] - c[8] - c[9]))
print('initial loads %.0f; pre-pass: staging of X %.0f, kernel derivatives %.0f, matrix instructions %.0f, tile to Dg %.0f' % (c[20], c[16], c[17], c[18], c[19]))
