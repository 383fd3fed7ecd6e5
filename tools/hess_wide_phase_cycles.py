"""Diagnostic: cycle stamps of thread 0 of k_hess_wide for candidate 0 (library built with
-DOEI_HESS_TIMING, which writes them over the first entries of that candidate's Hessian).
    OEI_LIB_OVERRIDE=<timing build> python tools/hess_wide_phase_cycles.py [B]"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = hart6.make_model(1000, d=10, lengthscale=0.8)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.8], 1e-6, floor_mode=1)
X = torch.as_tensor(np.random.RandomState(2).uniform(-.5, .5, (B, 50, 10)), device='cuda')
for _ in range(2):
    H, st = gp.hessian(X)
torch.cuda.synchronize()
c = H[0].flatten()[:16].cpu().numpy()
names = ['ZS pass', 'M1 = ZS diag(sq) U^T, t2', 'rhs, two solves', 'rd, U^T diag U', 'Z pass', 'Y = U Z, t2', 'dc, w1, w2',
         'dN mu, contraction']
for n_, v in zip(names, c[:8]):
    print('%-26s %10.0f cycles (all directions of the workgroup)' % (n_, v))
print('directions in all          %10.0f' % c[10])
print('prologue                   %10.0f  (pre-pass + tables %.0f, solver state %.0f, rest %.0f)' % (c[11], c[8], c[9], c[11] - c[8] - c[9]))
print('pre-pass %.0f, tables %.0f, covariance + self-covariance blocks %.0f; state copies + L^-1 by wave 0 %.0f' % (c[12], c[13], c[8] - c[12] - c[13], c[14]))
