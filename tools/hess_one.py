"""Diagnostic: a few oei_hessian_batched calls at one batch size (for counter passes)."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
X = torch.as_tensor(np.random.RandomState(1).uniform(-.5, .5, (B, 20, 6)), device='cuda')
for _ in range(3):
    H, st = gp.hessian(X)
torch.cuda.synchronize()
print('ok', int((st != 0).sum()))
