"""Diagnostic: calls the batched entry points a few times so that a kernel trace shows each
kernel's duration at the benchmark size (B=4096, q=20, n=310)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

B, q = 4096, 20
model = hart6.make_model(310)
dev = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, model.kernel.variance,
               model.kernel.lengthscales, model.noise, floor_mode=model.kernel.floor_mode)
X = torch.as_tensor(np.random.RandomState(1).uniform(-0.5, 0.5, (B, q, 6)), device='cuda').contiguous()
for _ in range(3):
    dev.omega(X)
    dev.value_grad(X)
torch.cuda.synchronize()
print('done')
