"""Profiling driver: a few value+gradient calls on the benchmark workload (q = 20, n = 310, B = 4096)
and nothing else, for rocprofv3 counter passes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
X = torch.as_tensor(np.random.RandomState(1).uniform(-.5, .5, (B, 20, 6)), device='cuda')
for _ in range(4):
    gp.value_grad(X, want_flags=False)
torch.cuda.synchronize()
