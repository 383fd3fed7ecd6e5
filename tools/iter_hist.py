"""Diagnostic: distribution of Newton iterations and Jacobi sweeps over the benchmark candidates."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP, sdp_batched

model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, model.kernel.variance, model.kernel.lengthscales, model.noise, floor_mode=model.kernel.floor_mode)
X = torch.as_tensor(np.random.RandomState(1).uniform(-.5, .5, (4096, 20, 6)), device='cuda')
om = gp.omega(X)
_, _, _, st, it = sdp_batched(om, gp.fmin, want_Y=False)
it = it.cpu().numpy()
nw, sw = it & 0xffff, it >> 16
print('newton', np.bincount(nw))
print('sweeps', np.bincount(sw))
