import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_optimization_amd import DeviceGP, sdp_batched
from oracle import hart6, gp as ogp
q = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
model = hart6.make_model(310) if q <= 20 else hart6.make_model(310, d=10, lengthscale=0.8)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, model.kernel.lengthscales, 1e-6, floor_mode=1)
X = np.random.RandomState(1).uniform(-.5, .5, (B, q, model.X.shape[1]))
om = gp.omega(X)
value, M, _, st, it = sdp_batched(om, gp.fmin, want_Y=False)
cyc = M.reshape(B, -1)[:, :8].cpu().numpy()
it = it.cpu().numpy()
print('B=%d q=%d  newton %.2f sweeps %.2f' % (B, q, (it & 0xffff).mean(), (it >> 16).mean()))
for name, c in zip(['eval set-up', 'jacobi', 'eval rest', 'newton dir', ' hess items', ' cholesky', ' tri solves', '-'], cyc.mean(axis=0)):
    print('%-12s %10.0f cycles' % (name, c))
print('total        %10.0f cycles' % cyc[:, :4].sum(axis=1).mean())
