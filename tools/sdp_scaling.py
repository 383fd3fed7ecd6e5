"""Times oei_sdp_batched at several batch sizes: single-wave latency vs throughput under occupancy."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_optimization_amd import DeviceGP, sdp_batched
from oracle import hart6, gp as ogp

q = int(sys.argv[1]) if len(sys.argv) > 1 else 20
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
X = np.random.RandomState(1).uniform(-.5, .5, (8192, q, 6))
om_all = gp.omega(X)
for B in (64, 256, 512, 1024, 1536, 2048, 4096, 8192):
    om = om_all[:B].contiguous()
    for _ in range(2):
        out = sdp_batched(om, gp.fmin, want_Y=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = sdp_batched(om, gp.fmin, want_Y=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    it = out[4].cpu().numpy()
    print('B=%5d  %.3f ms  %.1f us/candidate-slot  newton %.2f sweeps %.2f' % (B, dt * 1e3, dt * 1e6 / max(1, B / 256), (it & 0xffff).mean(), (it >> 16).mean()))
