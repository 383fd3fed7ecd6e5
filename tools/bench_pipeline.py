"""Diagnostic: wall time of the value+gradient call (B = 4096, q = 20, n = 310) for several
splits of the batch into pieces that alternate between the handle's two side streams
(DeviceGP.set_pipelining: 0 one stream, 1 the library's rule, n >= 2 pieces), and a bitwise
comparison of the results with the single-stream call."""
import os, sys, time, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
X = torch.as_tensor(np.random.RandomState(1).uniform(-.5, .5, (B, 20, 6)), device='cuda')

from bayesian_optimization_amd import multistart
STEP = os.environ.get('BENCH_STEP') == '1'       # time the benchmark's whole step instead of the bare call


def run(cfg, reps=30):
    gp.set_pipelining(int(cfg))
    call = (lambda: multistart.evaluate_and_select(gp, X)) if STEP else (lambda: gp.value_grad(X, want_flags=False))
    for _ in range(3):
        out = call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = call()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    v, g, st = out[0], out[1], out[2]
    h = hashlib.sha1(v.cpu().numpy().tobytes() + g.cpu().numpy().tobytes() + st.cpu().numpy().tobytes()).hexdigest()[:12]
    return ms, h

base = None
for cfg in (sys.argv[2].split(':') if len(sys.argv) > 2 else ['0', '2', '0', '1', '3', '4', '0', '2']):
    ms, h = run(cfg)
    base = base or h
    print('pieces %-3s %.4f ms  %s %s' % (cfg, ms, h, 'same' if h == base else 'DIFFERENT'), flush=True)
