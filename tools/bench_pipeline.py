"""Diagnostic: wall time of the value+gradient call (B = 4096, q = 20, n = 310) for several
splits of the batch into pieces that alternate between side streams (OEI_PIPELINE=chunks,streams
is read by the library at every call), and a bitwise comparison of the results with the
single-stream call."""
import os, sys, time, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
X = torch.as_tensor(np.random.RandomState(1).uniform(-.5, .5, (B, 20, 6)), device='cuda')

def run(cfg, reps=30):
    if cfg: os.environ['OEI_PIPELINE'] = cfg
    else: os.environ.pop('OEI_PIPELINE', None)
    for _ in range(3):
        out = gp.value_grad(X, want_flags=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = gp.value_grad(X, want_flags=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    v, g, st, _ = out
    h = hashlib.sha1(v.cpu().numpy().tobytes() + g.cpu().numpy().tobytes() + st.cpu().numpy().tobytes()).hexdigest()[:12]
    return ms, h

base = None
for cfg in (sys.argv[2].split(':') if len(sys.argv) > 2 else ['', '2,2', '', '2,2', '4,2', '']):
    cfg = '' if cfg == 'off' else cfg
    ms, h = run(cfg)
    base = base or h
    print('%-6s %.4f ms  %s %s' % (cfg or 'off', ms, h, 'same' if h == base else 'DIFFERENT'), flush=True)
