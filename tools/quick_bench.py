"""Diagnostic: stage times of the value+gradient path (HIP events), iteration counters and a
bitwise checksum of the results, B = 4096, q = 20, n = 310."""
import os, sys, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
X = torch.as_tensor(np.random.RandomState(1).uniform(-.5, .5, (B, 20, 6)), device='cuda')
for _ in range(3):
    out = gp.value_grad(X, want_flags=False)
gp.set_profiling(True)
st = np.zeros(5)
for _ in range(10):
    gp.value_grad(X, want_flags=False)
    st += np.array(gp.stage_times_ms())
gp.set_profiling(False)
st /= 10
v, g, status, _ = out
_, _, _, _, _, it = gp.value_grad_warm(X, want_iters=True)
print('stages ms: gram %.4f gemm1 %.4f stage2 %.4f gemm2 %.4f fin %.4f  total %.4f' % (*st, st.sum()))
print('newton %.3f sweeps %.3f unsolved %d' % ((it & 0xffff).double().mean(), (it >> 16).double().mean(), int((status != 0).sum())))
print('checksum value %s grad %s' % (hashlib.sha1(v.cpu().numpy().tobytes()).hexdigest()[:12], hashlib.sha1(g.cpu().numpy().tobytes()).hexdigest()[:12]))
print('value[:3]', v[:3].tolist())

if len(sys.argv) > 2:      # accuracy against the CPU restatement on a few candidates
    from oracle import reduced_newton, sdp_ref
    from bayesian_optimization_amd import sdp_batched
    Xh = X[:int(sys.argv[2])].cpu().numpy()
    omg = gp.omega(Xh)
    v2, M, Y, st2, _ = [t.cpu().numpy() for t in sdp_batched(omg, gp.fmin)]
    ev = eg = ec = 0.0
    for b in range(Xh.shape[0]):
        o = omg[b].cpu().numpy()
        cert = sdp_ref.kkt_certificate(o, gp.fmin, M[b], list(Y[b]))
        ec = max(ec, max(x for k, x in cert.items() if k not in ('pobj', 'dobj')))
        obj = reduced_newton.solve(o, gp.fmin)[0]
        ev = max(ev, abs(float(v[b]) - obj) / abs(obj))
        _, g0 = model.inner_and_grad(Xh[b], M[b])
        eg = max(eg, np.abs(g[b].cpu().numpy() - g0).max() / np.abs(g0).max())
    print('accuracy on %d candidates: value %.1e gradient %.1e KKT %.1e' % (Xh.shape[0], ev, eg, ec))
