"""Diagnostic: Newton iterations of the SDP solve from a cold start and from the weights of a
nearby evaluation, as a function of the distance between the two candidates."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP

model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
rs = np.random.RandomState(13)
X = torch.as_tensor(rs.uniform(-.45, .45, (512, 20, 6)), device='cuda')
D = torch.as_tensor(rs.randn(512, 20, 6), device='cuda')
v0, g0, st0, fl0, p0, it0 = gp.value_grad_warm(X, want_iters=True)
print('p range', float(p0.min()), float(p0.max()), 'cold newton', (it0 & 0xffff).double().mean().item(), 'sweeps', (it0 >> 16).double().mean().item())
for eps in (0.0, 1e-5, 1e-4, 1e-3, 1e-2, 3e-2, 1e-1):
    X1 = torch.clamp(X + eps * D, -0.5, 0.5)
    vc, gc, stc, _, pc, itc = gp.value_grad_warm(X1, want_iters=True)
    vw, gw, stw, _, pw, itw = gp.value_grad_warm(X1, p_init=p0, want_iters=True)
    nw = (itw & 0xffff).double()
    print('eps %.0e cold %.2f/%.1f warm %.2f/%.1f  max warm %d  hist %s  dv %.1e dg %.1e bad %d' % (
        eps, (itc & 0xffff).double().mean(), (itc >> 16).double().mean(), nw.mean(), (itw >> 16).double().mean(),
        int(nw.max()), np.bincount(nw.cpu().numpy().astype(int)).tolist(),
        float(((vw - vc).abs() / vc.abs()).max()), float((gw - gc).abs().max() / gc.abs().max()), int((stw != 0).sum())))
