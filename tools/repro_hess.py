"""Diagnostic: symmetry and finite-difference check of oei_hessian_batched for one shape.
Usage: python tools/repro_hess.py q d n kind ls noise B train_seed [case.npz with Xs, v]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gp as ogp, hart6
from bayesian_optimization_amd import DeviceGP
q, d, n, kind = [int(x) for x in sys.argv[1:5]]
ls, noise = float(sys.argv[5]), float(sys.argv[6])
B, seed = int(sys.argv[7]), int(sys.argv[8])
fm = 0 if kind == ogp.KERNEL_RBF else 1
om = hart6.make_model(n, d=d, train_seed=seed, kind=kind, floor_mode=fm, lengthscale=ls, noise=noise)
dm = DeviceGP(om.X, om.y, kind, om.kernel.variance, om.kernel.lengthscales, om.noise, floor_mode=om.kernel.floor_mode)
fix = np.load(sys.argv[9]) if len(sys.argv) > 9 else None
for xs in range(1 if fix is not None else 6):
    rs = np.random.RandomState(xs)
    Xs = rs.uniform(-.5, .5, (B, q, d)) if fix is None else fix['Xs']
    if os.environ.get('REPRO_PRE') == '1':          # the calls the sweep makes on the handle before the Hessian
        from bayesian_optimization_amd import sdp_batched
        dm.value_grad(Xs)
        omg = dm.omega(Xs)
        sdp_batched(omg, dm.fmin)
        torch.cuda.synchronize()
    H, hst = dm.hessian(Xs)
    x0 = torch.as_tensor(Xs, device=dm.device).reshape(B, -1)
    v = torch.as_tensor(rs.randn(B, q * d) if fix is None else fix['v'], device=dm.device)
    v = v / v.norm(dim=1, keepdim=True)
    h = 1e-5
    gp_, gm_ = dm.value_grad((x0 + h * v).reshape(B, q, d))[1], dm.value_grad((x0 - h * v).reshape(B, q, d))[1]
    fd = (gp_ - gm_).reshape(B, -1) / (2 * h)
    hv = torch.einsum('bij,bj->bi', H, v)
    eh = ((hv - fd).abs().amax(dim=1) / (1e-12 + fd.abs().amax(dim=1)))
    sym = (H - H.transpose(1, 2)).abs().amax(dim=(1, 2)) / (1e-300 + H.abs().amax(dim=(1, 2)))
    vals, _, st, fl = dm.value_grad(Xs)
    if fix is not None:
        Hh = H.cpu().numpy()
        for b in range(B):
            A = np.abs(Hh[b] - Hh[b].T)
            i, j = np.unravel_index(np.argmax(A), A.shape)
            print('b=%d worst asym at (%d,%d): %.3e vs %.3e; min pair distance %.2e' % (b, i, j, Hh[b][i, j], Hh[b][j, i],
                  min(abs(Xs[b][a, 0] - Xs[b][c, 0]) for a in range(q) for c in range(a))))
    print('xseed %d  fd %s  sym %s  hstatus %s  vstatus %s flags %s  |H| %s' % (xs, ['%.1e' % x for x in eh.tolist()], ['%.1e' % x for x in sym.tolist()], hst.tolist(), st.tolist(), fl.tolist(), ['%.1e' % x for x in H.abs().amax(dim=(1, 2)).tolist()]), flush=True)
