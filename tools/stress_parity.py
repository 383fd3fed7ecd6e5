"""Randomised parity sweep: random shapes, kernels and hyper-parameters through the C ABI; value and
gradient against the CPU restatement and the KKT certificate, the Hessian against central
differences of the device gradient.
Usage: python tools/stress_parity.py [seconds] [seed] [first case] [last case]
(cases before the first only draw their random numbers, so a stretch of an earlier sweep can be
re-run: what a call leaves behind in LDS and in freed buffers is part of the next call's input).
tests/test_gpu_stress.py runs a fixed slice of it in the collected suite."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gp as ogp, hart6, reduced_newton, sdp_ref          # noqa: E402


def sweep(budget=120.0, seed=0, first_case=1, last_case=1 << 30, verbose=True):
    """Returns (cases run, list of messages of cases above the tight bars, worst deviations)."""
    from bayesian_optimization_amd import DeviceGP, sdp_batched
    rs = np.random.RandomState(seed)
    t_end = time.time() + budget
    worst = dict(value=0.0, grad=0.0, grad_illcond=0.0, cert=0.0, hess=0.0, sym=0.0, status=0)
    fails = []
    case = 0
    while time.time() < t_end and case < last_case:
        case += 1
        q = int(rs.choice([1, 2, 3, 5, 8, 9, 13, 16, 17, 20, 24, 31, 32, 33, 41, 50, 64]))
        d = int(rs.randint(1, 13))
        n = int(rs.choice([3, 17, 63, 64, 65, 130, 310, 500]))
        kind = int(rs.choice([ogp.KERNEL_RBF, ogp.KERNEL_MATERN32]))
        fm = 0 if kind == ogp.KERNEL_RBF else int(rs.choice([1, 2]))
        ls = float(rs.uniform(0.3, 1.0) * (1 + 0.05 * d))
        noise = float(10.0 ** rs.uniform(-6, -2))
        B = int(rs.choice([1, 2, 5]))
        tag = 'case %d: q=%d d=%d n=%d kind=%d fm=%d ls=%.2f noise=%.1e B=%d' % (case, q, d, n, kind, fm, ls, noise, B)
        Xs = rs.uniform(-.5, .5, (B, q, d))
        vdir = rs.randn(B, q * d) if q * d <= 400 else None      # drawn for every case: the stream stays in step
        if case < first_case:
            continue
        om = hart6.make_model(n, d=d, train_seed=case, kind=kind, floor_mode=fm, lengthscale=ls, noise=noise)
        dm = DeviceGP(om.X, om.y, kind, om.kernel.variance, om.kernel.lengthscales, om.noise, floor_mode=om.kernel.floor_mode)
        value, grad, status, flags = [t.cpu().numpy() for t in dm.value_grad(Xs)]
        if (status != 0).any():
            worst['status'] += 1
            fails.append(tag + ' status %s' % status)
            if verbose:
                print(fails[-1], flush=True)
            dm.close()
            continue
        omg = dm.omega(Xs)
        v2, M, Y, st2, _ = [t.cpu().numpy() for t in sdp_batched(omg, dm.fmin)]
        msg = ''
        for b in range(B):
            o = omg[b].cpu().numpy()
            cert = sdp_ref.kkt_certificate(o, dm.fmin, M[b], list(Y[b]))
            c = max(v for k, v in cert.items() if k not in ('pobj', 'dobj'))
            obj = reduced_newton.solve(o, dm.fmin)[0]
            ev = abs(value[b] - obj) / abs(obj)
            _, g0 = om.inner_and_grad(Xs[b], M[b])
            eg = np.abs(grad[b] - g0).max() / (1e-12 + np.abs(g0).max())
            # RBF kernels at noise below 1e-5 (cond(K) 1e8 and more): the fp64 restatement of the
            # gradient is itself only good to about 1e-4 there, so those cases are kept apart
            gkey = 'grad_illcond' if (kind == ogp.KERNEL_RBF and noise < 1e-5) else 'grad'
            worst['value'] = max(worst['value'], ev); worst[gkey] = max(worst[gkey], eg); worst['cert'] = max(worst['cert'], c)
            if ev > 1e-8 or eg > 1e-6 or c > 1e-6 or not np.allclose(o, om.omega(Xs[b]), rtol=1e-7, atol=1e-9):
                msg += ' [b=%d value %.1e grad %.1e cert %.1e]' % (b, ev, eg, c)
        if vdir is not None:
            H, hst = dm.hessian(Xs)
            x0 = torch.as_tensor(Xs, device=dm.device).reshape(B, -1)
            v = torch.as_tensor(vdir, device=dm.device)
            v = v / v.norm(dim=1, keepdim=True)
            h = 1e-5
            gp_, gm_ = dm.value_grad((x0 + h * v).reshape(B, q, d))[1], dm.value_grad((x0 - h * v).reshape(B, q, d))[1]
            fd = (gp_ - gm_).reshape(B, -1) / (2 * h)
            hv = torch.einsum('bij,bj->bi', H, v)
            eh = float(((hv - fd).abs().amax(dim=1) / (1e-12 + fd.abs().amax(dim=1))).max())
            sym = float((H - H.transpose(1, 2)).abs().max() / (1e-300 + H.abs().max()))
            worst['hess'] = max(worst['hess'], eh)
            worst['sym'] = max(worst['sym'], sym)
            if int((hst != 0).sum()):
                worst['status'] += 1
            if eh > 1e-4 or sym > 1e-9 or int((hst != 0).sum()):
                msg += ' [hessian fd %.1e sym %.1e status %s]' % (eh, sym, hst.tolist())
        if msg:
            fails.append(tag + msg)
        if verbose:
            print(tag + (msg or ' ok'), flush=True)
        dm.close()
    return case, fails, worst


if __name__ == '__main__':
    a = sys.argv
    ncase, fails, worst = sweep(float(a[1]) if len(a) > 1 else 120.0, int(a[2]) if len(a) > 2 else 0,
                                int(a[3]) if len(a) > 3 else 1, int(a[4]) if len(a) > 4 else 1 << 30)
    print('cases %d, failures %d, worst %s' % (ncase, len(fails), worst))
    for f in fails:
        print('FAIL', f)
