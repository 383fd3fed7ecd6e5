"""Diagnostic: replays one case of tools/stress_parity.py (same random stream) and prints, per candidate
of a q = 1, d = 1 case, the Hessian entry beside central differences of the device gradient at several
steps and the distance to the nearest training point.
    python tools/stress_case_hessian.py <seed> <case>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import gp as ogp, hart6, oei as ooei
from bayesian_optimization_amd import DeviceGP
seed, target = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.RandomState(seed)
case = 0
while True:
    case += 1
    q = int(rs.choice([1, 2, 3, 5, 8, 9, 13, 16, 17, 20, 24, 31, 32, 33, 41, 50, 64]))
    d = int(rs.randint(1, 13))
    n = int(rs.choice([3, 17, 63, 64, 65, 130, 310, 500]))
    kind = int(rs.choice([ogp.KERNEL_RBF, ogp.KERNEL_MATERN32]))
    fm = 0 if kind == ogp.KERNEL_RBF else int(rs.choice([1, 2]))
    ls = float(rs.uniform(0.3, 1.0) * (1 + 0.05 * d))
    noise = float(10.0 ** rs.uniform(-6, -2))
    B = int(rs.choice([1, 2, 5]))
    Xs = rs.uniform(-.5, .5, (B, q, d))
    vdir = rs.randn(B, q * d) if q * d <= 400 else None
    if case == target:
        break
print(q, d, n, kind, fm, ls, noise, B)
om = hart6.make_model(n, d=d, train_seed=case, kind=kind, floor_mode=fm, lengthscale=ls, noise=noise)
dm = DeviceGP(om.X, om.y, kind, om.kernel.variance, om.kernel.lengthscales, om.noise, floor_mode=om.kernel.floor_mode)
H, hst = dm.hessian(Xs)
x0 = torch.as_tensor(Xs, device=dm.device).reshape(B, -1)
for b in range(B):
    dist = np.abs(om.X[:, 0] - Xs[b, 0, 0]).min()
    row = [float(H[b, 0, 0])]
    for h in (1e-4, 1e-5, 1e-6, 1e-7):
        e = torch.zeros_like(x0); e[b, 0] = h
        gp_ = dm.value_grad((x0 + e).reshape(B, q, d))[1].reshape(B, -1)[b, 0]
        gm_ = dm.value_grad((x0 - e).reshape(B, q, d))[1].reshape(B, -1)[b, 0]
        row.append(float((gp_ - gm_) / (2 * h)))
    print('b=%d  nearest training point at %.3e  H %.8e  fd(h=1e-4..1e-7) %s' % (b, dist, row[0], ' '.join('%.8e' % x for x in row[1:])))
