"""Diagnostic: the reference's warm start (sdp.py:101-112: the old solution moved along its
first-order sensitivity to Omega - Omega_old) as a start of the device solver, against the ratio
transport of oei_value_grad_warm_batched and the cold (closed-form) start: Newton iterations per
solve as a function of the distance between the two candidates (q = 20, n = 310).
The predicted weights are computed on the CPU (oracle/reduced_newton.sensitivity) and handed to the
device in the form its start state has (weights over the closed-form weights of the new problem)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hart6, gp as ogp, reduced_newton as rn
from bayesian_optimization_amd import DeviceGP

model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
rs = np.random.RandomState(13)
B, q, d = 96, 20, 6
X = rs.uniform(-.45, .45, (B, q, d))
D = rs.randn(B, q, d)


def closed_form(mu, S, fmin):                  # the solver's cold start (sdp_solve_impl)
    dl, s2 = mu - fmin, np.diag(S)
    r = np.sqrt(s2 + dl * dl)
    pn = np.where(dl >= 0, 0.5 * s2 / (r * (r + dl)), 0.5 * (r - dl) / r)
    return np.maximum(pn / (1.0 + pn.sum()), 1e-300)


v0, g0, st0, fl0, ratio0, it0 = gp.value_grad_warm(X, want_iters=True)
om0 = gp.omega(X).cpu().numpy()
states = []
for b in range(B):
    w0, mu, S = rn.split_omega(om0[b])
    pstar = ratio0[b].cpu().numpy() * closed_form(mu, S, gp.fmin)
    states.append((pstar, rn.state(pstar, mu, S, gp.fmin), mu, S))
print('cold: newton %.2f' % (it0 & 0xffff).double().mean().item())
for eps in (1e-3, 3e-3, 1e-2, 3e-2, 1e-1):
    X1 = np.clip(X + eps * D, -0.5, 0.5)
    om1 = gp.omega(X1).cpu().numpy()
    pred = np.full((B, q), np.nan)
    inside = 0
    for b in range(B):
        pstar, st, mu, S = states[b]
        _, dp = rn.sensitivity(pstar, st, mu, S, gp.fmin, om1[b] - om0[b], return_dp=True)
        pp = pstar + dp[:, 0]
        w1, mu1, S1 = rn.split_omega(om1[b])
        if (pp > 0).all() and pp.sum() < 1:
            inside += 1
            pred[b] = pp / closed_form(mu1, S1, gp.fmin)
    itc = gp.value_grad_warm(X1, want_iters=True)[5]
    itr = gp.value_grad_warm(X1, p_init=ratio0, want_iters=True)[5]
    itp = gp.value_grad_warm(X1, p_init=torch.as_tensor(pred), want_iters=True)[5]
    f = lambda it: (it & 0xffff).double().mean().item()
    s = lambda it: (it >> 16).double().mean().item()
    print('move %.0e: Newton iterations / sweeps  cold %.2f / %.1f   ratio transport %.2f / %.1f   first-order predictor %.2f / %.1f  (%d of %d predictions inside the simplex)'
          % (eps, f(itc), s(itc), f(itr), s(itr), f(itp), s(itp), inside, B))
