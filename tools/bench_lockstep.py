"""Diagnostic: the reference's sequential multistart (scipy L-BFGS-B, one acquisition call per
function evaluation) against the lock-step optimiser, same starts (q = 20, n = 310)."""
import sys, time
import numpy as np, scipy.optimize, torch
sys.path.insert(0, '.')
from oracle import hart6, gp as ogp
from bayesian_optimization_amd import DeviceGP, lockstep

q = 20
model = hart6.make_model(310)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.5], 1e-6, floor_mode=1)
starts = np.random.RandomState(3).uniform(-.5, .5, (20, q, 6))
gp.value_grad(starts)                                  # warm-up

def f1(z):
    v, g, _, _ = gp.value_grad(z.reshape(1, q, 6))
    return float(v.item()), g.cpu().numpy().ravel()
t0 = time.perf_counter(); seq = []; nfev = 0
for b in range(20):
    r = scipy.optimize.minimize(f1, starts[b].ravel(), jac=True, method='L-BFGS-B', bounds=[(-.5, .5)] * (q * 6))
    seq.append(r.fun); nfev += r.nfev
t_seq = time.perf_counter() - t0
torch.cuda.synchronize(); t0 = time.perf_counter()
res, best = lockstep.optimize_multistart(gp, starts, [-.5] * 6, [.5] * 6)
torch.cuda.synchronize(); t_lock = time.perf_counter() - t0
print('20 starts: sequential %.2f s (%d evaluations, best %.6f); lock-step %.2f s (max %d evaluations per run, best %.6f)'
      % (t_seq, nfev, min(seq), t_lock, int(res.nfev.max()), float(best[0])))
big = np.random.RandomState(4).uniform(-.5, .5, (2048, q, 6))
for warm in (False, True):
    st = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res, best = lockstep.optimize_multistart(gp, big, [-.5] * 6, [.5] * 6, warm_start=warm, stats=st)
    torch.cuda.synchronize(); t_big = time.perf_counter() - t0
    print('2048 starts lock-step, SDP warm start %s: %.2f s, %d evaluations in total, Newton iterations per evaluation %s, best %.6f, converged %d'
          % (warm, t_big, int(res.nfev.sum()), ('%.2f' % (st['newton'] / st['evals'])) if 'newton' in st else 'n/a (3.2 cold)',
             float(best[0]), int(res.converged.sum())))

for method in ('lbfgs', 'newton'):
    st = {}
    sub = big[:512]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res, best = lockstep.optimize_multistart(gp, sub, [-.5] * 6, [.5] * 6, method=method, stats=st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('512 starts lock-step, method %s: %.2f s, %d value+gradient evaluations, best %.6f, mean %.6f, converged %d'
          % (method, dt, st['evals'], float(best[0]), float(res.fun.mean()), int(res.converged.sum())))
