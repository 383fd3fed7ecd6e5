"""Lock-step multistart descent: many bounded quasi-Newton runs advanced together.

The reference optimises the acquisition function from `opt_restarts` random starts one after the
other, each with scipy's L-BFGS-B calling `acquisition` once per function evaluation
(bo.py:129-192, solvers.py:10-21).  The starts are independent, so here all of them step in
lock-step: one batched value+gradient call per line-search trial serves every run that still needs
one.  The iterates, the limited-memory pairs and all vector arithmetic stay on the device that the
evaluations run on; the host only sees a few scalars per iteration (how many runs are still active).

Method per run (the same for all, vectorised over the batch):
  * projected gradient test  max|x - P(x - g)| <= pgtol  (the stopping rule of L-BFGS-B, `pgtol`)
    and relative decrease  (f - f+) <= ftol max(|f|, |f+|, 1)  (`factr * epsmch`);
  * direction d = -H g restricted to the variables not blocked at a bound (lower bound with
    g > 0 or upper bound with g < 0), H the two-loop L-BFGS operator over the last `history`
    curvature pairs; steepest descent if d is not a descent direction;
  * projected backtracking line search  x+ = P(x + t d),  accept when
    f(x+) <= f(x) + c1 g.(x+ - x);  pairs (s, y) are stored when s.y > 1e-10 |s||y|.
This is a projected L-BFGS, not a port of the Fortran L-BFGS-B (no generalised Cauchy point, no
subspace minimisation); iterates differ from scipy's, local optima found from the same start are
the same in the tests.
"""
from collections import namedtuple

import torch

LockstepResult = namedtuple('LockstepResult', 'x fun grad nit nfev converged message')


def minimize_lockstep(fun, X0, lower, upper, history=10, max_iter=200, pgtol=1e-5, ftol=2.2e-9,
                      c1=1e-4, max_ls=25):
    """Minimise `fun` from every row of X0 [B, n] inside the box [lower, upper] (tensors [n] or
    scalars).  `fun(X[b, n]) -> (f[b], g[b, n])` is called with sub-batches (only the runs that
    need an evaluation) and may return +inf / NaN for points it cannot evaluate (they are
    rejected by the line search).  Returns LockstepResult with per-run fields."""
    x = torch.as_tensor(X0).clone()
    B, n = x.shape
    dev, dt = x.device, x.dtype
    lo = torch.as_tensor(lower, dtype=dt, device=dev).expand(n).clone()
    hi = torch.as_tensor(upper, dtype=dt, device=dev).expand(n).clone()
    x = torch.minimum(torch.maximum(x, lo), hi)
    indexed = getattr(fun, 'wants_index', False)     # fun(X, idx): idx = the runs the rows belong to
    f, g = fun(x, torch.arange(B, device=dev)) if indexed else fun(x)
    f = f.clone()
    g = g.clone().reshape(B, n)
    nfev = torch.ones(B, dtype=torch.int64, device=dev)
    nit = torch.zeros(B, dtype=torch.int64, device=dev)
    bad0 = ~torch.isfinite(f)
    S = torch.zeros((B, history, n), dtype=dt, device=dev)
    Y = torch.zeros((B, history, n), dtype=dt, device=dev)
    rho = torch.zeros((B, history), dtype=dt, device=dev)
    gamma = torch.ones(B, dtype=dt, device=dev)
    done = bad0.clone()
    converged = torch.zeros(B, dtype=torch.bool, device=dev)
    first = torch.ones(B, dtype=torch.bool, device=dev)
    slot = 0
    for _ in range(max_iter):
        pg = x - torch.minimum(torch.maximum(x - g, lo), hi)
        hit = (pg.abs().amax(dim=1) <= pgtol) & ~done
        converged |= hit
        done |= hit
        if bool(done.all()):
            break
        free = ~(((x <= lo) & (g > 0)) | ((x >= hi) & (g < 0)))
        fm = free.to(dt)
        # two-loop recursion, newest pair first (ring buffer: slot-1, slot-2, ...)
        q = g * fm
        alphas = []
        order = [(slot - 1 - i) % history for i in range(history)]
        for i in order:
            a = rho[:, i] * (S[:, i] * q).sum(dim=1)
            alphas.append(a)
            q = q - a[:, None] * Y[:, i]
        r = gamma[:, None] * q
        for i, a in zip(reversed(order), reversed(alphas)):
            b = rho[:, i] * (Y[:, i] * r).sum(dim=1)
            r = r + (a - b)[:, None] * S[:, i]
        d = -r * fm
        gd = (g * d).sum(dim=1)
        reset = ~(gd < 0)
        if bool(reset.any()):
            d = torch.where(reset[:, None], -g * fm, d)
            gd = (g * d).sum(dim=1)
        dn = d.norm(dim=1)
        t = torch.where(first, torch.clamp(1.0 / torch.clamp(dn, min=1e-300), max=1.0), torch.ones_like(dn))
        need = ~done & (dn > 0)
        stalled = ~done & ~(dn > 0)
        done |= stalled
        x_new, f_new, g_new = x.clone(), f.clone(), g.clone()
        accepted = torch.zeros(B, dtype=torch.bool, device=dev)
        for _ls in range(max_ls):
            idx = torch.nonzero(need & ~accepted).reshape(-1)
            if idx.numel() == 0:
                break
            xt = torch.minimum(torch.maximum(x[idx] + t[idx, None] * d[idx], lo), hi)
            ft, gt = fun(xt, idx) if indexed else fun(xt)
            gt = gt.reshape(idx.numel(), n)
            nfev[idx] += 1
            ok = ft <= f[idx] + c1 * (g[idx] * (xt - x[idx])).sum(dim=1)      # False for NaN / inf
            oki = idx[ok]
            x_new[oki], f_new[oki], g_new[oki] = xt[ok], ft[ok], gt[ok]
            accepted[oki] = True
            t[idx[~ok]] *= 0.5
        failed = need & ~accepted
        done |= failed
        s = x_new - x
        y = g_new - g
        sy = (s * y).sum(dim=1)
        valid = accepted & (sy > 1e-10 * s.norm(dim=1) * y.norm(dim=1))
        S[:, slot] = torch.where(valid[:, None], s, torch.zeros_like(s))
        Y[:, slot] = torch.where(valid[:, None], y, torch.zeros_like(y))
        rho[:, slot] = torch.where(valid, 1.0 / torch.where(valid, sy, torch.ones_like(sy)), torch.zeros_like(sy))
        gamma = torch.where(valid, sy / torch.clamp((y * y).sum(dim=1), min=1e-300), gamma)
        slot = (slot + 1) % history
        small = accepted & ((f - f_new) <= ftol * torch.maximum(torch.maximum(f.abs(), f_new.abs()),
                                                                 torch.ones_like(f)))
        converged |= small
        done |= small
        nit += accepted.to(torch.int64)
        first &= ~accepted
        x, f, g = x_new, f_new, g_new
    message = ['converged' if c else ('no finite start value' if b else 'stopped')
               for c, b in zip(converged.tolist(), bad0.tolist())]
    return LockstepResult(x=x, fun=f, grad=g, nit=nit, nfev=nfev, converged=converged, message=message)


LBFGS_D = dict(X=0, F=1, G=2, DIR=3, STEP=4, XT=5, S=6, Y=7, RHO=8, GAMMA=9, XC=10, LO=11, HI=12)
LBFGS_I = dict(STATUS=0, SLOT=1, NIT=2, NFEV=3, LS=4, FIRST=5, ROWOF=6, MAP=7, COUNT=8, NEED=9)
LBFGS_STATUS = {0: 'stopped', 1: 'converged', 2: 'converged', 3: 'stopped', 4: 'stopped',
                5: 'no finite start value', 6: 'stopped'}


def minimize_lockstep_device(fun, X0, lower, upper, history=10, max_iter=200, pgtol=1e-5, ftol=2.2e-9,
                             c1=1e-4, max_ls=25):
    """minimize_lockstep with the optimiser state and all of its arithmetic in HIP kernels
    (`oei_lbfgs_*`): per round one call of `fun` on the dense list of trial points and one
    launch that advances every run -- Armijo test, curvature pair, stopping tests, two-loop
    direction, next trial point, one wave per run.  Runs do not wait for each other at iteration
    boundaries, so the number of rounds is the largest number of evaluations any run needs.
    Same method, constants and result type as minimize_lockstep; X0 must be a GPU tensor."""
    import ctypes
    from ._lib import lib, check
    x0 = torch.as_tensor(X0)
    if not x0.is_cuda:
        raise ValueError('minimize_lockstep_device needs GPU tensors (minimize_lockstep is the generic torch version)')
    x0 = x0.to(torch.float64).contiguous()
    B, n = x0.shape
    dev = x0.device
    lo = torch.as_tensor(lower, dtype=torch.float64, device=dev).expand(n).contiguous()
    hi = torch.as_tensor(upper, dtype=torch.float64, device=dev).expand(n).contiguous()
    nd, ni = ctypes.c_size_t(0), ctypes.c_size_t(0)
    od = (ctypes.c_size_t * len(LBFGS_D))()
    oi = (ctypes.c_size_t * len(LBFGS_I))()
    check(lib.oei_lbfgs_layout(B, n, history, ctypes.byref(nd), ctypes.byref(ni), od, oi), 'oei_lbfgs_layout')
    dws = torch.zeros(nd.value, dtype=torch.float64, device=dev)
    iws = torch.zeros(ni.value, dtype=torch.int32, device=dev)

    def dview(name, *shape):
        o = od[LBFGS_D[name]]
        return dws[o:o + int(torch.Size(shape).numel())].view(*shape)

    def iview(name, count=B):
        o = oi[LBFGS_I[name]]
        return iws[o:o + count]
    indexed = getattr(fun, 'wants_index', False)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.oei_lbfgs_init(B, n, history, p(x0), p(lo), p(hi), p(dws), p(iws), stream), 'oei_lbfgs_init')
        Xc, rows = dview('XC', B, n), iview('MAP')
        count = B
        cnt = ctypes.c_int(0)
        for _ in range(max(1, max_iter) * max_ls + 2):      # every round ends a trial of every running run
            ft, gt = fun(Xc[:count], rows[:count].to(torch.int64)) if indexed else fun(Xc[:count])
            ft = ft.to(torch.float64).contiguous()
            gt = gt.to(torch.float64).reshape(count, n).contiguous()
            check(lib.oei_lbfgs_advance(B, n, history, p(ft), p(gt), pgtol, ftol, c1, max_ls, max_iter,
                                        p(dws), p(iws), ctypes.byref(cnt), stream), 'oei_lbfgs_advance')
            count = cnt.value
            if count == 0:
                break
    status = iview('STATUS').clone()
    converged = (status == 1) | (status == 2)
    message = [LBFGS_STATUS[s] for s in status.tolist()]
    return LockstepResult(x=dview('X', B, n).clone(), fun=dview('F', B).clone(), grad=dview('G', B, n).clone(),
                          nit=iview('NIT').to(torch.int64), nfev=iview('NFEV').to(torch.int64),
                          converged=converged, message=message)


def minimize_lockstep_newton(fun, hess, X0, lower, upper, max_iter=60, pgtol=1e-5, ftol=2.2e-9,
                             c1=1e-4, max_ls=20):
    """Projected Newton descent of every row of X0 [B, n] in lock-step: the batched consumer of
    the exact Hessian (the role KNITRO's SQP plays in the reference, solvers.py:56-69, 71-141).
    `fun(X[b, n]) -> (f[b], g[b, n])`, `hess(X[b, n]) -> H[b, n, n]` (a zero or non-finite matrix
    is allowed: the step then falls back to a scaled gradient step).  Per iteration and run:
      * variables blocked at a bound (lower with g > 0, upper with g < 0) are frozen;
      * on the free variables (H + lam I) d = -g with the smallest lam from a geometric ladder
        that makes the matrix positive definite (batched Cholesky; the acquisition is not
        convex, so lam > 0 is the rule far from an optimum);
      * projected backtracking line search x+ = P(x + t d) with the Armijo rule of
        minimize_lockstep; same stopping rules (projected gradient, relative decrease).
    Returns LockstepResult; `nfev` counts value+gradient calls, Hessian calls equal `nit` + 1 at most."""
    x = torch.as_tensor(X0).clone()
    B, n = x.shape
    dev, dt = x.device, x.dtype
    lo = torch.as_tensor(lower, dtype=dt, device=dev).expand(n).clone()
    hi = torch.as_tensor(upper, dtype=dt, device=dev).expand(n).clone()
    x = torch.minimum(torch.maximum(x, lo), hi)
    f, g = fun(x)
    f = f.clone()
    g = g.clone().reshape(B, n)
    nfev = torch.ones(B, dtype=torch.int64, device=dev)
    nit = torch.zeros(B, dtype=torch.int64, device=dev)
    bad0 = ~torch.isfinite(f)
    done = bad0.clone()
    converged = torch.zeros(B, dtype=torch.bool, device=dev)
    eye = torch.eye(n, dtype=dt, device=dev)
    rung = torch.zeros(B, dtype=torch.int64, device=dev)
    for _ in range(max_iter):
        pg = x - torch.minimum(torch.maximum(x - g, lo), hi)
        hit = (pg.abs().amax(dim=1) <= pgtol) & ~done
        converged |= hit
        done |= hit
        idx = torch.nonzero(~done).reshape(-1)
        if idx.numel() == 0:
            break
        xa, ga, fa = x[idx], g[idx], f[idx]
        H = hess(xa)
        H = torch.where(torch.isfinite(H), H, torch.zeros_like(H))
        H = 0.5 * (H + H.transpose(1, 2))
        free = ~(((xa <= lo) & (ga > 0)) | ((xa >= hi) & (ga < 0)))
        fm = free.to(dt)
        Hm = H * fm[:, :, None] * fm[:, None, :] + torch.diag_embed(1.0 - fm)
        scale = torch.clamp(Hm.abs().amax(dim=(1, 2)), min=1e-12)
        # rung k of the ladder: lam = 0 (k = 0), 1e-4 scale 10^(k-1) above; a run starts its climb two
        # rungs below the rung its previous iteration ended on
        k_run = torch.clamp(rung[idx] - 2, min=0)
        lam = torch.where(k_run > 0, 1e-4 * scale * 10.0 ** (k_run - 1).to(dt), torch.zeros_like(scale))
        L = eye.expand(idx.numel(), n, n).clone()
        solved = torch.zeros(idx.numel(), dtype=torch.int32, device=dev)
        for _k in range(12):
            trying = (solved == 0) & (k_run < 12)
            if not bool(trying.any()):
                break
            Lk, info = torch.linalg.cholesky_ex(Hm + lam[:, None, None] * eye)
            ok = (info == 0) & trying
            L = torch.where(ok[:, None, None], Lk, L)
            solved = torch.maximum(solved, ok.to(torch.int32))
            climb = trying & ~ok
            lam = torch.where(climb, torch.where(lam > 0, 10.0 * lam, 1e-4 * scale), lam)
            k_run = torch.where(climb, k_run + 1, k_run)
        unsolved = solved == 0
        rung[idx] = torch.where(unsolved, torch.zeros_like(k_run), k_run)
        d = -torch.cholesky_solve((ga * fm)[:, :, None], torch.where(unsolved[:, None, None], eye.expand_as(L), L))[:, :, 0] * fm
        gd = (ga * d).sum(dim=1)
        resetv = ~(gd < 0) | unsolved
        d = torch.where(resetv[:, None], -ga * fm, d)
        dn = d.norm(dim=1)
        # first trial: the Newton step, but no longer than the box diagonal
        t = torch.clamp((hi - lo).norm() / torch.clamp(dn, min=1e-300), max=1.0)
        stalled = ~(dn > 0)
        x_new, f_new, g_new = xa.clone(), fa.clone(), ga.clone()
        accepted = torch.zeros(idx.numel(), dtype=torch.bool, device=dev)
        for _ls in range(max_ls):
            sub = torch.nonzero(~accepted & ~stalled).reshape(-1)
            if sub.numel() == 0:
                break
            xt = torch.minimum(torch.maximum(xa[sub] + t[sub, None] * d[sub], lo), hi)
            ft, gt = fun(xt)
            gt = gt.reshape(sub.numel(), n)
            nfev[idx[sub]] += 1
            ok = ft <= fa[sub] + c1 * (ga[sub] * (xt - xa[sub])).sum(dim=1)      # False for NaN / inf
            oki = sub[ok]
            x_new[oki], f_new[oki], g_new[oki] = xt[ok], ft[ok], gt[ok]
            accepted[oki] = True
            t[sub[~ok]] *= 0.25
        small = accepted & ((fa - f_new) <= ftol * torch.maximum(torch.maximum(fa.abs(), f_new.abs()),
                                                                  torch.ones_like(fa)))
        converged[idx] |= small
        done[idx] |= small | ~accepted
        nit[idx] += accepted.to(torch.int64)
        x[idx], f[idx], g[idx] = x_new, f_new, g_new
    message = ['converged' if c else ('no finite start value' if b else 'stopped')
               for c, b in zip(converged.tolist(), bad0.tolist())]
    return LockstepResult(x=x, fun=f, grad=g, nit=nit, nfev=nfev, converged=converged, message=message)


NEWTON_D = dict(X=0, F=1, G=2, DIR=3, STEP=4, XT=5, XC=6, XH=7, LO=8, HI=9)
NEWTON_I = dict(STATUS=0, NIT=1, NFEV=2, NHEV=3, LS=4, PHASE=5, ROWOF=6, MAP_T=7, MAP_H=8, COUNTS=9, NEED=10, RUNG=11)
NEWTON_MAX_N = 141


def minimize_lockstep_newton_device(fun, hess, X0, lower, upper, max_iter=60, pgtol=1e-5, ftol=2.2e-9,
                                    c1=1e-4, max_ls=20):
    """minimize_lockstep_newton with the state, the Levenberg ladder, the Cholesky factorisation
    and the line search in HIP kernels (`oei_newton_*`, one workgroup per run).  Per round the
    host evaluates two dense lists -- `fun` on the trial points, `hess` on the iterates that were
    just accepted -- and one launch advances every run.  n <= 141 variables per run (the factor
    lives in LDS); X0 must be a GPU tensor."""
    import ctypes
    from ._lib import lib, check
    x0 = torch.as_tensor(X0)
    if not x0.is_cuda:
        raise ValueError('minimize_lockstep_newton_device needs GPU tensors')
    x0 = x0.to(torch.float64).contiguous()
    B, n = x0.shape
    dev = x0.device
    lo = torch.as_tensor(lower, dtype=torch.float64, device=dev).expand(n).contiguous()
    hi = torch.as_tensor(upper, dtype=torch.float64, device=dev).expand(n).contiguous()
    nd, ni = ctypes.c_size_t(0), ctypes.c_size_t(0)
    od = (ctypes.c_size_t * len(NEWTON_D))()
    oi = (ctypes.c_size_t * len(NEWTON_I))()
    check(lib.oei_newton_layout(B, n, ctypes.byref(nd), ctypes.byref(ni), od, oi), 'oei_newton_layout')
    dws = torch.zeros(nd.value, dtype=torch.float64, device=dev)
    iws = torch.zeros(ni.value, dtype=torch.int32, device=dev)

    def dview(name, *shape):
        o = od[NEWTON_D[name]]
        return dws[o:o + int(torch.Size(shape).numel())].view(*shape)

    def iview(name, count=B):
        o = oi[NEWTON_I[name]]
        return iws[o:o + count]
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.oei_newton_init(B, n, p(x0), p(lo), p(hi), p(dws), p(iws), stream), 'oei_newton_init')
        Xc, Xh = dview('XC', B, n), dview('XH', B, n)
        n_trial, n_hess = B, 0
        cnt = (ctypes.c_int * 2)()
        for _ in range(2 * (max(1, max_iter) * max_ls + 2)):
            ft = gt = Ht = None
            if n_trial > 0:
                ft, gt = fun(Xc[:n_trial])
                ft = ft.to(torch.float64).contiguous()
                gt = gt.to(torch.float64).reshape(n_trial, n).contiguous()
            if n_hess > 0:
                Ht = hess(Xh[:n_hess]).to(torch.float64).reshape(n_hess, n, n).contiguous()
            check(lib.oei_newton_advance(B, n, p(ft), p(gt), p(Ht), pgtol, ftol, c1, max_ls, max_iter,
                                         p(dws), p(iws), cnt, stream), 'oei_newton_advance')
            n_trial, n_hess = cnt[0], cnt[1]
            if n_trial == 0 and n_hess == 0:
                break
    status = iview('STATUS').clone()
    converged = (status == 1) | (status == 2)
    message = [LBFGS_STATUS[s] for s in status.tolist()]
    return LockstepResult(x=dview('X', B, n).clone(), fun=dview('F', B).clone(), grad=dview('G', B, n).clone(),
                          nit=iview('NIT').to(torch.int64), nfev=iview('NFEV').to(torch.int64),
                          converged=converged, message=message)


def minimize_acquisition_rounds(gp, q, X0, lower, upper, method='lbfgs', project_fallback=None, history=10,
                                max_iter=None, pgtol=1e-5, ftol=2.2e-9, c1=1e-4, max_ls=None):
    """The multistart descent on the acquisition of `gp` with the loop over rounds inside the
    library (`oei_lbfgs_minimize_rounds` / `oei_newton_minimize_rounds`): same state machines, same
    constants and results as `minimize_lockstep_device` / `minimize_lockstep_newton_device` driven
    with `acquisition_fun` / `acquisition_hess_fun`, without the per-round host work of a Python loop
    (at the 20 starts of a BO loop a round is then bound by the device, not by the host).
    A trial point that the reference's `project` would shrink (only with `project_fallback`, and
    noise <= 1e-4) hands one round back to Python: the list is evaluated through `acquisition_fun`
    with the fallback, the runs advance once, and the library loop goes on.
    Iteration caps: L-BFGS 2000 (scipy's L-BFGS-B default is 15000, solvers.py:11-17: the stopping
    rules pgtol / ftol end the runs, the cap is a safety net), Newton 200."""
    import ctypes
    from ._lib import lib, check
    newton = method == 'newton'
    if max_iter is None:
        max_iter = 200 if newton else 2000
    if max_ls is None:
        max_ls = 20 if newton else 25
    x0 = torch.as_tensor(X0, dtype=torch.float64, device=gp.device).contiguous()
    B, n = x0.shape
    d = gp.dim
    assert n == q * d
    dev = gp.device
    lo = torch.as_tensor(lower, dtype=torch.float64, device=dev).expand(n).contiguous()
    hi = torch.as_tensor(upper, dtype=torch.float64, device=dev).expand(n).contiguous()
    nd, ni, nw = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    names_d, names_i = (NEWTON_D, NEWTON_I) if newton else (LBFGS_D, LBFGS_I)
    od = (ctypes.c_size_t * len(names_d))()
    oi = (ctypes.c_size_t * len(names_i))()
    if newton:
        check(lib.oei_newton_layout(B, n, ctypes.byref(nd), ctypes.byref(ni), od, oi), 'oei_newton_layout')
    else:
        check(lib.oei_lbfgs_layout(B, n, history, ctypes.byref(nd), ctypes.byref(ni), od, oi), 'oei_lbfgs_layout')
    check(lib.oei_multistart_workspace(B, n, int(newton), ctypes.byref(nw)), 'oei_multistart_workspace')
    dws = torch.zeros(nd.value, dtype=torch.float64, device=dev)
    iws = torch.zeros(ni.value, dtype=torch.int32, device=dev)
    work = torch.empty(nw.value, dtype=torch.float64, device=dev)

    def dview(name, *shape):
        o = od[names_d[name]]
        return dws[o:o + int(torch.Size(shape).numel())].view(*shape)

    def iview(name, count=B):
        o = oi[names_i[name]]
        return iws[o:o + count]
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    projecting = project_fallback is not None and gp.noise <= 1e-4
    use_flags = (1 if projecting else 0) | (2 if gp.noise <= 1e-4 else 0)
    fun = hess = None
    info = (ctypes.c_int * 4)()
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if newton:
            check(lib.oei_newton_init(B, n, p(x0), p(lo), p(hi), p(dws), p(iws), stream), 'oei_newton_init')
        else:
            check(lib.oei_lbfgs_init(B, n, history, p(x0), p(lo), p(hi), p(dws), p(iws), stream), 'oei_lbfgs_init')
        info[0], info[1] = B, 0
        while True:
            if newton:
                check(lib.oei_newton_minimize_rounds(gp._h, B, q, pgtol, ftol, c1, max_ls, max_iter, p(dws), p(iws),
                                                     p(work), use_flags, 1 << 30, info, stream), 'oei_newton_minimize_rounds')
                n_trial, n_hess, flagged = info[0], info[1], info[2]
            else:
                check(lib.oei_lbfgs_minimize_rounds(gp._h, B, q, history, pgtol, ftol, c1, max_ls, max_iter, p(dws),
                                                    p(iws), p(work), use_flags, 1 << 30, info, stream), 'oei_lbfgs_minimize_rounds')
                n_trial, n_hess, flagged = info[0], 0, info[1]
            if n_trial == 0 and n_hess == 0:
                break
            assert flagged > 0, 'the round loop returned with runs left and nothing to hand back'
            # one round through the host: the flagged trial points get the reduced problem's value
            if fun is None:
                fun = acquisition_fun(gp, q, project_fallback)
                hess = acquisition_hess_fun(gp, q)
            ft = gt = Ht = None
            if n_trial > 0:
                ft, gt = fun(dview('XC', B, n)[:n_trial])
                ft = ft.to(torch.float64).contiguous()
                gt = gt.to(torch.float64).reshape(n_trial, n).contiguous()
            if newton:
                if n_hess > 0:
                    Ht = hess(dview('XH', B, n)[:n_hess]).to(torch.float64).reshape(n_hess, n, n).contiguous()
                cnt = (ctypes.c_int * 2)()
                check(lib.oei_newton_advance(B, n, p(ft), p(gt), p(Ht), pgtol, ftol, c1, max_ls, max_iter,
                                             p(dws), p(iws), cnt, stream), 'oei_newton_advance')
                info[0], info[1] = cnt[0], cnt[1]
            else:
                cnt = ctypes.c_int(0)
                check(lib.oei_lbfgs_advance(B, n, history, p(ft), p(gt), pgtol, ftol, c1, max_ls, max_iter,
                                            p(dws), p(iws), ctypes.byref(cnt), stream), 'oei_lbfgs_advance')
                info[0] = cnt.value
            if info[0] == 0 and (not newton or info[1] == 0):
                break
    status = iview('STATUS').clone()
    converged = (status == 1) | (status == 2)
    message = [LBFGS_STATUS[s] for s in status.tolist()]
    return LockstepResult(x=dview('X', B, n).clone(), fun=dview('F', B).clone(), grad=dview('G', B, n).clone(),
                          nit=iview('NIT').to(torch.int64), nfev=iview('NFEV').to(torch.int64),
                          converged=converged, message=message)


def acquisition_hess_fun(gp, q):
    """Adapter from a DeviceGP to the `hess` of minimize_lockstep_newton: X [b, q*d] -> H [b, q*d, q*d]
    through the batched acquisition_hessian (oei.py:41-64).  Candidates whose solve failed, and
    candidates the reference's `project` would shrink (its Hessian is then zero, oei.py:53-55),
    get a zero matrix."""
    d = gp.dim

    def hess(X):
        b = X.shape[0]
        Xb = X.reshape(b, q, d)
        H, status = gp.hessian(Xb)
        keep = status == 0
        if gp.noise <= 1e-4:                      # `project` is the identity above that (oei.py:174-175)
            keep = keep & (gp.project_flags(Xb) == 0)
        return torch.where(keep[:, None, None], H, torch.zeros_like(H))
    return hess


def acquisition_fun(gp, q, project_fallback=None, n_runs=None, warm_start=False, stats=None):
    """Adapter from a DeviceGP to the `fun` of minimize_lockstep: X [b, q*d] -> (value, grad).
    Candidates whose SDP did not solve get +inf (the line search then backs off).
    `project_fallback(X_row numpy) -> (value, grad numpy)` re-evaluates candidates that the
    duplicate test of the reference's `project` (oei.py:157-199) flags.
    With `n_runs` and `warm_start` every run keeps the solver state of its last evaluation and
    starts the next SDP solve from it, the role of the reference's warm start (sdp.py:41-48, 63-65,
    77-121; the history is per run, so there is nothing to reset between multistarts as bo.py:146
    does).  Measured (q = 20, n = 310): the state beats the closed-form start for moves up to
    about 3e-2 of the box per coordinate (2.1 instead of 3.2 Newton iterations at 1e-3), but the
    quasi-Newton steps of a descent are larger than that on average (3.14 against 3.2 iterations
    per evaluation over a whole multistart), so it is off by default.
    `stats`: dict that receives the total of Newton iterations and evaluations."""
    d = gp.dim
    p_state = None
    if warm_start and n_runs is not None:
        p_state = torch.full((n_runs, q), float('nan'), dtype=torch.float64, device=gp.device)

    def fun(X, idx=None):
        b = X.shape[0]
        if p_state is not None and idx is not None:
            value, grad, status, flags, p_out, iters = gp.value_grad_warm(
                X.reshape(b, q, d), p_init=p_state[idx], want_iters=stats is not None)
            p_state[idx] = torch.where((status == 0)[:, None], p_out, p_state[idx])
            if stats is not None:
                stats['newton'] = stats.get('newton', 0) + int((iters & 0xffff).sum())
                stats['evals'] = stats.get('evals', 0) + b
        else:
            value, grad, status, flags = gp.value_grad(X.reshape(b, q, d), want_flags=project_fallback is not None)
            if stats is not None:
                stats['evals'] = stats.get('evals', 0) + b
        grad = grad.reshape(b, q * d)
        value = torch.where(status == 0, value, torch.full_like(value, float('inf')))
        if project_fallback is not None and bool(flags.any()):
            for i in torch.nonzero(flags).reshape(-1).tolist():
                v, gr = project_fallback(X[i].cpu().numpy())
                value[i] = float(v)
                grad[i] = torch.as_tensor(gr, dtype=grad.dtype, device=grad.device)
        return value, grad
    fun.wants_index = p_state is not None
    return fun


def optimize_multistart(gp, X_starts, lower, upper, index_base=0, group=None, project_fallback=None,
                        warm_start=False, stats=None, method='lbfgs', **kwargs):
    """Lock-step descent from the local shard of starts X_starts [B, q, d], then the global best
    over all ranks (one all-gather of a (value, index, X) record, see multistart.select_best).
    Returns (LockstepResult of the local shard, (best_value, best_index, best_X [q, d]))."""
    from . import engine, multistart
    B, q, d = X_starts.shape
    X0 = torch.as_tensor(X_starts, dtype=torch.float64, device=gp.device).reshape(B, q * d)
    lo = torch.as_tensor(lower, dtype=torch.float64, device=gp.device)
    hi = torch.as_tensor(upper, dtype=torch.float64, device=gp.device)
    if lo.numel() == d:
        lo, hi = lo.repeat(q), hi.repeat(q)
    plain = not warm_start and stats is None
    if method == 'newton':
        # exact second-order steps on the batched Hessian, then the first-order method polishes
        # the runs the Newton iteration left unconverged (it stops at the first rejected step)
        fun = acquisition_fun(gp, q, project_fallback, stats=stats)
        nk = {k: v for k, v in kwargs.items() if k in ('max_iter', 'pgtol', 'ftol')}
        if plain and q * d <= NEWTON_MAX_N:
            res = minimize_acquisition_rounds(gp, q, X0, lo, hi, method='newton', project_fallback=project_fallback, **nk)
        else:
            newton = minimize_lockstep_newton_device if q * d <= NEWTON_MAX_N else minimize_lockstep_newton
            res = newton(fun, acquisition_hess_fun(gp, q), X0, lo, hi, **nk)
        todo = torch.nonzero(~res.converged & torch.isfinite(res.fun)).reshape(-1)
        if todo.numel() > 0:
            pk = {k: v for k, v in kwargs.items() if k != 'max_iter'}
            if plain:
                r2 = minimize_acquisition_rounds(gp, q, res.x[todo], lo, hi, project_fallback=project_fallback, **pk)
            else:
                r2 = minimize_lockstep_device(fun, res.x[todo], lo, hi, **pk)
            res.x[todo], res.fun[todo], res.grad[todo] = r2.x, r2.fun, r2.grad
            res.nit[todo] += r2.nit
            res.nfev[todo] += r2.nfev
            res.converged[todo] = r2.converged
    elif plain:
        res = minimize_acquisition_rounds(gp, q, X0, lo, hi, project_fallback=project_fallback, **kwargs)
    else:
        kw = dict(kwargs)
        kw.setdefault('max_iter', 2000)
        res = minimize_lockstep_device(acquisition_fun(gp, q, project_fallback, n_runs=B, warm_start=warm_start, stats=stats),
                                       X0, lo, hi, **kw)
    fv = torch.where(torch.isfinite(res.fun), res.fun, torch.full_like(res.fun, float('nan')))
    bv, bi = engine.argmin(fv, index_base)
    local = torch.clamp(bi - index_base, min=0)
    Xb = res.x.index_select(0, local.reshape(1)).reshape(-1)
    rec = multistart.make_record(bv, bi, Xb)
    best_value, best_index, best_X = multistart.select_best(rec, group)
    return res, (best_value, best_index, best_X.reshape(q, d))
