"""Oracle (test infrastructure): the CPU legs of the benchmark.

Times the CPU restatement of the reference's evaluation path (oei.py:19-64 with the first-order
solve at eps = 1e-5 of sdp.py:51) on the host cores of the box the benchmark runs on, on a bounded
sample of the benchmark's own candidates.  Four legs:

  value           value+gradient, one candidate at a time on one core: the reference's own
                  structure (bo.py:142-175 is a sequential loop)
  all_cores       the same evaluations spread over a process pool
  same_algorithm  value+gradient with the reduced Newton solve the GPU path uses (numpy, one core):
                  separates what the algorithm buys from what the hardware buys
  hessian         acquisition_hessian with the dense solve of the reference's KKT system
                  (sdp.py:159-234), one core

Only bench.py's cpu_baseline leg and the tests call this.
"""
import multiprocessing as mp
import os
import time

import numpy as np

from . import gp as ogp
from . import hart6, oei as ooei, reduced_newton

_STATE = {}


def _limit_threads():
    try:
        from threadpoolctl import threadpool_limits
        _STATE['tp'] = threadpool_limits(limits=1)
    except Exception:
        pass


def _init_worker(n):
    for k in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):
        os.environ[k] = '1'
    _limit_threads()
    _STATE['orc'] = ooei.OEIOracle(hart6.make_model(n))


def _eval_first_order(X):
    v, g = _STATE['orc'].acquisition(X, first_order=True)
    return float(v[0])


def _timed(fn, items, budget_s):
    done, t0 = 0, time.time()
    for x in items:
        fn(x)
        done += 1
        if time.time() - t0 > budget_s:
            break
    return done, time.time() - t0


def run_all(n, q, X_sample, budget_s=24.0, pool_cores=None):
    """X_sample [m, q, d] candidates of the benchmark workload.  Returns the cpu_baseline object."""
    ncpu = os.cpu_count() or 1
    # the box's CPU share for one GPU is 16 cores even where os.cpu_count() shows the whole machine
    cores = pool_cores or max(1, min(16, ncpu))
    model = hart6.make_model(n)
    orc = ooei.OEIOracle(model)
    _limit_threads()
    m = X_sample.shape[0]

    done1, dt1 = _timed(lambda X: orc.acquisition(X, first_order=True), X_sample, 0.30 * budget_s)
    per_eval = dt1 / done1

    # all cores: enough evaluations for about a third of the budget per worker, none of them in
    # the one-core sample
    per_worker = max(1, int(0.30 * budget_s / per_eval))
    n_par = min(m - done1, per_worker * cores) if m > done1 else 0
    par = None
    if n_par > 0 and cores > 1:
        ctx = mp.get_context('spawn')              # never fork a process that holds a GPU context
        with ctx.Pool(cores, initializer=_init_worker, initargs=(n,)) as pool:
            pool.map(_eval_first_order, [X_sample[done1]] * cores)          # start-up and imports, untimed
            t0 = time.time()
            pool.map(_eval_first_order, list(X_sample[done1:done1 + n_par]), chunksize=1)
            dtp = time.time() - t0
        par = {'value': n_par / dtp, 'unit': 'evals/s', 'cores': cores,
               'sample': '%d candidate batches over a pool of %d processes; %.1f s' % (n_par, cores, dtp)}

    def same_alg(X):
        omega = model.omega(X)
        value, M, _, _, _ = reduced_newton.solve(omega, model.fmin)
        model.inner_and_grad(X, M)

    done3, dt3 = _timed(same_alg, X_sample, 0.12 * budget_s)
    done4, dt4 = _timed(lambda X: orc.acquisition_hessian(X.ravel()), X_sample, 0.2 * budget_s)

    return {
        'value': done1 / dt1, 'unit': 'evals/s', 'cores': 1, 'kind': 'port',
        'sample': '%d candidate batches (q=%d, n=%d) of the same workload, one at a time, numpy '
                  'restatement of the reference path with the first-order SDP solve at eps=1e-5; %.1f s'
                  % (done1, q, n, dt1),
        'host_cpu_count': ncpu,
        'all_cores': par,
        'same_algorithm': {'value': done3 / dt3, 'unit': 'evals/s', 'cores': 1,
                           'sample': '%d candidate batches, numpy version of the reduced Newton solve '
                                     'the GPU path uses; %.1f s' % (done3, dt3)},
        'hessian': {'value': done4 / dt4, 'unit': 'Hessians/s', 'cores': 1,
                    'sample': '%d candidate batches, acquisition_hessian with the dense solve of the '
                              "reference's KKT sensitivity system; %.1f s" % (done4, dt4)},
    }
