"""Host-side BO base class with the attribute and method names of the reference's methods/bo.py.

The reference's `BO` subclasses gpflow's `GPR` (bo.py:15-36); here the GP lives on the GPU behind
`engine.DeviceGP` and this class only keeps the data, the kernel description and the options, and
rebuilds the device model when any of them changes.  What callers of the hot path read stays
available under the same names: `.dim`, `.bounds`, `.X.value`, `.Y.value`, `.options`,
`.likelihood.variance.value`, `.kern.lengthscales.value`, `predict_f`.

The outer loop (`bayesian_optimization`, `get_suggestion`) follows bo.py:38-192.  It is the caller
of the hot path, not part of it; the multistart loop evaluates all starts' initial points in one
batched GPU call and then runs the reference's bounded quasi-Newton solver per start.
"""
import copy
import logging

import numpy as np

from .. import engine
from ..kernels import Param
from . import sdp as sdp_module
from .solvers import solve


class _Holder:
    def __init__(self, value):
        self.value = value


class _Likelihood:
    def __init__(self, variance):
        self.variance = Param(variance)


class BO(object):
    def __init__(self, options):
        self.bounds = np.asarray(options['objective'].bounds, dtype=np.float64).copy()
        self.dim = self.bounds.shape[0]
        if 'mean_function' not in options:
            options['mean_function'] = None
        if options['mean_function'] is not None:
            raise NotImplementedError('only the zero mean function of the reference runs is supported')
        self.kern = options['kernel']
        self.options = options.copy()
        self._X = _Holder(np.zeros((0, self.dim)))
        self._Y = _Holder(np.zeros((0, 1)))
        self.likelihood = _Likelihood(1.0)
        # bo.py:33-36: fix the noise if its value is provided
        if self.options.get('noise') is not None:
            self.likelihood.variance = Param(self.options['noise'])
            self.likelihood.variance.fixed = True
        self._gp = None
        self._gp_key = None

    # ---- data: assigning X or Y invalidates the device model (bo.py:60-61,116-117) -------------
    @property
    def X(self):
        return self._X

    @X.setter
    def X(self, value):
        self._X = _Holder(np.ascontiguousarray(value, dtype=np.float64))
        self._gp = None

    @property
    def Y(self):
        return self._Y

    @Y.setter
    def Y(self, value):
        self._Y = _Holder(np.ascontiguousarray(value, dtype=np.float64))
        self._gp = None

    def _key(self):
        return (float(self.kern.variance.value), tuple(np.atleast_1d(self.kern.lengthscales.value).tolist()),
                float(self.likelihood.variance.value), self.kern.kind, self.kern.floor_mode)

    def device_model(self):
        """The factorised GP on the GPU for the current data and hyper-parameters."""
        key = self._key()
        if self._gp is None or self._gp_key != key:
            if self._gp is not None:
                self._gp.close()
            self._gp = engine.DeviceGP(self.X.value, self.Y.value[:, 0], self.kern.kind,
                                       float(self.kern.variance.value),
                                       np.atleast_1d(self.kern.lengthscales.value),
                                       float(self.likelihood.variance.value),
                                       floor_mode=self.kern.floor_mode)
            self._gp_key = key
        return self._gp

    def predict_f(self, Xnew):
        """gpflow GPR.predict_f: (mean [m,1], marginal variance [m,1])."""
        mean, var = self.device_model().predict_f(np.asarray(Xnew, dtype=np.float64))
        return mean.cpu().numpy()[:, None], var.cpu().numpy()[:, None]

    # ---- outer loop -----------------------------------------------------------------------------
    def bayesian_optimization(self):
        """bo.py:38-127."""
        objective = copy.copy(self.options['objective'])
        X0 = self.random_sample(self.bounds, self.options['initial_size'])
        ret = objective.f(X0)
        y0, X0 = ret if isinstance(ret, tuple) else (ret, X0)
        self.X = X0
        self.Y = self.normalize(y0[:, 0:1])
        X_all, y_all = X0, y0
        log = logging.getLogger('evals')
        for i in range(len(X0)):
            log.info('X:' + str(X0[i, :]) + ' y: ' + str(y0[i, :]))
        for i in range(self.options['iterations']):
            self.optimize_restarts(restarts=self.options['model_restarts'])
            logging.getLogger('').info('#Iteration:' + str(i + 1))
            X_new = self.get_suggestion(self.options['batch_size'])
            ret = objective.f(X_new)
            y_new, X_new = ret if isinstance(ret, tuple) else (ret, X_new)
            X_all = np.concatenate((X_all, X_new))
            y_all = np.concatenate((y_all, y_new))
            self.X = X_all
            self.Y = self.normalize(y_all[:, 0:1])
            for j in range(len(X_new)):
                log.info('X:' + str(X_new[j, :]) + ' y: ' + str(y_new[j, :]))
        return X_all, y_all

    def get_suggestion(self, batch_size):
        """bo.py:129-192: multistart descent on the acquisition function."""
        if self.options['nl_solver'] in ('lockstep', 'lockstep-newton'):
            return self._get_suggestion_lockstep(batch_size)
        X, y = None, None
        bounds_tiled = np.tile(self.bounds, (batch_size, 1))
        for j in range(self.options['opt_restarts']):
            sdp_module.reset_warm_starting()
            X_init = self.random_sample(self.bounds, batch_size)
            y_init = self.acquisition(X_init)[0]
            X0, y0, status = solve(X_init=X_init, bounds=bounds_tiled, hessian=self.options['hessian'],
                                   bo=self, solver=self.options['nl_solver'])
            if X is None or y0 < y:
                X, y = X0, y0
            logging.getLogger('opt').info('##Opt_it:' + str(j + 1) + ' Val:' + '%.2e' % y0 +
                                          ' Diff:' + '%.2e' % float(np.ravel(y_init - y0)[0]) + ' It:' + str(status.nit))
            if not status.success:
                logging.getLogger('opt').warning('NLP Warning:' + str(status.message))
        assert X is not None
        return X

    def _get_suggestion_lockstep(self, batch_size):
        """The multistart loop of bo.py:129-192 with all `opt_restarts` descents advanced together
        (lockstep.py): same random starts in the same order of the np.random stream, the best
        final point wins."""
        from .. import lockstep
        starts = np.stack([self.random_sample(self.bounds, batch_size)
                           for _ in range(self.options['opt_restarts'])])
        # 'lockstep' is the batched counterpart of the reference's 'bfgs' (which ignores the
        # `hessian` option, solvers.py:10-17); 'lockstep-newton' that of its SQP solver, which is
        # handed exact Hessians when options['hessian'] is set (run.py --hessian, solvers.py:56-69)
        newton = self.options['nl_solver'] == 'lockstep-newton' and bool(self.options.get('hessian', 1))
        res, best = lockstep.optimize_multistart(
            self.device_model(), starts, self.bounds[:, 0], self.bounds[:, 1],
            project_fallback=lambda x: (lambda r: (r[0][0], r[1]))(self.acquisition(x)),
            method='newton' if newton else 'lbfgs')
        log = logging.getLogger('opt')
        fun, nit = res.fun.cpu().numpy(), res.nit.cpu().numpy()
        for j in range(len(fun)):
            log.info('##Opt_it:' + str(j + 1) + ' Val:' + '%.2e' % fun[j] + ' It:' + str(int(nit[j])))
        # how the runs ended (scipy reports this per start as status.message, bo.py:176-177)
        stopped = len(fun) - int(res.converged.sum().item())
        log.info('##Opt_runs:' + str(len(fun)) + ' converged:' + str(len(fun) - stopped) + ' stopped at a limit:' + str(stopped))
        if stopped:
            log.warning('NLP Warning: %d of %d runs stopped at the iteration or line-search limit' % (stopped, len(fun)))
        assert int(best[1].item()) >= 0
        return best[2].cpu().numpy()

    def optimize_restarts(self, restarts=1, **kwargs):
        """Hyper-parameter fit of bo.py:194-221: best of `restarts` maximum-likelihood runs (MAP
        with `options['priors']`), objective and gradient evaluated on the GPU (hyperfit.py).
        The fit runs when `options['fit_hyperparameters']` is set (the driver sets it); without
        it the hyper-parameters stay at their current values."""
        if not self.options.get('fit_hyperparameters') or restarts is None:
            return None
        from .. import hyperfit
        res = hyperfit.fit(self.X.value, self.Y.value[:, 0], self.kern.kind,
                           float(self.kern.variance.value), np.atleast_1d(self.kern.lengthscales.value),
                           float(self.likelihood.variance.value), floor_mode=self.kern.floor_mode,
                           ard=self.kern.ARD, fixed_noise=self.likelihood.variance.fixed,
                           priors=bool(self.options.get('priors')), restarts=restarts)
        self.kern.variance = Param(res['variance'])
        self.kern.lengthscales = Param(res['lengthscales'] if self.kern.ARD else res['lengthscales'][0])
        if not self.likelihood.variance.fixed:
            self.likelihood.variance = Param(res['noise'])
        self._gp = None
        logging.getLogger('model').info('variance %.6g lengthscales %s noise %.6g nlml %.6f'
                                        % (res['variance'], res['lengthscales'], res['noise'], res['nlml']))
        return res

    @staticmethod
    def random_sample(bounds, k):
        """bo.py:223-239: k points uniform in the box, one coordinate at a time (same stream use)."""
        n = bounds.shape[0]
        X = np.zeros((k, n))
        for i in range(n):
            X[:, i] = np.random.uniform(bounds[i, 0], bounds[i, 1], k)
        return X

    def normalize(self, Y):
        """bo.py:241-255."""
        Y_ = Y.copy()
        if self.options['normalize_Y'] and np.std(Y[:, 0]) > 0:
            Y_[:, 0] = (Y[:, 0] - np.mean(Y[:, 0])) / np.std(Y[:, 0])
        return Y_
