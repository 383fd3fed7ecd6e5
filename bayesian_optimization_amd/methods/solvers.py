"""Non-linear optimiser adapters with the dispatcher signature of the reference's
methods/solvers.py:151-175.  They are callers of the hot path: they only consume
`bo.acquisition(x) -> ([value], gradient)` and `bo.acquisition_hessian(x) -> [qd, qd]`.

'bfgs'    scipy L-BFGS-B, as solvers.py:10-21.
'knitro'  the reference's SQP with exact Hessians is proprietary and absent; the same role (the
          only consumer of acquisition_hessian, solvers.py:56-69) is filled by scipy's bounded
          trust-region method 'trust-constr' with the exact Hessian, selected as 'newton'.
"""
from collections import namedtuple

import numpy as np
import scipy.optimize

OUTPUT_LEVEL = 0

Result = namedtuple('Result', 'status nit success message')


def bfgs_solve(x_init, bounds, hessian, bo):
    res = scipy.optimize.minimize(fun=bo.acquisition, x0=x_init, method='L-BFGS-B', jac=True,
                                  bounds=bounds)
    y = res.fun if isinstance(res.fun, float) else np.asarray(res.fun).reshape(-1)[0]
    return res.x, y, res


def newton_solve(x_init, bounds, hessian, bo):
    """Bounded second-order solve using the exact acquisition Hessian (role of solvers.py:71-141)."""
    last = {}           # value and gradient of one iterate come from one device evaluation

    def both(x):
        if last.get('x') is None or not np.array_equal(last['x'], x):
            value, grad = bo.acquisition(x)
            last['x'], last['value'], last['grad'] = np.array(x, copy=True), float(value[0]), grad
        return last

    fun = lambda x: both(x)['value']
    jac = lambda x: both(x)['grad']
    hess = (lambda x: bo.acquisition_hessian(x)) if hessian else scipy.optimize.SR1()
    res = scipy.optimize.minimize(fun, x_init, method='trust-constr', jac=jac, hess=hess,
                                  bounds=scipy.optimize.Bounds(bounds[:, 0], bounds[:, 1]),
                                  options={'verbose': 0, 'maxiter': 200})
    status = Result(status=res.status, nit=res.nit, success=bool(res.success), message=str(res.message))
    return res.x, float(res.fun), status


def solve(X_init, bounds, hessian, bo, solver):
    x_init = X_init.flatten()
    if solver == 'bfgs':
        x, y, status = bfgs_solve(x_init=x_init, bounds=bounds, hessian=hessian, bo=bo)
    elif solver == 'newton':
        x, y, status = newton_solve(x_init=x_init, bounds=bounds, hessian=hessian, bo=bo)
    elif solver == 'knitro':
        raise NameError("KNITRO is proprietary and not available; use nl_solver='newton' "
                        "(exact-Hessian trust region) or 'bfgs'")
    else:
        assert False, 'Invalid nonlinear solver choice!'
    k = x.size // bo.dim
    return x.reshape(k, bo.dim), y, status
