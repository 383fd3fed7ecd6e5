"""Command-line driver with the flags and output files of the reference's run.py:

    python -m bayesian_optimization_amd.run --seed=123 --function=hart6 --batch_size=20 \\
        --initial_size=10 --iterations=15 --noise=1e-6

writes `out/<function>_<algorithm>/<seed>.npz` (arrays X, Y), `arguments.pkl`, `fmin.txt`
(run.py:26-71, 105-133) and logs to `log/<job>/<seed>/` (bo.py:257-276).  Differences, all forced by what
is installed: `--nl_solver` defaults to `lockstep` (the reference's default, KNITRO, is proprietary):
all `opt_restarts` bounded L-BFGS runs of a BO iteration advanced together on the GPU, the batched
counterpart of `bfgs`; `lockstep-newton` is the batched consumer of the exact Hessian (`--hessian 1`),
which KNITRO is in the reference; `bfgs` and `newton` run one start after the other, the objective comes from `oracle/hart6.py`-equivalent
formulas in `benchmarks.py`, the kernels are this package's descriptions instead of gpflow objects,
and the GP hyper-parameters are fitted by `hyperfit.py` on the GPU.
"""
import argparse
import copy
import logging
import os
import pickle
import random
import time

import numpy as np


def setup_logging(job_name, seed):
    """Three log files per run, `log/<job>/<seed>/{evals,opt,model}.log`, bare messages, no
    propagation: bo.py:257-276 with logging.yaml:15-42."""
    folder = os.path.join('log', job_name, str(seed))
    os.makedirs(folder, exist_ok=True)
    for name in ('evals', 'model', 'opt'):
        logger = logging.getLogger(name)
        logger.setLevel(logging.INFO)
        logger.propagate = False
        for h in list(logger.handlers):
            logger.removeHandler(h)
        handler = logging.FileHandler(os.path.join(folder, name + '.log'), mode='w')
        handler.setFormatter(logging.Formatter('%(message)s'))
        logger.addHandler(handler)


def teardown_logging():
    """Closes the run's log files and gives the three loggers back their default behaviour."""
    for name in ('evals', 'model', 'opt'):
        logger = logging.getLogger(name)
        for h in list(logger.handlers):
            logger.removeHandler(h)
            h.close()
        logger.propagate = True


def run(options, seed, robust=False, save=False):
    from .methods.oei import OEI
    options = dict(options)
    options['kernel'] = copy.deepcopy(options['kernel'])
    options['seed'] = seed
    np.random.seed(seed)
    random.seed(seed)
    setup_logging(options['job_name'], seed)
    bo = OEI(options)
    try:
        start = time.time()
        X, Y = bo.bayesian_optimization()
        print('Done with:', bo.options['job_name'], 'seed:', seed,
              'Time:', '%.2f' % ((time.time() - start) / 60), 'min')
    except KeyboardInterrupt:
        raise
    except Exception:
        print('Experiment of', bo.options['job_name'], 'with seed', seed, 'failed')
        X, Y = None, None
        if not robust:
            teardown_logging()
            raise
    teardown_logging()
    if save:
        folder = os.path.join('out', bo.options['job_name'])
        os.makedirs(folder, exist_ok=True)
        path = os.path.join(folder, str(seed) + '.npz')
        if os.path.exists(path):
            os.remove(path)
        np.savez(path, X=X, Y=Y)
    return X, Y


def create_options(args):
    from . import benchmarks
    from .kernels import RBF, SafeMatern32
    functions = {'hart6': benchmarks.hart6}
    kernels = {'RBF': RBF, 'Matern32': SafeMatern32}
    options = vars(copy.copy(args))
    if options['algorithm'] != 'OEI':
        raise NotImplementedError("only --algorithm OEI is provided")
    options['objective'] = benchmarks.scale_function(functions[options['function']]())
    input_dim = options['objective'].bounds.shape[0]
    options['kernel'] = kernels[options['kernel']](input_dim, ARD=bool(options['ard']))
    options['job_name'] = options['function'] + '_' + options['algorithm']
    options['fit_hyperparameters'] = 1
    return options


def main(args):
    options = create_options(args)
    folder = os.path.join('out', options['job_name'])
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, 'arguments.pkl'), 'wb') as fh:
        pickle.dump(args, fh, pickle.HIGHEST_PROTOCOL)
    fmin = getattr(options['objective'], 'fmin', 0)
    np.savetxt(os.path.join(folder, 'fmin.txt'), np.array([fmin]))
    for seed in range(args.seed, args.seed + args.num_seeds):
        run(options, seed=seed, save=options['save'])


def create_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--function', default='hart6')
    parser.add_argument('--algorithm', default='OEI')
    parser.add_argument('--seed', type=int, default=123)
    parser.add_argument('--num_seeds', type=int, default=1)
    parser.add_argument('--save', type=int, default=1)
    parser.add_argument('--batch_size', type=int, default=5)
    parser.add_argument('--iterations', type=int, default=10)
    parser.add_argument('--initial_size', type=int, default=10)
    parser.add_argument('--model_restarts', type=int, default=20,
                        help='Random restarts when optimizing the Likelihood of the GP.')
    parser.add_argument('--opt_restarts', type=int, default=20,
                        help='Random restarts when optimizing the acquisition function.')
    parser.add_argument('--normalize_Y', type=int, default=1)
    parser.add_argument('--noise', type=float, help='Used to set the likelihood to a fixed value')
    parser.add_argument('--kernel', default='Matern32')
    parser.add_argument('--ard', type=int, default=0)
    parser.add_argument('--nl_solver', default='lockstep')
    parser.add_argument('--hessian', type=int, default=1)
    parser.add_argument('--priors', type=int, default=0)
    return parser


if __name__ == '__main__':
    main(create_parser().parse_args())
