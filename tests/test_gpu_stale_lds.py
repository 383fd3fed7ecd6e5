"""Results must not depend on what earlier kernels left in LDS.

LDS is not cleared between kernels.  A solver kernel that multiplies a never-written padding
entry by zero is correct as long as the stale bits are a number and returns NaN (or a breakdown
status) when they happen to be an Inf or NaN pattern -- which once surfaced as one wrong Hessian
in a 3 500-case sweep, after a particular pair of earlier calls.  `make poison` builds the library
with every kernel filling its dynamic LDS with NaN first; this test runs the entry points through
that build in a child process and compares with the regular library, bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import gp as ogp, hart6

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POISON = os.path.join(ROOT, 'bayesian_optimization_amd', 'liboei_hip_poison.so')

CASES = [  # q, d, n, B: even and odd q on both solver paths, sliced and unsliced Hessians
    (20, 6, 310, 3), (8, 2, 40, 2), (21, 3, 64, 2), (32, 5, 64, 1),
    (34, 2, 70, 2), (50, 1, 3, 5), (50, 3, 130, 2), (64, 2, 65, 1), (33, 4, 63, 130),
]

CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from oracle import gp as ogp, hart6
from bayesian_optimization_amd import DeviceGP, sdp_batched
out = {}
for i, (q, d, n, B) in enumerate(%(cases)r):
    om = hart6.make_model(n, d=d, train_seed=100 + i, kind=ogp.KERNEL_MATERN32 if i %% 2 else ogp.KERNEL_RBF,
                          floor_mode=1 if i %% 2 else 0, lengthscale=0.6, noise=1e-4)
    dm = DeviceGP(om.X, om.y, om.kernel.kind, om.kernel.variance, om.kernel.lengthscales, om.noise,
                  floor_mode=om.kernel.floor_mode)
    X = np.random.RandomState(i).uniform(-.5, .5, (B, q, d))
    v, g, st, fl = dm.value_grad(X)
    H, hst = dm.hessian(X[:min(B, 6)])
    omg = dm.omega(X[:2])
    sv, M, Y, sst, _ = sdp_batched(omg, dm.fmin)
    for k, t in (('v', v), ('g', g), ('st', st), ('H', H), ('hst', hst), ('sv', sv), ('M', M), ('sst', sst)):
        out['%%d_%%s' %% (i, k)] = t.cpu().numpy()
    dm.close()
np.savez(sys.argv[1], **out)
'''


@pytest.mark.gpu
def test_results_do_not_depend_on_stale_lds(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    assert os.path.exists(POISON), 'build the test variant first: make -C bayesian_optimization_amd/csrc poison'
    script = CHILD % {'root': ROOT, 'cases': CASES}
    res = {}
    for name, lib in (('regular', None), ('poison', POISON)):
        env = dict(os.environ)
        env.pop('OEI_LIB_OVERRIDE', None)
        if lib:
            env['OEI_LIB_OVERRIDE'] = lib
        path = str(tmp_path / (name + '.npz'))
        subprocess.run([sys.executable, '-c', script, path], env=env, check=True, timeout=300)
        res[name] = np.load(path)
    for key in res['regular'].files:
        a, b = res['regular'][key], res['poison'][key]
        if key.endswith('st'):
            assert not a.any() and not b.any(), (key, a, b)          # every solve converged, in both builds
        else:
            assert np.isfinite(b).all(), key
            assert a.tobytes() == b.tobytes(), key                   # and to the same bits
