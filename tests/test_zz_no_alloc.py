"""The entry point that allocates nothing (oei_value_grad_batched_noalloc) against the ordinary one,
through the C ABI: two calls in a row, the second with a smaller input, results bitwise equal to
oei_value_grad_batched; oversize requests are rejected; no device memory is allocated by the call."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import gp as ogp, hart6

pytestmark = pytest.mark.gpu


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def test_no_alloc_entry_point_matches_and_reuses_scratch(cuda):
    from bayesian_optimization_amd import DeviceGP
    from bayesian_optimization_amd._lib import lib
    om = hart6.make_model(310)
    dm = DeviceGP(om.X, om.y, ogp.KERNEL_MATERN32, om.kernel.variance, om.kernel.lengthscales, om.noise,
                  floor_mode=om.kernel.floor_mode)
    rs = np.random.RandomState(71)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for B, q in [(150, 20), (37, 20), (5, 7)]:          # large first, then smaller inputs into the same scratch
        Xh = rs.uniform(-.5, .5, (B, q, 6))
        if B == 37:
            Xh[3, 2] = om.X[5] + 1e-5                   # a duplicate of a training point: the flag must come through
        X = torch.as_tensor(Xh, device=cuda)
        ref = dm.value_grad(X)
        out = (torch.full((B,), 7.0, dtype=torch.float64, device=cuda),
               torch.full((B, q, 6), 7.0, dtype=torch.float64, device=cuda),
               torch.full((B,), 9, dtype=torch.int32, device=cuda),
               torch.full((B,), 9, dtype=torch.int32, device=cuda))
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
        rc = lib.oei_value_grad_batched_noalloc(dm._h, B, q, _ptr(X), _ptr(out[0]), _ptr(out[1]),
                                                _ptr(out[2]), _ptr(out[3]), stream)
        torch.cuda.synchronize()
        after = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
        assert rc == 0, lib.oei_last_error()
        assert before == after                           # neither torch nor the driver handed out memory
        assert int((ref[2] != 0).sum()) == 0
        for a, b in zip(ref, out):
            assert torch.equal(a, b)
        if B == 37:
            assert int(out[3][3]) == 1 and int(out[3].sum()) == 1
    # the engine's wrapper takes the same route
    X = torch.as_tensor(rs.uniform(-.5, .5, (11, 20, 6)), device=cuda)
    a, b = dm.value_grad(X), dm.value_grad(X, no_alloc=True)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_no_alloc_entry_point_rejects_oversize(cuda):
    from bayesian_optimization_amd import DeviceGP
    from bayesian_optimization_amd._lib import lib
    om = hart6.make_model(310)
    dm = DeviceGP(om.X, om.y, ogp.KERNEL_MATERN32, om.kernel.variance, om.kernel.lengthscales, om.noise,
                  floor_mode=om.kernel.floor_mode)
    B, q = 4096, 20
    X = torch.zeros((B, q, 6), dtype=torch.float64, device=cuda)
    v = torch.empty(B, dtype=torch.float64, device=cuda)
    g = torch.empty((B, q, 6), dtype=torch.float64, device=cuda)
    st = torch.empty(B, dtype=torch.int32, device=cuda)
    rc = lib.oei_value_grad_batched_noalloc(dm._h, B, q, _ptr(X), _ptr(v), _ptr(g), _ptr(st), None, None)
    assert rc == -1 and b'scratch' in lib.oei_last_error()
    # empty batch: nothing to do, no error
    assert lib.oei_value_grad_batched_noalloc(dm._h, 0, q, None, None, None, None, None, None) == 0
