"""The value + gradient path without the Z buffer (oei_set_z_free: second GEMM on the A rows, M11
applied per tile) against the default path: same results up to the order of the sums, for batch
sizes whose rows do and do not fill 64-row tiles with whole candidates, register-path and wide
sizes, and a batch that runs as two halves on two streams."""
import numpy as np
import pytest
import torch

from oracle import gp as ogp, hart6

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cuda():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


@pytest.mark.parametrize('n,q,d,B', [(310, 20, 6, 37), (310, 20, 6, 4200), (65, 1, 3, 130), (130, 7, 2, 19),
                                     (64, 16, 11, 9), (200, 32, 4, 5), (90, 33, 3, 4), (150, 50, 10, 3),
                                     (40, 64, 2, 3)])
def test_z_free_matches_default(cuda, n, q, d, B):
    from bayesian_optimization_amd import DeviceGP
    model = hart6.make_model(n, d=d, lengthscale=0.5 * (1 + 0.05 * d))
    gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, model.kernel.variance, model.kernel.lengthscales,
                  model.noise, floor_mode=model.kernel.floor_mode)
    X = torch.as_tensor(np.random.RandomState(5).uniform(-.5, .5, (B, q, d)), device=cuda)
    v0, g0, st0, fl0 = gp.value_grad(X)
    gp.set_z_free(True)
    v1, g1, st1, fl1 = gp.value_grad(X)
    gp.set_z_free(False)
    assert int((st0 != 0).sum()) == 0 and int((st1 != 0).sum()) == 0
    assert torch.equal(v0, v1) and torch.equal(fl0, fl1)          # the value does not go through the second GEMM
    scale = g0.abs().amax(dim=(1, 2), keepdim=True)
    assert float(((g1 - g0).abs() / scale).max()) < 1e-10
    gp.close()
