"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Tolerances are those of the task statement: 1e-5 relative on the OEI value, 1e-4 on the gradient,
1e-3 on Hessian entries -- all far looser than what two machine-precision solutions differ by, so
most assertions here use much tighter bounds and say so.
"""
import numpy as np
import pytest
import torch

from oracle import gp as ogp, hart6, oei as ooei, sdp_ref

pytestmark = pytest.mark.gpu


def _mk(n, d=6, kind=ogp.KERNEL_MATERN32, floor_mode=None, seed=0, lengthscale=0.5):
    from bayesian_optimization_amd import DeviceGP
    om = hart6.make_model(n, d=d, train_seed=seed, kind=kind, floor_mode=floor_mode,
                          lengthscale=lengthscale)
    dm = DeviceGP(om.X, om.y, kind, om.kernel.variance, om.kernel.lengthscales, om.noise,
                  floor_mode=om.kernel.floor_mode)
    return om, dm


def _ref_sdp_inputs():
    """The reference's own SDP test inputs (tests/test_sdp.py:45-62): legacy seed-0 stream."""
    np.random.seed(0)
    k = 5
    out = []
    for _ in range(10):
        tmp = np.random.randn(k, k)
        sigma = tmp.dot(tmp.T) + 0.01 * np.eye(k)
        mu = np.random.randn(k, 1)
        omega = np.zeros((k + 1, k + 1))
        omega[0:k, 0:k] = sigma + mu.dot(mu.T)
        omega[-1, 0:k] = mu.flatten()
        omega[0:k, -1] = mu.flatten()
        omega[-1, -1] = 1
        fmin = np.random.randn(1)
        out.append((omega, float(fmin[0])))
    return out


@pytest.mark.parametrize('n,kind,fm', [(10, ogp.KERNEL_MATERN32, 1), (70, ogp.KERNEL_MATERN32, 2),
                                        (310, ogp.KERNEL_MATERN32, 1), (150, ogp.KERNEL_RBF, 0)])
def test_model_factor_and_fmin(cuda, n, kind, fm):
    om, dm = _mk(n, kind=kind, floor_mode=fm, lengthscale=0.5 if kind else 0.3)
    L = dm.factor()
    assert np.allclose(np.tril(L), om.L, rtol=1e-9, atol=1e-11)
    assert abs(dm.fmin - om.fmin) <= 1e-8 * (1 + abs(om.fmin))


def test_predict_f(cuda):
    om, dm = _mk(70)
    Xs = np.random.RandomState(3).uniform(-.5, .5, (37, 6))
    mean, var = dm.predict_f(Xs)
    m0, v0 = om.predict_f(Xs)
    assert np.allclose(mean.cpu().numpy(), m0[:, 0], rtol=1e-8, atol=1e-9)
    assert np.allclose(var.cpu().numpy(), v0[:, 0], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize('n,q', [(10, 1), (25, 3), (70, 7), (310, 20)])
def test_omega(cuda, n, q):
    om, dm = _mk(n)
    Xs = np.random.RandomState(5).uniform(-.5, .5, (6, q, 6))
    got = dm.omega(Xs).cpu().numpy()
    for b in range(Xs.shape[0]):
        assert np.allclose(got[b], om.omega(Xs[b]), rtol=1e-9, atol=1e-10)


def test_sdp_reference_inputs(cuda):
    """sdp.sdp on the reference's test instances: value, M against the reference-form optimum and
    the solver-independent KKT certificate on (M, Y)."""
    from bayesian_optimization_amd import sdp_batched
    inst = _ref_sdp_inputs()
    om = np.stack([o for o, _ in inst])
    fm = np.array([f for _, f in inst])
    value, M, Y, status, iters = [t.cpu().numpy() for t in sdp_batched(om, fm)]
    assert (status == 0).all()
    for b, (o, f) in enumerate(inst):
        cert = sdp_ref.kkt_certificate(o, f, M[b], list(Y[b]))
        worst = max(v for k, v in cert.items() if k not in ('pobj', 'dobj'))
        assert worst < 1e-9, cert
        obj, Mx, Yx, _ = sdp_ref.sdp_exact(o, f)
        assert abs(value[b] - obj) <= 1e-10 * abs(obj)          # task bar: 1e-5
        assert np.abs(M[b] - Mx).max() <= 1e-7 * np.abs(Mx).max()


def test_sdp_q1_closed_form(cuda):
    from bayesian_optimization_amd import sdp_batched
    rs = np.random.RandomState(1)
    mu, s2, fm = rs.randn(64), rs.uniform(0.01, 3, 64), rs.randn(64)
    om = np.zeros((64, 2, 2))
    om[:, 0, 0] = s2 + mu * mu
    om[:, 0, 1] = om[:, 1, 0] = mu
    om[:, 1, 1] = 1
    value = sdp_batched(om, fm)[0].cpu().numpy()
    exact = -0.5 * (np.sqrt(s2 + (mu - fm) ** 2) - (mu - fm))
    assert np.allclose(value, exact, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize('n,q,B', [(10, 3, 8), (25, 5, 8), (310, 20, 32)])
def test_value_grad_certificate_and_oracle(cuda, n, q, B):
    from bayesian_optimization_amd import sdp_batched
    om, dm = _mk(n)
    oracle = ooei.OEIOracle(om)
    Xs = np.random.RandomState(11).uniform(-.5, .5, (B, q, 6))
    value, grad, status, flags = [t.cpu().numpy() for t in dm.value_grad(Xs)]
    assert (status == 0).all() and (flags == 0).all()
    # full-size, solver-independent: the SDP solution behind each value satisfies KKT
    omg = dm.omega(Xs)
    v2, M, Y, st2, _ = [t.cpu().numpy() for t in sdp_batched(omg, dm.fmin)]
    assert np.allclose(v2, value, rtol=1e-11, atol=1e-13)
    for b in range(B):
        cert = sdp_ref.kkt_certificate(omg[b].cpu().numpy(), dm.fmin, M[b], list(Y[b]))
        assert max(v for k, v in cert.items() if k not in ('pobj', 'dobj')) < 1e-8, cert
        # gradient = d<Omega(X), M>/dX with M fixed (oei.py:66-77)
        _, g0 = om.inner_and_grad(Xs[b], M[b])
        assert np.abs(grad[b] - g0).max() <= 1e-7 * (1e-12 + np.abs(g0).max())   # task bar: 1e-4
    # oracle path end to end on a few candidates
    for b in range(min(B, 3)):
        v0, g0 = oracle.acquisition(Xs[b])
        assert abs(value[b] - v0[0]) <= 1e-9 * abs(v0[0])
        assert np.abs(grad[b].ravel() - g0).max() <= 1e-6 * np.abs(g0).max()


def test_project_flags(cuda):
    om, dm = _mk(25)
    rs = np.random.RandomState(13)
    Xs = rs.uniform(-.5, .5, (4, 5, 6))
    Xs[1, 2] = om.X[7] + 1e-4            # duplicate of a training point
    Xs[3, 4] = Xs[3, 0] + 1e-5           # mutual duplicate
    flags = dm.value_grad(Xs)[3].cpu().numpy()
    assert flags.tolist() == [0, 1, 0, 1]
    # the stand-alone duplicate test (what masks the Hessian of such a candidate) agrees, also on
    # a batch that spans many workgroups and for q > 32
    assert dm.project_flags(Xs).cpu().numpy().tolist() == [0, 1, 0, 1]
    for q in (1, 20, 33):
        Xb = rs.uniform(-.5, .5, (300, q, 6))
        hit = rs.rand(300) < 0.3
        for b in np.flatnonzero(hit):
            if q > 1 and b % 2:
                Xb[b, q - 1] = Xb[b, 0]          # (a small offset could straddle a rounding boundary of X/l)
            else:
                Xb[b, b % q] = om.X[b % 25] - 1e-4
        assert np.array_equal(dm.project_flags(Xb).cpu().numpy() != 0, hit)
        assert np.array_equal(dm.value_grad(Xb)[3].cpu().numpy() != 0, hit)


def test_argmin_ties(cuda):
    from bayesian_optimization_amd import argmin
    v = torch.tensor([3.0, -1.0, 2.0, -1.0, 5.0], dtype=torch.float64, device=cuda)
    bv, bi = argmin(v, index_base=100)
    assert bv.item() == -1.0 and bi.item() == 101


@pytest.mark.parametrize('n,q', [(25, 3), (25, 8), (310, 20)])
def test_solution_derivative_vs_dense_kkt(cuda, n, q):
    """sdp.solution_derivative: the reduced sensitivity solve against a dense solve of the
    reference's KKT system (sdp.py:159-234), for dOmega/dx directions and a generic direction
    with dOmega[-1,-1] != 0; plus dY for a single direction (sdp.py:195-204)."""
    from bayesian_optimization_amd import sdp_batched, solution_derivative_batched
    om, dm = _mk(n)
    rs = np.random.RandomState(17)
    Xs = rs.uniform(-.5, .5, (q, 6))
    omega = om.omega(Xs)
    value, M, Y, status, _ = [t.cpu().numpy() for t in sdp_batched(omega, om.fmin)]
    assert status[0] == 0
    C = sdp_ref.create_C(q + 1, om.fmin)
    dom = om.domega(Xs)
    D = rs.randn(q + 1, q + 1)
    D = D + D.T
    dom = np.concatenate([dom, D[:, :, None]], axis=2)
    dM_ref = sdp_ref.solution_derivative(M[0], list(Y[0]), C, dom)
    dM, _, st = solution_derivative_batched(M, Y, om.fmin, dom[None])
    dM = dM.cpu().numpy()[0]
    assert st.cpu().numpy()[0] == 0
    scale = np.abs(dM_ref).max(axis=(0, 1), keepdims=True)
    assert (np.abs(dM - dM_ref) / scale).max() < 1e-6         # task bar on the Hessian: 1e-3
    dM1_ref, dY_ref = sdp_ref.solution_derivative(M[0], list(Y[0]), C, D, return_dY=True)
    dM1, dY1, _ = solution_derivative_batched(M, Y, om.fmin, D[None, :, :, None], want_dY=True)
    dY1 = dY1.cpu().numpy()[0]
    assert np.abs(dM1.cpu().numpy()[0, :, :, 0] - dM1_ref[:, :, 0]).max() < 1e-6 * np.abs(dM1_ref).max()
    dYr = np.stack(dY_ref)
    assert np.abs(dY1 - dYr).max() < 1e-6 * np.abs(dYr).max()


@pytest.mark.parametrize('n,q,kind,fm', [(25, 3, ogp.KERNEL_MATERN32, 1), (70, 7, ogp.KERNEL_RBF, 0),
                                          (310, 20, ogp.KERNEL_MATERN32, 2)])
def test_domega(cuda, n, q, kind, fm):
    om, dm = _mk(n, kind=kind, floor_mode=fm)
    Xs = np.random.RandomState(19).uniform(-.5, .5, (3, q, 6))
    got = dm.domega(Xs).cpu().numpy()
    for b in range(3):
        ref = om.domega(Xs[b])
        assert np.abs(got[b] - ref).max() <= 1e-8 * np.abs(ref).max()


@pytest.mark.parametrize('n,q,kind,fm', [(25, 3, ogp.KERNEL_MATERN32, 1), (70, 5, ogp.KERNEL_RBF, 0),
                                          (310, 20, ogp.KERNEL_MATERN32, 1)])
def test_hessian_vs_oracle(cuda, n, q, kind, fm):
    """acquisition_hessian: the fused HIP pipeline against the oracle's restatement of
    oei.py:41-64 (dense KKT sensitivity + closed-form chain rule).  Task bar: 1e-3 on entries."""
    om, dm = _mk(n, kind=kind, floor_mode=fm)
    oracle = ooei.OEIOracle(om)
    B = 2
    Xs = np.random.RandomState(23).uniform(-.5, .5, (B, q, 6))
    H, status = dm.hessian(Xs)
    H = H.cpu().numpy()
    assert (status.cpu().numpy() == 0).all()
    for b in range(B):
        Href = oracle.acquisition_hessian(Xs[b].ravel())
        scale = np.abs(Href).max()
        assert np.abs(H[b] - H[b].T).max() <= 1e-8 * scale
        assert np.abs(H[b] - Href).max() <= 1e-6 * scale, np.abs(H[b] - Href).max() / scale


def test_best_record_ties_nan_and_empty(cuda):
    """oei_best_record: the arg-min's winner with its X row in one launch; ties go to the lowest
    index, NaN never wins, no finite value gives index -1."""
    from bayesian_optimization_amd.engine import best_record
    X = torch.arange(5 * 6, dtype=torch.float64, device=cuda).reshape(5, 2, 3)
    v = torch.tensor([3.0, -1.0, float('nan'), -1.0, 0.5], dtype=torch.float64, device=cuda)
    rec = best_record(v, X, index_base=100)
    assert rec[0].item() == -1.0 and rec[1:2].view(torch.int64).item() == 101
    assert torch.equal(rec[2:], X[1].reshape(-1))
    rec = best_record(torch.full((4,), float('nan'), dtype=torch.float64, device=cuda), X[:4], index_base=7)
    assert rec[1:2].view(torch.int64).item() == -1 and bool(torch.isnan(rec[0]))
    big = torch.randn(5000, dtype=torch.float64, device=cuda)
    Xb = torch.randn(5000, 4, dtype=torch.float64, device=cuda)
    rec = best_record(big, Xb)
    k = int(torch.argmin(big))
    assert rec[1:2].view(torch.int64).item() == k and torch.equal(rec[2:], Xb[k])
