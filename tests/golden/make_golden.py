"""Generates the golden fixtures of tests/golden with the CPU oracle.

    python tests/golden/make_golden.py

The reference stores no expected outputs for this path and its solvers cannot be run here, so
these vectors are produced by the oracle (the reference's formulation solved to machine precision)
and each one is certified before it is written: KKT residuals of the SDP below 1e-9 and
finite-difference agreement of gradient and Hessian.  Inputs are regenerable: the SDP instances
are the reference's own test inputs (tests/test_sdp.py:45-62, legacy numpy seed 0); the GP models
follow oracle/hart6.make_model (hart6 on [-1/2,1/2]^6, Matern-3/2, variance 1, lengthscale 0.5,
noise 1e-6, normalised outputs, training design RandomState(0)).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import gp as ogp, hart6, oei as ooei, sdp_ref  # noqa: E402


def cert_max(cert):
    return max(v for k, v in cert.items() if k not in ('pobj', 'dobj'))


def ref_sdp_inputs():
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


def main(only=None):
    if only is None:
        make_sdp_fixture()
    make_oei_fixtures(only)


def make_sdp_fixture():
    inst = ref_sdp_inputs()
    om = np.stack([o for o, _ in inst])
    fm = np.array([f for _, f in inst])
    vals, Ms, Ys = [], [], []
    for o, f in inst:
        v, M, Y, _ = sdp_ref.sdp_exact(o, f)
        assert cert_max(sdp_ref.kkt_certificate(o, f, M, Y)) < 1e-9
        vals.append(v), Ms.append(M), Ys.append(np.stack(Y))
    np.savez_compressed(os.path.join(HERE, 'sdp_reference_inputs.npz'), omega=om, fmin=fm,
                        value=np.array(vals), M=np.stack(Ms), Y=np.stack(Ys))
    print('sdp_reference_inputs: values', np.round(vals, 6))



def make_oei_fixtures(only=None):
    # (10, 20): the single-evaluation configuration of the benchmark table (q = 20 on the 10-point
    # initial design)
    for n, q, seed in [(10, 1, 101), (10, 3, 102), (25, 3, 103), (25, 8, 104), (310, 20, 105),
                       (10, 20, 106)]:
        if only is not None and (n, q) not in only:
            continue
        model = hart6.make_model(n)
        orc = ooei.OEIOracle(model)
        X = np.random.RandomState(seed).uniform(-0.5, 0.5, (q, 6))
        assert len(orc.project(X)[0]) == q
        omega = model.omega(X)
        value, M, Y, C = sdp_ref.sdp_exact(omega, model.fmin)
        assert cert_max(sdp_ref.kkt_certificate(omega, model.fmin, M, Y)) < 1e-9
        _, grad = model.inner_and_grad(X, M)
        hess = orc.acquisition_hessian(X.ravel())
        # finite-difference certification of gradient (value) and Hessian (gradient)
        h = 1e-5
        x0 = X.ravel()
        k = min(6, x0.size)
        for j in range(k):
            e = np.zeros_like(x0)
            e[j] = h
            vp = orc.acquisition(x0 + e)
            vm = orc.acquisition(x0 - e)
            assert abs((vp[0][0] - vm[0][0]) / (2 * h) - grad.ravel()[j]) < 1e-6 * (1 + np.abs(grad).max())
            assert np.abs((vp[1] - vm[1]) / (2 * h) - hess[j]).max() < 1e-5 * (1 + np.abs(hess).max())
        out = dict(n=n, q=q, X_train=model.X, y_train=model.y, variance=1.0, lengthscale=0.5,
                   noise=model.noise, fmin=model.fmin, X=X, omega=omega, value=value, M=M,
                   gradient=grad.ravel(), hessian=hess)
        if q <= 8:
            out['domega'] = model.domega(X)
            out['dM'] = sdp_ref.solution_derivative(M, Y, C, out['domega'])
        np.savez_compressed(os.path.join(HERE, 'oei_n%d_q%d.npz' % (n, q)), **out)
        print('oei_n%d_q%d: value %.12f' % (n, q, value))


def make_hessian_cfg3():
    """Rows 0, 10, 20, ... of the acquisition Hessian at the largest shape of the benchmark table
    (n = 1000, q = 50, d = 10: the oracle needs minutes here, so the GPU test reads this fixture),
    plus value and gradient."""
    model = hart6.make_model(1000, d=10, lengthscale=0.8)
    orc = ooei.OEIOracle(model)
    X = np.random.RandomState(107).uniform(-0.5, 0.5, (50, 10))
    assert len(orc.project(X)[0]) == 50
    value, grad = orc.acquisition(X)
    H = orc.acquisition_hessian(X.ravel())
    assert np.abs(H - H.T).max() < 1e-10 * np.abs(H).max()
    h = 1e-5                                   # finite-difference certification of two rows
    for j in (0, 250):
        e = np.zeros(500)
        e[j] = h
        gp_, gm_ = orc.acquisition(X.ravel() + e)[1], orc.acquisition(X.ravel() - e)[1]
        assert np.abs((gp_ - gm_) / (2 * h) - H[j]).max() < 1e-5 * (1 + np.abs(H).max())
    np.savez_compressed(os.path.join(HERE, 'hess_n1000_q50_d10.npz'), n=1000, q=50, d=10, lengthscale=0.8,
                        X=X, value=value[0], gradient=grad, rows=np.arange(0, 500, 10), H_rows=H[::10],
                        H_diag=np.diag(H).copy(), H_absmax=np.abs(H).max())
    print('hess_n1000_q50_d10: value %.12f' % value[0])


if __name__ == '__main__':
    if sys.argv[1:] == ['cfg3-hessian']:
        make_hessian_cfg3()
        sys.exit(0)
    # optional arguments "n,q" restrict the run to those fixtures
    sel = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
    main(sel or None)
