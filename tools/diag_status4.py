"""Diagnostic: runs the stand-alone SDP entry point on the padded Q = 32 instance (q = 21, 25) and
on q = 32 with whatever library OEI_LIB_OVERRIDE names, and prints status / iteration words next
to the CPU restatement's value."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hart6, reduced_newton  # noqa: E402


def main():
    from bayesian_optimization_amd import DeviceGP, sdp_batched
    from oracle import gp as ogp
    om = hart6.make_model(60)
    dm = DeviceGP(om.X, om.y, ogp.KERNEL_MATERN32, om.kernel.variance, om.kernel.lengthscales, om.noise,
                  floor_mode=om.kernel.floor_mode)
    print('library:', os.environ.get('OEI_LIB_OVERRIDE', 'default'))
    for q in (17, 21, 25, 29, 32):
        Xs = np.random.RandomState(31 + q).uniform(-.5, .5, (3, q, 6))
        omg = dm.omega(Xs)
        value, M, _, status, iters = [t.cpu().numpy() if t is not None else None for t in sdp_batched(omg, dm.fmin, want_Y=False)]
        ref = np.array([reduced_newton.solve(omg[b].cpu().numpy(), dm.fmin)[0] for b in range(3)])
        print('q=%d status=%s newton=%s sweeps=%s relerr=%s' % (
            q, status.tolist(), (iters & 0xffff).tolist(), (iters >> 16).tolist(),
            ['%.1e' % abs((value[b] - ref[b]) / ref[b]) for b in range(3)]), flush=True)


if __name__ == '__main__':
    main()
