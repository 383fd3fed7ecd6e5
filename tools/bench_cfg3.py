"""Diagnostic: value+gradient throughput at the largest configuration (n = 1000, q = 50, d = 10) and
the roofline figures of its dominant kernel, k_stage2<256, 0> (the wide solver path).
Writes gpurun_out/r03_cfg3_roofline.json when run from the repository root with --json
(tools/publish_profiles_r03.sh adds the counter bytes and copies it to profiles/)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hart6, gp as ogp                                   # noqa: E402
from bayesian_optimization_amd import DeviceGP, sdp_batched           # noqa: E402
import bench                                                          # noqa: E402

q, d, n = 50, 10, 1000
model = hart6.make_model(n, d=d, lengthscale=0.8)
gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, 1.0, [0.8], 1e-6, floor_mode=1)
gp.set_profiling(True)
if '--z-free' in sys.argv:          # measurement: second GEMM on the A rows, no Z buffer
    gp.set_z_free(True)
out = {}
# --only-2048: the counter passes, where every launch of a kernel should be the same size
for B in ((2048,) if '--only-2048' in sys.argv else (64, 512, 2048)):
    X = torch.as_tensor(np.random.RandomState(2).uniform(-.5, .5, (B, q, d)), device='cuda')
    for _ in range(2):
        gp.value_grad(X)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    stage = np.zeros(5)
    kms = 0.0
    for _ in range(reps):
        v, g, st, fl = gp.value_grad(X)
        stage += np.array(gp.stage_times_ms())
        kms += gp.solver_kernel_ms()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    stage /= reps
    kms /= reps
    print('B=%5d  %.2f ms  %.0f evals/s  unsolved %d  stages %s  solver kernel %.2f ms'
          % (B, dt * 1e3, B / dt, int((st != 0).sum()), np.round(stage, 2), kms))
    out[B] = dict(ms_per_call=dt * 1e3, evals_per_s=B / dt, stage_ms=[float(x) for x in stage], solver_kernel_ms=kms)
omg = gp.omega(X)
_, _, _, st2, iters = sdp_batched(omg, gp.fmin, want_Y=False)
iters = iters.cpu().numpy()
newton, sweeps = (iters & 0xffff).astype(float), (iters >> 16).astype(float)
print('newton mean %.2f max %d  sweeps mean %.2f max %d' % (newton.mean(), newton.max(), sweeps.mean(), sweeps.max()))
npad = n + (-n) % 64
fl = sum(bench.flops_model(q, n, npad, a, b) for a, b in zip(newton, sweeps))
kms = out[2048]['solver_kernel_ms']
ach = fl / (kms * 1e-3) / 1e12
roof = {'workload': 'cfg3: smooth 10-d GP n=1000, q=50, d=10, 2048 candidate batches (seed 2), value+gradient',
        'kernel': 'k_stage2<256, 0> (posterior moments + SDP solve, wide path: four waves per candidate)',
        'bound': 'fp64 (vector unit and LDS)', 'flops_per_candidate': fl / 2048, 'flops_per_launch': fl,
        'launch_ms': kms, 'achieved': ach, 'peak': bench.FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
        'frac': ach / bench.FP64_PEAK_TFLOPS, 'newton_iters_mean': float(newton.mean()),
        'jacobi_sweeps_mean': float(sweeps.mean()), 'evals_per_s': out[2048]['evals_per_s'],
        'ms_per_call': out[2048]['ms_per_call'], 'stage_ms': dict(zip(['gram', 'gemm_cross', 'moments_sdp', 'gemm_back', 'gradient'], out[2048]['stage_ms'])),
        'algorithmic_hbm_bytes_per_candidate': 8 * (2 * q * d + 1), 'by_batch': out,
        'note': 'flops: bench.flops_model (useful work, from the solver\'s own Newton and sweep counters); time: HIP events on '
                'the dispatch of k_stage2; two workgroups per compute unit (LDS 67 KB each), so 2048 candidates are four '
                'rounds of resident workgroups'}
print(json.dumps(roof))
if '--json' in sys.argv:
    with open(os.path.join(ROOT, 'gpurun_out', 'r03_cfg3_roofline.json'), 'w') as fh:
        json.dump(roof, fh, indent=1)
