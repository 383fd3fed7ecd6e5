#!/usr/bin/env python
"""Benchmark of the OEI value+gradient path: candidate evaluations per second.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic candidate batches already resident
in HBM: value + gradient of B candidates (q=20, d=6) under the n=310 hart6 GP, the local arg-min
and, for N > 1, the all-gather of the (value, index, X) record that picks the best start.  Work per
GPU is fixed (weak scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = fp64 matrix peak (datasheet)


def flops_model(q, n, npad, newton_iters, sweeps):
    """Algorithmic flops of the moments+SDP kernel for one candidate (fma = 2 flops), from the
    counters the solver returns (register path, q <= 32).  Useful work only: padding lanes and
    the repeated rotation-parameter arithmetic in idle lanes are not counted.
      moments   posterior mean and covariance: n-length dot products for q(q+1)/2 + q pairs
      zmat      Z = M11 A: q^2 multiply-adds for each of npad columns
      factor    Cholesky of the q x q covariance, once
      per_eval  m = Lc^T p, G, the warm-start product G V (2 q^3), back substitution (q^3),
                gradient (3 q^2)
      per_sweep one-sided Jacobi: q(q-1)/2 pairs x (dot product 2q + two column updates 6q +
                about 30 for the rotation) + refresh of the norms (2 q^2)
      per_dir   q(q+1)/2 + q + 1 quadratic forms of q^2 + 6q flops, kernel q^2, Cholesky q^3/3,
                two triangular solves 2 q^2
    """
    qe = q + (q & 1)
    per_sweep = (qe - 1) * (qe // 2) * (8 * q + 30) + 2 * q * q
    per_eval = 3 * q ** 3 + 6 * q * q
    nitems = q * (q + 1) // 2 + q + 1
    per_dir = nitems * (q * q + 6 * q) + q * q + q ** 3 / 3 + 2 * q * q
    moments = 2 * n * (q * (q + 1) / 2 + q)
    zmat = 2 * q * q * npad
    factor = q ** 3 / 3
    evals = newton_iters + 1
    return moments + zmat + factor + evals * per_eval + sweeps * per_sweep + newton_iters * per_dir


def cpu_baseline(model, X_sample, seconds_hint=20.0):
    """The CPU restatement of the reference algorithm (oracle, first-order solve at eps=1e-5,
    sdp.py:51) timed on the host, one candidate at a time, single thread."""
    from oracle import oei as ooei
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:
        ctx = None
    orc = ooei.OEIOracle(model)
    done, t0 = 0, time.time()
    for b in range(X_sample.shape[0]):
        orc.acquisition(X_sample[b], first_order=True)
        done += 1
        if time.time() - t0 > seconds_hint:
            break
    dt = time.time() - t0
    if ctx is not None:
        ctx.restore_original_limits() if hasattr(ctx, 'restore_original_limits') else None
    return done / dt, done, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=4096, help='candidate batches per GPU')
    ap.add_argument('--q', type=int, default=20)
    ap.add_argument('--n', type=int, default=310)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-hessian', action='store_true', help='skip the Hessian throughput leg (profile runs: keeps the kernel statistics to the timed path)')
    ap.add_argument('--backend', default='nccl', help="process-group backend; 'gloo' only for rehearsing N > 1 on one GPU")
    ap.add_argument('--one-device', action='store_true', help='rehearsal: every rank uses GPU 0')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from bayesian_optimization_amd import DeviceGP, sdp_batched, multistart
    from oracle import hart6, gp as ogp

    q, n, B = args.q, args.n, args.batch
    d = 6
    model = hart6.make_model(n)                         # CPU copy: data + hyper-parameters only
    gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, model.kernel.variance,
                  model.kernel.lengthscales, model.noise, floor_mode=model.kernel.floor_mode)
    total = B * world
    lo, hi = multistart.shard_bounds(total, world, rank)
    # candidates: seed 1 stream, each rank takes its contiguous block of the global batch
    rs = np.random.RandomState(1)
    X_all = rs.uniform(-0.5, 0.5, (total, q, d)) if total <= 65536 else None
    X_local = torch.as_tensor(X_all[lo:hi], dtype=torch.float64, device=dev).contiguous()
    del X_all

    def step():
        return multistart.evaluate_and_select(gp, X_local, index_base=lo)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value, grad, status, best = out
    n_bad = int((status != 0).sum().item())

    result = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        evals_per_s = total * args.steps / dt
        # ---- roofline of the dominant kernel, measured live with HIP events on its stream ----
        gp.set_profiling(True)
        stage = np.zeros(5)
        reps = 5
        for _ in range(reps):
            gp.value_grad(X_local, want_flags=False)
            stage += np.array(gp.stage_times_ms())
        stage /= reps
        gp.set_profiling(False)
        names = ['gram', 'gemm_cross', 'moments_sdp', 'gemm_back', 'gradient']
        omg = gp.omega(X_local)
        _, _, _, st2, iters = sdp_batched(omg, gp.fmin, want_Y=False)
        iters = iters.cpu().numpy()
        newton = (iters & 0xffff).astype(np.float64)
        sweeps = (iters >> 16).astype(np.float64)
        fl = sum(flops_model(q, n, gp.n + (-gp.n) % 64, a, b) for a, b in zip(newton, sweeps))
        dom = int(np.argmax(stage))
        sdp_ms = stage[2]
        achieved = fl / (sdp_ms * 1e-3) / 1e12
        traffic = None
        prof_json = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(prof_json):
            try:
                traffic = json.load(open(prof_json)).get('moments_sdp_hbm_bytes_per_launch')
            except Exception:
                traffic = None
        roofline = {'bound': 'mfma', 'kernel': 'k_stage2 (posterior moments + SDP solve)',
                    'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': achieved / FP64_PEAK_TFLOPS, 'traffic': traffic,
                    'launch_ms': sdp_ms, 'flops_per_launch': fl,
                    'stage_ms': dict(zip(names, [float(x) for x in stage])),
                    'dominant_stage': names[dom],
                    'newton_iters_mean': float(newton.mean()), 'jacobi_sweeps_mean': float(sweeps.mean())}
        # Hessian throughput of the same model and candidates (second half of the metric), outside
        # the timed region: a few calls on a slice of this rank's candidates
        hessian = None
        if not args.no_hessian:
            hb = min(B, 2048)
            gp.hessian(X_local[:hb])
            torch.cuda.synchronize()
            th = time.perf_counter()
            for _ in range(3):
                _, hst = gp.hessian(X_local[:hb])
            torch.cuda.synchronize()
            th = (time.perf_counter() - th) / 3
            hessian = {'value': hb / th, 'unit': 'Hessians/s (one GPU)', 'candidates_per_call': hb,
                       'ms_per_call': th * 1e3, 'unsolved': int((hst != 0).sum().item())}
        cpu = None
        if not args.no_cpu_baseline:
            v, done, secs = cpu_baseline(model, X_local[:24].cpu().numpy())
            cpu = {'value': v, 'unit': 'evals/s', 'cores': 1, 'kind': 'port',
                   'sample': '%d candidate batches (q=%d, n=%d) of the same workload, one at a time, '
                             'numpy restatement of the reference path with the first-order SDP solve '
                             'at eps=1e-5; %.1f s' % (done, q, n, secs)}
        result = {
            'metric': 'OEI value+grad evals/sec (q=%d, n=%d)' % (q, n),
            'value': evals_per_s, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'cfg2: hart6 GP n=%d, q=%d, d=%d, %d multistart candidate batches '
                                   'per GPU, value+gradient + best-start selection' % (n, q, d, B),
                       'candidates_total': total, 'candidates_per_gpu': B,
                       'parallelism': 'candidates sharded over %d GPU(s), one all-gather of a %d-byte record'
                                      % (world, 8 * (2 + q * d))},
            'unconverged': n_bad,
            'best': {'value': float(best[0].item()), 'index': int(best[1].item())},
            'hessian': hessian, 'roofline': roofline, 'cpu_baseline': cpu,
        }
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == '__main__':
    main()
