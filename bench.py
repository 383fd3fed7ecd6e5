#!/usr/bin/env python
"""Benchmark of the OEI value+gradient path: candidate evaluations per second.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic candidate batches already resident
in HBM: value + gradient of B candidates (q=20, d=6) under the n=310 hart6 GP, the local arg-min
and, for N > 1, the all-gather of the (value, index, X) record that picks the best start.  Work per
GPU is fixed (weak scaling).  Rank 0 prints ONE JSON line.  Order inside a rank: model, Hessian
throughput leg and the profiled calls for the roofline (untimed; GPU work that also leaves clocks,
caches and allocator as a running loop has them), W warm-up steps, barrier, K timed steps, barrier;
then the callers' legs and the CPU baseline.

With --gpus N > 1 and no WORLD_SIZE in the environment this process only launches: it starts N
ranks of itself (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, rendezvous on 127.0.0.1),
relays rank 0's JSON line and exits with the worst exit code.  It does so before anything touches
the GPU.  Under an external launcher (torch.distributed.run) WORLD_SIZE is set and the process is a
rank.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# cfg 2 (one GPU): 4096 candidate batches, seed 1.  cfg 4 (eight GPUs): 65 536 = 8192 per GPU, seed 3;
# every multi-GPU run uses that per-GPU size so that work per GPU is the same for N = 2, 4, 8.
BATCH_SINGLE, SEED_SINGLE = 4096, 1
BATCH_MULTI, SEED_MULTI = 8192, 3

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


def hessian_flops_model(q, d, n, npad, sdp_flops, executed=False):
    """Algorithmic flops of one acquisition Hessian (fma = 2 flops), q <= 32 path:
      front     q(d+1)+1 data rows (values, d derivative components, y) through the triangular GEMM
                with L^-1 (K range of a 64-column tile stops at its diagonal), then the Gram matrix
                of the block (lower triangle, n-length sums)
      solve     the SDP solve of the value+gradient path (`sdp_flops`) plus W = Z L^-1
      dirs      the table U^T v' of every direction (q d rows of q^2 multiply-adds), then q d
                directions: three q^3 products (the direction stops at Y = U Z), about 12 q^2 for
                the vector passes and the two triangular solves, and the contraction of the
                direction's row with every dOmega_(b,m): q d (2 q + 30)
    Useful work only: the rows that carry data.  `executed` counts what the kernels multiply
    instead -- blocks padded to a multiple of 16 rows, whole 64 x 64 tiles (or 16-row strips) of the block Gram matrix."""
    rows = q * (d + 1) + 1
    nt = npad // 64
    tri = 64 * 64 * nt * (nt + 1)                      # per row: column tile j takes 64 * 64 (j + 1) multiply-adds
    if executed:
        RP = -(-rows // 16) * 16                       # blocks are packed to a multiple of 16 rows
        nfull, rem = RP // 64, RP % 64
        rem_exec = 0 if rem == 0 else (16 if rem <= 16 else 64)     # a last tile of 16 rows runs as a strip
        front = RP * tri + (nfull * (nfull + 1) // 2 * 64 * 64 + rem_exec * 64 * (nfull + 1)) * 2 * npad
    else:
        front = rows * tri + rows * (rows + 1) * npad  # Gram: rows (rows + 1) / 2 sums of npad multiply-adds
    solve = sdp_flops + q * tri
    qd = q * d
    dirs = qd * 2 * q * q + qd * (6 * q ** 3 + 12 * q * q) + qd * qd * (2 * q + 30)
    return front + solve + dirs


def candidate_block(total, q, d, seed, lo, hi, chunk=4096):
    """Rows [lo, hi) of the global candidate array RandomState(seed).uniform(-.5, .5, (total, q, d)),
    drawn chunk by chunk so that a rank never holds more than its block plus one chunk."""
    rs = np.random.RandomState(seed)
    out = np.empty((hi - lo, q, d))
    for c0 in range(0, min(total, hi), chunk):
        c1 = min(total, c0 + chunk)
        blk = rs.uniform(-0.5, 0.5, (c1 - c0, q, d))
        a, b = max(lo, c0), min(hi, c1)
        if a < b:
            out[a - lo:b - lo] = blk[a - c0:b - c0]
    return out


# ---- launcher ------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n_ranks, argv):
    """Start `n_ranks` copies of this script, one per GPU, and relay rank 0's output.  The parent
    never initialises the GPU (it imports neither torch nor the package)."""
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks),
                   LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = procs[0].stdout.read().decode()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [c for c in codes if c != 0]
    return bad[0] if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200,
                    help='timed steps (a step is about 1.1 ms: the first few dozen run while the GPU clocks still ramp up)')
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--batch', type=int, default=0,
                    help='candidate batches per GPU (default: %d on one GPU, %d per GPU on several)'
                         % (BATCH_SINGLE, BATCH_MULTI))
    ap.add_argument('--seed', type=int, default=-1, help='candidate seed (default: 1 on one GPU, 3 on several)')
    ap.add_argument('--q', type=int, default=20)
    ap.add_argument('--n', type=int, default=310)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=24.0, help='CPU-baseline budget, all legs together')
    ap.add_argument('--no-hessian', action='store_true', help='skip the Hessian throughput leg (profile runs: keeps the kernel statistics to the timed path)')
    ap.add_argument('--single-stream', action='store_true',
                    help='keep every launch of a step on one stream (kernel-trace profiles of whole-batch launches); '
                         'by default batches of two solver rounds and more run as two halves on two internal streams')
    ap.add_argument('--z-free', action='store_true',
                    help='measurement: second GEMM on the A rows with M11 applied in its epilogue (no Z buffer); off by default')
    ap.add_argument('--backend', default='nccl', help="process-group backend; 'gloo' only for rehearsing N > 1 on one GPU or on the CPU")
    ap.add_argument('--one-device', action='store_true', help='rehearsal: every rank uses GPU 0')
    ap.add_argument('--force-collective', action='store_true',
                    help='with --gpus 1: create a process group of one rank and send the best-start record through '
                         'the all-gather anyway (loads RCCL and runs the collective on a single GPU)')
    ap.add_argument('--rehearse-cpu', action='store_true',
                    help='rehearsal of the launcher, the sharding and the best-start exchange without a GPU: '
                         'the evaluation is replaced by a closed-form stand-in; the printed rate means nothing')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    cpu_only = args.rehearse_cpu
    if cpu_only:
        args.backend = 'gloo'
        dev = torch.device('cpu')
    else:
        if args.one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)            # before the package import: its context goes here
        dev = torch.device('cuda', local_rank)
    grouped = world > 1 or args.force_collective
    if grouped:
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(_free_port()))
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)
        world = dist.get_world_size()                # what the process group saw, not the flag
    red_dev = dev if args.backend == 'nccl' else torch.device('cpu')

    from bayesian_optimization_amd import multistart
    from oracle import hart6, gp as ogp

    q, n, d = args.q, args.n, 6
    B = args.batch or (BATCH_SINGLE if world == 1 else BATCH_MULTI)
    seed = args.seed if args.seed >= 0 else (SEED_SINGLE if world == 1 else SEED_MULTI)
    total = B * world
    lo, hi = multistart.shard_bounds(total, world, rank)
    X_local = torch.as_tensor(candidate_block(total, q, d, seed, lo, hi), dtype=torch.float64,
                              device=dev).contiguous()

    if cpu_only:
        gp = None

        def step():
            value = torch.sin(3.0 * X_local).sum(dim=(1, 2)) + X_local[:, 0, 0] ** 2
            li = torch.argmin(value).reshape(1)
            rec = multistart.make_record(value[li], li + lo, X_local.reshape(hi - lo, -1)[li])
            bv, bi, bx = multistart.select_best(rec, force=args.force_collective)
            return value, None, torch.zeros(hi - lo, dtype=torch.int32), (bv, bi, bx.reshape(q, d))

        def sync():
            pass
    else:
        from bayesian_optimization_amd import DeviceGP, sdp_batched
        model = hart6.make_model(n)                     # CPU copy: data + hyper-parameters only
        gp = DeviceGP(model.X, model.y, ogp.KERNEL_MATERN32, model.kernel.variance,
                      model.kernel.lengthscales, model.noise, floor_mode=model.kernel.floor_mode)
        if args.single_stream:
            gp.set_pipelining(False)
        if args.z_free:
            gp.set_z_free(True)

        def step():
            return multistart.evaluate_and_select(gp, X_local, index_base=lo, force_collective=args.force_collective)

        sync = torch.cuda.synchronize

    # Hessian throughput of the same model and candidates (second half of the metric), outside the
    # timed region and ahead of it: a few calls on a slice of this rank's candidates.  Every rank
    # runs it, so that all of them enter the timed steps with the clocks the preceding work left
    # (a run with a handful of warm-up steps otherwise times the clock ramp after the idle set-up).
    hess_timing = None
    if not cpu_only and not args.no_hessian:
        hb = min(B, 2048)
        hreps = 20
        for _ in range(15):                     # buffers, then clocks (this leg runs first, after the idle set-up)
            gp.hessian(X_local[:hb])
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(hreps):
            _, hst = gp.hessian(X_local[:hb])
        torch.cuda.synchronize()
        hess_timing = (hb, hreps, (time.perf_counter() - th) / hreps, int((hst != 0).sum().item()))

    # ---- roofline of the dominant kernel, measured live with HIP events on its stream: profiled
    # calls of the timed workload on one stream, ahead of the timed steps (they also leave caches,
    # allocator and clocks in the state of a running loop) ----
    prof_timing = None
    if not cpu_only:
        gp.set_profiling(True)
        for _ in range(30):
            gp.value_grad(X_local, want_flags=False)
        stage = np.zeros(5)
        kernel_ms = 0.0
        reps = 10
        for i in range(2 * reps):
            gp.value_grad(X_local, want_flags=False)
            if i >= reps:
                stage += np.array(gp.stage_times_ms())
                kernel_ms += gp.solver_kernel_ms()
        stage /= reps
        kernel_ms /= reps
        gp.set_profiling(False)
        prof_timing = (stage, kernel_ms, reps)

    for _ in range(args.warmup):
        step()
    sync()
    if grouped:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    if grouped:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    value, grad, status, best = out
    n_bad = int((status != 0).sum().item())
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        nb = torch.tensor([n_bad], dtype=torch.int64, device=red_dev)
        dist.all_reduce(nb)
        n_bad = int(nb.item())

    result = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        evals_per_s = total * args.steps / dt
        roofline = hessian = cpu = callers = None
        if not cpu_only:
            stage, kernel_ms, reps = prof_timing
            names = ['gram', 'gemm_cross', 'moments_sdp', 'gemm_back', 'gradient']
            omg = gp.omega(X_local)
            _, _, _, st2, iters = sdp_batched(omg, gp.fmin, want_Y=False)
            iters = iters.cpu().numpy()
            newton = (iters & 0xffff).astype(np.float64)
            sweeps = (iters >> 16).astype(np.float64)
            fl = sum(flops_model(q, n, gp.n + (-gp.n) % 64, a, b) for a, b in zip(newton, sweeps))
            dom = int(np.argmax(stage))
            # the kernel's own duration, from events attached to its dispatch (what a kernel trace
            # reports); stage[2], the interval between two events recorded around it on the stream,
            # also holds the launch gap those events open
            sdp_ms = kernel_ms
            achieved = fl / (sdp_ms * 1e-3) / 1e12
            traffic = step_bytes = None
            prof_json = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
            if os.path.exists(prof_json):
                try:
                    pj = json.load(open(prof_json))
                    traffic = pj.get('moments_sdp_hbm_bytes_per_launch')
                    if B == BATCH_SINGLE and q == 20:
                        step_bytes = pj.get('step_hbm_bytes')
                except Exception:
                    traffic = None
            roofline = {'bound': 'fp64 VALU issue', 'kernel': 'k_stage2 (posterior moments + SDP solve)',
                        'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': achieved / FP64_PEAK_TFLOPS, 'traffic': traffic,
                        'peak_note': 'fp64 vector peak = fp64 matrix peak on MI355X; two thirds of the '
                                     "kernel's flops and nearly all of its time are vector arithmetic",
                        'sustained_note': 'memory-free loops on this chip sustain 59 TFLOP/s of v_fma_f64 (two waves per SIMD) and 77 '
                                          'TFLOP/s of v_mfma_f64_16x16x4 at any occupancy (tools/micro/mixed_pipes.hip, mfma_occupancy.hip)',
                        'launch_ms': sdp_ms, 'flops_per_launch': fl,
                        'launch_ms_note': 'HIP events attached to the dispatch of k_stage2 on its stream (hipExtLaunchKernel), '
                                          'mean of %d whole-batch launches on one stream; stage_ms are intervals between '
                                          'events recorded between the kernels and include the launch gaps these open' % reps,
                        'hbm_step': None if step_bytes is None else {
                            'bytes': step_bytes, 'GB/s': step_bytes / (ms_per_step * 1e-3) / 1e9, 'peak': 8000.0,
                            'note': 'PMC bytes of the five kernels of a step (profiles/pmc_traffic.json) over the measured step time; no kernel of the step is HBM-bound'},
                        'stage_ms': dict(zip(names, [float(x) for x in stage])),
                        'dominant_stage': names[dom],
                        'newton_iters_mean': float(newton.mean()), 'jacobi_sweeps_mean': float(sweeps.mean())}
            if hess_timing is not None:
                hb, hreps, th, h_unsolved = hess_timing
                hfl = hessian_flops_model(q, d, n, gp.n + (-gp.n) % 64, fl / B)
                hfl_exec = hessian_flops_model(q, d, n, gp.n + (-gp.n) % 64, fl / B, executed=True)
                hessian = {'value': hb / th, 'unit': 'Hessians/s (one GPU)', 'candidates_per_call': hb,
                           'ms_per_call': th * 1e3, 'unsolved': h_unsolved,
                           'roofline': {'bound': 'fp64 (MFMA front end, vector-unit directions)',
                                        'flops_per_hessian': hfl, 'flops_per_hessian_executed': hfl_exec,
                                        'achieved': hfl * hb / th / 1e12,
                                        'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                        'frac': hfl * hb / th / 1e12 / FP64_PEAK_TFLOPS,
                                        'note': 'whole call (8 launches), wall clock over %d calls; flops of the rows that '
                                                'carry data (the executed count includes the padding of a block to 16 rows and whole tiles of its Gram matrix)' % hreps}}
            if not args.no_hessian and world == 1:
                # the callers of the path, outside the timed region: 2048 multistart descents
                # through the device optimiser, and a hyper-parameter fit with 20 restarts
                from bayesian_optimization_amd import hyperfit, lockstep
                nstart = min(B, 2048)
                Xs = X_local[:nstart]
                lockstep.optimize_multistart(gp, Xs[:64], [-0.5] * d, [0.5] * d, max_iter=3)
                torch.cuda.synchronize()
                tm = time.perf_counter()
                res, _ = lockstep.optimize_multistart(gp, Xs, [-0.5] * d, [0.5] * d)
                torch.cuda.synchronize()
                tm = time.perf_counter() - tm
                import collections
                callers = {'multistart': {'starts': nstart, 'seconds': tm, 'evaluations': int(res.nfev.sum().item()),
                                          'converged': int(res.converged.sum().item()),
                                          'rounds': int(res.nfev.max().item()),
                                          'stop_reasons': dict(collections.Counter(res.message)),
                                          'method': 'bounded L-BFGS, state on the device, rounds driven inside the library '
                                                    '(oei_lbfgs_minimize_rounds), iteration cap 2000'}}
                hyperfit.fit(gp.X_train, gp.y_train, gp.kernel_kind, 1.0, 1.0, gp.noise, floor_mode=gp.floor_mode,
                             restarts=2, rng=np.random.RandomState(0))
                tf = time.perf_counter()
                fit = hyperfit.fit(gp.X_train, gp.y_train, gp.kernel_kind, 1.0, 1.0, gp.noise, floor_mode=gp.floor_mode,
                                   restarts=20, rng=np.random.RandomState(0))
                tf = time.perf_counter() - tf
                callers['hyperparameter_fit'] = {'restarts': 20, 'seconds': tf, 'nlml': fit['nlml'],
                                                 'method': 'all restarts batched (oei_fit_nlml_batched)'}
            if not args.no_cpu_baseline and world == 1:
                # the CPU restatement of the reference path (oracle) timed on this box's host
                # cores, on a bounded sample of the benchmark's own candidates
                from oracle import cpu_bench
                cpu = cpu_bench.run_all(n, q, X_local[:256].cpu().numpy(), args.cpu_seconds)
        bx = best[2].detach().cpu().contiguous().numpy()
        result = {
            'metric': 'OEI value+grad evals/sec (q=%d, n=%d)' % (q, n),
            'value': evals_per_s, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%s: hart6 GP n=%d, q=%d, d=%d, %d multistart candidate batches '
                                   'per GPU (seed %d), value+gradient + best-start selection'
                                   % ('cfg2' if world == 1 else 'cfg4 recipe', n, q, d, B, seed),
                       'candidates_total': total, 'candidates_per_gpu': B,
                       'streams': 'one' if (args.single_stream or cpu_only or B < 4096) else
                                  'two halves of the batch on two internal streams (the per-stage times of '
                                  '`roofline` are whole-batch launches on one stream)',
                       'parallelism': 'candidates sharded over %d rank(s) (%s), one GPU each, '
                                      'one all-gather of a %d-byte record'
                                      % (world, ('world size seen by ' + args.backend) if grouped else 'no process group',
                                         8 * (2 + q * d))},
            'unconverged': n_bad,
            'collective': ('all_gather_into_tensor of the record over %s, %d rank(s)' % (args.backend, world)) if grouped else None,
            'best': {'value': float(best[0].item()), 'index': int(best[1].item()),
                     'x_sha1': hashlib.sha1(bx.tobytes()).hexdigest()},
            'hessian': hessian, 'callers': callers, 'roofline': roofline, 'cpu_baseline': cpu,
        }
        if cpu_only:
            result['data'] = 'synthetic; CPU rehearsal of the launcher (stand-in evaluation, rate meaningless)'
        print(json.dumps(result), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == '__main__':
    main()
