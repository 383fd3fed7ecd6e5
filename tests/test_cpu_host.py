"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, argument
validation works without a GPU, the sharding / best-start selection logic runs under gloo with
world_size 2.  No compute call is made on a GPU here."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, 'include', 'oei_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(oei_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from bayesian_optimization_amd import _lib
    names = _declared_functions()
    assert len(names) >= 14
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(raw, name), 'missing export: ' + name
    # and the binding covers the same set
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib.oei_abi_version() == _lib.ABI_VERSION


def test_argument_validation_without_gpu():
    from bayesian_optimization_amd._lib import lib
    null = ctypes.c_void_p(0)
    # empty batches succeed without touching a device
    assert lib.oei_sdp_batched(0, 5, null, null, null, null, null, null, null, null) == 0
    # out-of-range sizes and null pointers are rejected with OEI_ERR_ARG and a message
    assert lib.oei_sdp_batched(4, 1, null, null, null, null, null, null, null, null) == -1
    assert lib.oei_sdp_batched(4, 200, null, null, null, null, null, null, null, null) == -1
    assert lib.oei_sdp_batched(4, 5, null, null, null, null, null, null, null, null) == -1
    assert b'null' in lib.oei_last_error()
    assert lib.oei_value_grad_batched(null, 1, 3, null, null, null, null, null, null) == -1
    assert lib.oei_hessian_batched(null, 1, 3, null, null, null, null) == -1
    assert lib.oei_solution_derivative_batched(2, 5, 0, null, null, null, null, null, null, null, null) == -1
    handle = ctypes.c_void_p(0)
    X = np.zeros((4, 2))
    y = np.zeros(4)
    ls = np.ones(1)
    rc = lib.oei_model_create(ctypes.byref(handle), 4, 2, X.ctypes.data_as(ctypes.c_void_p),
                              y.ctypes.data_as(ctypes.c_void_p), 7, 0, 1.0,
                              ls.ctypes.data_as(ctypes.c_void_p), 1, 1e-6)
    assert rc == -1 and b'kernel' in lib.oei_last_error()
    if not torch.cuda.is_available():
        rc = lib.oei_model_create(ctypes.byref(handle), 4, 2, X.ctypes.data_as(ctypes.c_void_p),
                                  y.ctypes.data_as(ctypes.c_void_p), 1, 1, 1.0,
                                  ls.ctypes.data_as(ctypes.c_void_p), 1, 1e-6)
        assert rc == -4          # OEI_ERR_NO_DEVICE: the product path fails loudly without a GPU


def test_product_path_has_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from bayesian_optimization_amd import DeviceGP, sdp_batched
    from bayesian_optimization_amd._lib import OeiError
    with pytest.raises(OeiError):
        DeviceGP(np.zeros((4, 2)), np.zeros(4), 1, 1.0, [1.0], 1e-6)
    with pytest.raises(OeiError):
        sdp_batched(np.eye(3)[None], np.zeros(1))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'bayesian_optimization_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.cuh', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f


def test_shard_bounds_cover_everything():
    from bayesian_optimization_amd.multistart import shard_bounds
    for total, world in [(65536, 8), (4096, 1), (10, 4), (7, 8)]:
        spans = [shard_bounds(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_shim_helpers_pack_unpack_and_C():
    from bayesian_optimization_amd.methods import sdp as msdp
    from oracle import sdp_ref
    Z = np.random.RandomState(1).randn(5, 5)
    Z = Z + Z.T
    assert np.array_equal(msdp.pack(Z, 5), sdp_ref.pack(Z))
    assert np.allclose(msdp.unpack(msdp.pack(Z, 5), 5), Z)
    C1, C2 = msdp.create_C(4, np.array([0.3])), sdp_ref.create_C(4, 0.3)
    assert all(np.array_equal(a, b) for a, b in zip(C1, C2))
    msdp.reset_warm_starting()
    assert len(msdp.past_omegas) == 0


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _select_worker(rank, world, port, total, q, d, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from bayesian_optimization_amd import multistart
    rs = np.random.RandomState(7)
    values = torch.as_tensor(rs.randn(total))
    values[5] = values[total - 3] = values.min() - 1.0       # a tie across ranks
    X = torch.as_tensor(rs.rand(total, q, d))
    lo, hi = multistart.shard_bounds(total, world, rank)
    local = values[lo:hi]
    li = int(torch.argmin(local))                            # first minimum = lowest index
    rec = multistart.make_record(local[li], torch.tensor(lo + li), X[lo + li])
    v, i, xb = multistart.select_best(rec)
    torch.save({'v': float(v), 'i': int(i), 'x': xb.clone()}, os.path.join(out_dir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_select_best_gloo_world2(tmp_path):
    """Sharded arg-min with one all-gather: every rank picks the same start, the one a single rank
    would pick, with ties going to the lowest global index."""
    total, q, d = 64, 3, 2
    port = _free_port()
    mp.spawn(_select_worker, args=(2, port, total, q, d, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'r0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'r1.pt'))
    rs = np.random.RandomState(7)
    values = rs.randn(total)
    values[5] = values[total - 3] = values.min() - 1.0
    X = rs.rand(total, q, d)
    assert r0['i'] == r1['i'] == 5
    assert r0['v'] == r1['v'] == values[5]
    assert np.array_equal(r0['x'].numpy(), X[5].ravel()) and torch.equal(r0['x'], r1['x'])


def test_select_best_single_process_handles_nan():
    from bayesian_optimization_amd import multistart
    rec = multistart.make_record(torch.tensor(float('nan')), torch.tensor(3), torch.zeros(4))
    v, i, x = multistart.select_best(rec)
    assert int(i) == 3


def _run_bench(*flags):
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(flags), env=env,
                         stdout=subprocess.PIPE, timeout=300, check=True).stdout.decode()
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out                      # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_launcher_starts_n_ranks():
    """`bench.py --gpus 2` must itself start two ranks (rehearsed on the CPU: gloo, stand-in
    evaluation), shard the same global candidate array and pick the start one rank would pick."""
    two = _run_bench('--gpus', '2', '--rehearse-cpu', '--batch', '96', '--seed', '3', '--steps', '2', '--warmup', '1')
    one = _run_bench('--gpus', '1', '--rehearse-cpu', '--batch', '192', '--seed', '3', '--steps', '2', '--warmup', '1')
    assert two['n_gpus'] == 2 and one['n_gpus'] == 1
    assert two['config']['candidates_total'] == one['config']['candidates_total'] == 192
    assert two['best']['index'] == one['best']['index']
    assert two['best']['x_sha1'] == one['best']['x_sha1']          # bit-identical best X
    assert two['best']['value'] == one['best']['value']
    assert '2 rank(s)' in two['config']['parallelism']


def test_bench_candidate_blocks_tile_the_global_array():
    import bench
    full = bench.candidate_block(300, 4, 3, 5, 0, 300, chunk=64)
    assert np.array_equal(full, np.random.RandomState(5).uniform(-.5, .5, (300, 4, 3)))
    for lo, hi in [(0, 100), (100, 200), (37, 38), (250, 300)]:
        assert np.array_equal(bench.candidate_block(300, 4, 3, 5, lo, hi, chunk=64), full[lo:hi])


def test_built_library_keeps_scalar_spill_carriers_in_registers():
    """A kernel that parks spilled scalar registers in vector-register lanes and then writes those
    carrier registers to scratch memory was observed to compute wrong results (padded Q = 32 solver
    instance, q = 21..31).  The shipped library must not contain that pattern
    (tools/check_spills.py disassembles every embedded gfx950 code object)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_spills
    from bayesian_optimization_amd import _lib
    bad, total = check_spills.scan(_lib.LIB_PATH)
    assert total > 40
    assert not bad, bad


def test_optimiser_workspace_layouts_and_argument_errors():
    """oei_lbfgs_layout / oei_newton_layout are host-only: sizes, offsets in header order without
    overlap, and the shape limits (history, the LDS-resident factor of the Newton step)."""
    import ctypes
    from bayesian_optimization_amd import _lib
    from bayesian_optimization_amd.lockstep import LBFGS_D, LBFGS_I, NEWTON_D, NEWTON_I, NEWTON_MAX_N
    lib = _lib.lib
    B, n, m = 37, 120, 10
    nd, ni = ctypes.c_size_t(0), ctypes.c_size_t(0)
    od = (ctypes.c_size_t * len(LBFGS_D))()
    oi = (ctypes.c_size_t * len(LBFGS_I))()
    assert lib.oei_lbfgs_layout(B, n, m, ctypes.byref(nd), ctypes.byref(ni), od, oi) == 0
    sizes = dict(X=B * n, F=B, G=B * n, DIR=B * n, STEP=B, XT=B * n, S=B * m * n, Y=B * m * n, RHO=B * m,
                 GAMMA=B, XC=B * n, LO=n, HI=n)
    order = sorted(LBFGS_D, key=LBFGS_D.get)
    for a, b in zip(order, order[1:] + [None]):
        end = od[LBFGS_D[b]] if b else nd.value
        assert od[LBFGS_D[a]] + sizes[a] <= end and od[LBFGS_D[a]] % 32 == 0
    assert oi[LBFGS_I['COUNT']] + 1 <= oi[LBFGS_I['NEED']] and oi[LBFGS_I['NEED']] + B <= ni.value
    assert lib.oei_lbfgs_layout(B, n, 0, ctypes.byref(nd), ctypes.byref(ni), None, None) == -1
    assert lib.oei_lbfgs_layout(B, n, 65, ctypes.byref(nd), ctypes.byref(ni), None, None) == -1
    assert lib.oei_lbfgs_layout(0, n, m, ctypes.byref(nd), ctypes.byref(ni), None, None) == -1
    od = (ctypes.c_size_t * len(NEWTON_D))()
    oi = (ctypes.c_size_t * len(NEWTON_I))()
    assert lib.oei_newton_layout(B, NEWTON_MAX_N, ctypes.byref(nd), ctypes.byref(ni), od, oi) == 0
    assert od[NEWTON_D['XH']] - od[NEWTON_D['XC']] >= B * NEWTON_MAX_N and oi[NEWTON_I['COUNTS']] + 2 <= oi[NEWTON_I['NEED']]
    assert lib.oei_newton_layout(B, NEWTON_MAX_N + 1, ctypes.byref(nd), ctypes.byref(ni), None, None) == -1
    assert 'LDS' in lib.oei_last_error().decode()


def test_device_optimisers_refuse_cpu_tensors():
    import torch
    from bayesian_optimization_amd import lockstep
    f = lambda X: ((X * X).sum(dim=1), 2 * X)
    with pytest.raises(ValueError):
        lockstep.minimize_lockstep_device(f, torch.zeros(2, 3, dtype=torch.float64), -1.0, 1.0)
    with pytest.raises(ValueError):
        lockstep.minimize_lockstep_newton_device(f, lambda X: None, torch.zeros(2, 3, dtype=torch.float64), -1.0, 1.0)
