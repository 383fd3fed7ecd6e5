#!/bin/bash
# Collects what profiles/r03_* is made from.  Run from the repository root on the GPU box; writes
# under gpurun_out/r03/.  Each counter pass is a run of its own, the program directly after `--`.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03
rm -rf $O && mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 python3 bench.py --single-stream --no-cpu-baseline --no-hessian > $O/bench_single.json 2> $O/bench_single.err || exit 1
timeout -k 10 300 python3 bench.py --force-collective --no-cpu-baseline --no-hessian > $O/bench_collective.json 2> $O/bench_collective.err || exit 1
# kernel statistics and counters of whole-batch launches: --single-stream (by default a step runs as two
# halves on two streams, whose kernels overlap and are half the size)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-hessian --single-stream > $O/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats2 -o stats2 --output-format csv -- python3 bench.py --no-cpu-baseline --no-hessian > $O/stats2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/statsh -o statsh --output-format csv -- python3 bench.py --no-cpu-baseline --single-stream > $O/statsh.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- python3 bench.py --no-cpu-baseline --no-hessian --single-stream --steps 5 --warmup 1 > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- python3 bench.py --no-cpu-baseline --no-hessian --single-stream --steps 5 --warmup 1 > $O/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/cfg3 -o cfg3 --output-format csv -- python3 tools/bench_cfg3.py --json > $O/cfg3.log 2>&1 || exit 1
cp gpurun_out/r03_cfg3_roofline.json $O/cfg3_roofline.json || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/cfg3_fetch -o fetch --output-format csv -- python3 tools/bench_cfg3.py --only-2048 > $O/cfg3_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/cfg3_write -o write --output-format csv -- python3 tools/bench_cfg3.py --only-2048 > $O/cfg3_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/hess3 -o hess3 --output-format csv -- python3 tools/bench_hessian_cfg3.py > $O/hess3.log 2>&1 || exit 1
timeout -k 10 100 python3 tools/bench_hessian_small.py > $O/hess_small.log 2>&1 || exit 1
timeout -k 10 200 python3 tools/bench_model.py > $O/model.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/bench_lockstep.py > $O/lockstep.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/bench_fit.py > $O/fit.log 2>&1 || exit 1
# one 20-start multistart with the round loop inside the library, under a kernel trace
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/round_lbfgs -o round_lbfgs --output-format csv -- python3 tools/round_profile.py lbfgs > $O/round_lbfgs.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/round_newton -o round_newton --output-format csv -- python3 tools/round_profile.py newton > $O/round_newton.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/profile_bo_loop.py --hessian=0 > $O/bo_loop_lbfgs.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/profile_bo_loop.py --nl_solver=lockstep-newton > $O/bo_loop_newton.log 2>&1 || exit 1
timeout -k 10 400 bash tools/sdp_phase_cycles.sh 20 4096 > $O/phase_cycles.log 2>&1 || exit 1
timeout -k 10 400 bash tools/sdp_phase_cycles.sh 32 2048 > $O/phase_cycles_q32.log 2>&1 || exit 1
timeout -k 10 400 bash tools/sdp_phase_cycles.sh 50 256 > $O/phase_cycles_q50.log 2>&1 || exit 1
timeout -k 10 400 bash tools/hess_wide_phase_cycles.sh 256 > $O/hess_wide_phase_cycles.log 2>&1 || exit 1
# do fp64 matrix and vector instructions of two waves on one SIMD overlap?
hipcc -O3 --offload-arch=gfx950 -o /tmp/mixed_pipes tools/micro/mixed_pipes.hip > $O/micro_build.log 2>&1 || exit 1
timeout -k 10 60 /tmp/mixed_pipes > $O/micro_mixed_pipes.log 2>&1 || exit 1
# cycles of one q x q x q product of LDS matrices by four waves (the wide Hessian kernel's products)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -mllvm -pragma-unroll-threshold=1000000 -I bayesian_optimization_amd/csrc -I include -o /tmp/mm_lds tools/micro/mm_lds.hip >> $O/micro_build.log 2>&1 || exit 1
(timeout -k 10 60 /tmp/mm_lds 50 && timeout -k 10 60 /tmp/mm_lds 64) > $O/micro_mm_lds.log 2>&1 || exit 1
# the variant without the Z buffer
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/zfree_fetch -o fetch --output-format csv -- python3 bench.py --no-cpu-baseline --no-hessian --single-stream --z-free --steps 5 --warmup 1 > $O/zfree_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/zfree_write -o write --output-format csv -- python3 bench.py --no-cpu-baseline --no-hessian --single-stream --z-free --steps 5 --warmup 1 > $O/zfree_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/zfree_stats -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-hessian --single-stream --z-free > $O/zfree_stats.log 2>&1 || exit 1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-hessian --z-free > $O/bench_zfree.json 2> $O/bench_zfree.err || exit 1
timeout -k 10 200 python3 tools/warm_probe_first_order.py > $O/warm_probe_first_order.log 2>&1 || exit 1
echo done > $O/done
