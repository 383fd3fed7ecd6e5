#!/bin/bash
# Collects what profiles/ is made from: bench line, rocprofv3 kernel statistics, and the two HBM
# counter passes (each in a run of its own).  Run from the repository root on the GPU box;
# writes under gpurun_out/.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/p6_stats $O/p6_fetch $O/p6_write
timeout -k 10 300 python3 bench.py > $O/p6_bench.json 2> $O/p6_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p6_stats -o stats -- python3 bench.py --no-cpu-baseline --no-hessian > $O/p6_stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/p6_fetch -o fetch -- python3 bench.py --no-cpu-baseline --no-hessian --steps 5 --warmup 1 > $O/p6_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/p6_write -o write -- python3 bench.py --no-cpu-baseline --no-hessian --steps 5 --warmup 1 > $O/p6_write.log 2>&1
echo done > $O/p6_done
