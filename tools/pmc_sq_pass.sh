#!/bin/bash
# One rocprofv3 counter pass (at most 8 SQ counters) over tools/prof_stage2.py; prints, per kernel,
# the counters summed over the launches.  Usage: bash tools/pmc_sq_pass.sh "<counters>" [name filter]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmc_sq_$$
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc $1 -d $O -o p --output-format csv -- python3 tools/prof_stage2.py > $O/log 2>&1 || { tail -5 $O/log; exit 1; }
python3 - "$O" "${2:-}" <<'PY'
import csv, collections, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k in sorted(acc):
    if sys.argv[2] in k:
        print(k, ' '.join('%s=%.4g' % kv for kv in sorted(acc[k].items())))
PY
rm -rf $O
