"""Per-kernel HBM counters from two rocprofv3 --pmc passes written as CSV (FETCH_SIZE and
WRITE_SIZE, one pass each):
    python tools/pmc_summary_csv.py fetch_counter_collection.csv write_counter_collection.csv [out.json]
Values are KB per launch as the counters report them; `hbm_bytes_per_launch` applies the gfx950
correction (FETCH_SIZE counts 128-byte requests as 64 bytes: doubled) and KB = 1024 bytes."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


def short(name):
    return name.replace('void ', '').split('(')[0]


fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, (0, 0.0))
    w = write.get(k, (0, 0.0))
    out[short(k)] = {
        'launches_FETCH_SIZE': f[0], 'FETCH_SIZE_KB_per_launch': f[1],
        'launches_WRITE_SIZE': w[0], 'WRITE_SIZE_KB_per_launch': w[1],
        'hbm_bytes_per_launch': (2.0 * f[1] + w[1]) * 1024.0,
    }
text = json.dumps(out, indent=1, sort_keys=True)
print(text)
if len(sys.argv) > 3:
    open(sys.argv[3], 'w').write(text + '\n')
