"""Per-kernel HBM counters from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, one rocpd
database each): python tools/pmc_summary.py fetch.db write.db [out.json]
Values are KB per launch as the counters report them; `hbm_bytes_per_launch` applies the gfx950
correction (FETCH_SIZE counts 128-byte requests as 64 bytes: doubled) and KB = 1024 bytes."""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection "
                      "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {n: (c, v) for n, c, v in rows}


def short(name):
    name = name.replace('void ', '')
    return name.split('(')[0]


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
