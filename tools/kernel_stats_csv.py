"""rocprofv3 `*_kernel_stats.csv` -> the short form kept under profiles/ (kernel name without its
argument list, microseconds).  Usage: python tools/kernel_stats_csv.py in.csv out.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], 'w') as fh:
    fh.write('kernel,calls,total_us,avg_us,min_us,max_us,percent\n')
    for r in rows:
        name = re.sub(r'^void ', '', r['Name'])
        name = re.sub(r'\(.*$', '', name)
        fh.write('"%s",%s,%.1f,%.2f,%.2f,%.2f,%s\n' % (name[:100], r['Calls'], float(r['TotalDurationNs']) / 1e3,
                                                        float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3,
                                                        float(r['MaxNs']) / 1e3, r['Percentage']))
