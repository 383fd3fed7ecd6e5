"""Per-kernel statistics (calls, total/average/min/max duration in us) from a rocprofv3 rocpd
database: python tools/rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
rows = db.execute(
    f"select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
    f"max(d.end - d.start), max(d.group_segment_size), max(d.private_segment_size) "
    f"from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
    f"group by s.{name_col} order by 3 desc").fetchall()
lines = ['kernel,calls,total_us,avg_us,min_us,max_us,lds_bytes,scratch_bytes']
for n, c, tot, avg, mn, mx, lds, scr in rows:
    lines.append('"%s",%d,%.1f,%.2f,%.2f,%.2f,%d,%d' % (n[:90], c, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, lds, scr))
text = '\n'.join(lines)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(text + '\n')
