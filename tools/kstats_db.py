"""Per-kernel totals of a rocprofv3 kernel trace in rocpd (sqlite) form: python tools/kstats_db.py <results.db> [top N] [divide by]."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kd, ks)).fetchall()
tot = sum(r[2] for r in rows)
print("total %.3f ms (/%g = %.3f)" % (tot / 1e6, div, tot / 1e6 / div))
for r in rows[:top]:
    print("%-84s n=%5d total %9.3f ms avg %9.1f us %5.1f%%" % (r[0][:84], r[1], r[2] / 1e6, r[3] / 1e3, 100 * r[2] / tot))
