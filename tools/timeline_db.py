"""Per-dispatch timeline of the LAST step in a rocprofv3 kernel trace (rocpd sqlite): python tools/timeline_db.py <results.db> [min us]."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.stream_id from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
idx = [i for i, r in enumerate(rows) if "fold_conditioning" in r[0]]
st = idx[-1] if idx else 0
t0 = rows[st][1]
for r in rows[st:]:
    if (r[2] - r[1]) > min_us * 1e3:
        print("%9.3f ms  +%8.1f us  grid %7d  stream %s  %s" % ((r[1] - t0) / 1e6, (r[2] - r[1]) / 1e3, r[3], r[4], r[0].replace("_ZN4sahs", "")[:64]))
print("step end %.3f ms" % ((rows[-1][2] - t0) / 1e6))
