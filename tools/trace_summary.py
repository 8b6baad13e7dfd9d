#!/usr/bin/env python3
"""tools/trace_summary.py DB [N] -- per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output):
calls, average / minimum duration, total; with N, also the timeline of the last N dispatches (start offset, duration,
gap to the kernel before).  Diagnostic; the summaries under profiles/ are made with it."""
import glob
import sqlite3
import sys

db = sys.argv[1]
if not db.endswith(".db"):
    db = sorted(glob.glob(db + "/**/*.db", recursive=True))[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
print("%-70s %6s %12s %12s %12s" % ("kernel", "calls", "avg us", "min us", "total us"))
q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 "
     "from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 5 desc" % (kd, ks))
for name, n, avg, mn, tot in c.execute(q).fetchall():
    print("%-70s %6d %12.2f %12.2f %12.1f" % (name[:70], n, avg, mn, tot))
if len(sys.argv) > 2:
    n = int(sys.argv[2])
    rows = c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start desc limit %d"
                     % (kd, ks, n)).fetchall()[::-1]
    t0, prev = rows[0][1], None
    print()
    for name, st, en in rows:
        print("%10.2f us  +%8.2f us  gap %7.2f us  %s" % ((st - t0) / 1e3, (en - st) / 1e3,
                                                          (st - prev) / 1e3 if prev else 0.0, name[:60]))
        prev = en
