#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) as a per-kernel table.

    python tools/prof_summary.py gpurun_out/prof4/r4_results.db 8 > profiles/r01_kernel_stats.txt

(second argument = number of training steps in the profiled run, to print per-step figures)."""
import sqlite3
import sys

db, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                        "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary: %s" % db)
print("# %d kernel launches, %.1f ms total GPU kernel time; per step (%g steps): %.1f launches, %.3f ms"
      % (sum(r[1] for r in rows), tot / 1e3, steps, sum(r[1] for r in rows) / steps, tot / 1e3 / steps))
print("%-78s %8s %11s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    print("%-78s %8d %11.1f %10.2f %10.2f %10.2f %6.2f" % (r[0][:78], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
