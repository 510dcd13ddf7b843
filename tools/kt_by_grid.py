#!/usr/bin/env python3
"""Kernel-trace durations grouped by (kernel, grid size): python tools/kt_by_grid.py <results.db | directory> [name filter ...]"""
import glob
import os
import sqlite3
import sys

src = sys.argv[1]
db = src if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*.db"), recursive=True)[0]
filt = sys.argv[2:] or [""]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, grid_x, workgroup_x, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels "
                   "group by name, grid_x order by name, grid_x")
for name, gx, wx, n, avg, mn, mx in rows:
    if any(f in name for f in filt):
        print("%-58s groups %5d  n %4d  avg %8.2f us  min %8.2f  max %8.2f" % (name[:58], gx // max(wx, 1), n, avg, mn, mx))
