#!/usr/bin/env python3
"""Summarise the counter_collection csv files under a tools/pmc_collect.sh output directory: per kernel and counter, the average
per dispatch (summed over the dimension instances rocprofv3 reports)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            if "elementwise" in k or "rocclr" in k or "repack" in k:
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            ndisp[(k, row["Counter_Name"])].add(row.get("Dispatch_Id"))
print("# rocprofv3 --pmc passes under %s: per-dispatch averages" % root)
for k, cs in acc.items():
    print(k[:100])
    for c, v in sorted(cs.items()):
        n = max(1, len(ndisp[(k, c)]))
        print("    %-34s %16.4e   (%d dispatches)" % (c, v / n, n))
