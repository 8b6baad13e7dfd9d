#!/usr/bin/env python3
"""Summarises the rocprofv3 CSVs written by tools/prof_pmc.sh: per kernel, mean of every
counter over its dispatches (counter_collection.csv) and mean duration (kernel_trace.csv)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: [0.0, 0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (row["Kernel_Name"].split("(")[0].replace("void ", "")[-64:], row["Counter_Name"])
                acc[k][0] += float(row["Counter_Value"])
                acc[k][1] += 1
        print("== %s" % name)
        for (kern, ctr), (s, n) in sorted(acc.items()):
            if "k_decode" in kern or "k_sync" in kern or "k_chase" in kern:
                print("%-62s %-26s mean=%.6g n=%d" % (kern, ctr, s / n, n))
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        print("== %s kernel stats" % name)
        with open(f) as fh:
            for i, line in enumerate(fh):
                if i < 6:
                    print(line.rstrip()[:200])
