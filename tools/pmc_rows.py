#!/usr/bin/env python3
"""tools/pmc_rows.py DIR PATTERN -- per-dispatch rows of the PMC counters rocprofv3 collected under DIR for the kernels whose
name contains PATTERN (one line per dispatch: the launches of one kernel can differ by orders of magnitude).  Diagnostic."""
import csv, glob, sys
from collections import defaultdict
rows = defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            rows[(f, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(rows, key=lambda k: (k[0], int(k[1]))):
    print(k[1], {n: "%.4g" % v for n, v in sorted(rows[k].items())})
