#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel over all dispatches.
usage: pmc_summary.py DIR  (DIR/pmc*/…counter_collection.csv) -> CSV on stdout"""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(sys.argv[1], "pmc*", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            a = acc[k]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
w = csv.writer(sys.stdout)  # kernel names hold commas (template arguments): quoted
w.writerow(["kernel", "counter", "avg_per_dispatch", "dispatches"])
for (kern, ctr), (tot, cnt) in sorted(acc.items()):
    w.writerow([kern, ctr, f"{tot / cnt:.6g}", cnt])
