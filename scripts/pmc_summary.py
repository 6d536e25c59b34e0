#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel.
usage: pmc_summary.py DIR [LAST]  (DIR/pmc*/…counter_collection.csv) -> CSV on stdout
LAST: average only over the last LAST dispatches of each kernel (the timed steps of a bench
run: warm-up and settle steps come first), besides the average over all dispatches."""
import csv
import glob
import os
import sys
from collections import defaultdict

last = int(sys.argv[2]) if len(sys.argv) > 2 else 0
vals = defaultdict(list)
for path in glob.glob(os.path.join(sys.argv[1], "pmc*", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            vals[k].append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
w = csv.writer(sys.stdout)  # kernel names hold commas (template arguments): quoted
hdr = ["kernel", "counter", "avg_per_dispatch", "dispatches"]
if last:
    hdr += [f"avg_last_{last}_dispatches"]
w.writerow(hdr)
for (kern, ctr), v in sorted(vals.items()):
    v.sort()
    row = [kern, ctr, f"{sum(x for _, x in v) / len(v):.6g}", len(v)]
    if last:
        t = v[-last:] if len(v) >= last else v
        row.append(f"{sum(x for _, x in t) / len(t):.6g}")
    w.writerow(row)
