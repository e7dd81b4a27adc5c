#!/usr/bin/env python3
"""Per-kernel mean of one rocprofv3 --pmc counter (counter_collection.csv rows are per dispatch)."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root, counter = Path(sys.argv[1]), sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
for f in root.rglob("*counter_collection.csv"):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Dispatches", f"mean_{counter}_per_dispatch"])
for k, (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    w.writerow([k, n, f"{tot / n:.3f}"])
