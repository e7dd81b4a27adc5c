#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: calls, avg us, total %, short kernel name."""
import csv
import sys

for path in sys.argv[1:]:
    print("==", path)
    for r in csv.DictReader(open(path)):
        name = r["Name"].split("(")[0].replace("void ", "").replace("th::", "")
        print(f"{int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:10.2f} us {float(r['Percentage']):6.2f}%  {name[:90]}")
