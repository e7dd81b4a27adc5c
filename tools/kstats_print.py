#!/usr/bin/env python3
"""prints the top rows of every rocprofv3 *kernel_stats.csv under a directory: kstats_print.py DIR [ROWS]"""
import csv
import glob
import sys
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < rows:
            print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.2f} us  {r['Percentage']:>6s} %")
