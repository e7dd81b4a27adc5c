#!/bin/bash
# prints name / calls / avg / min / max (us) of a rocprofv3 kernel_stats.csv
python3 - "$@" <<'PY'
import csv, sys
for f in sys.argv[1:]:
    print("==", f)
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1000:8.1f}us min={float(r['MinNs'])/1000:8.1f} max={float(r['MaxNs'])/1000:8.1f}")
PY
