#!/usr/bin/env python3
"""profiles/rNN_mlp_b64_pmc_traffic.json from the two per-kernel PMC summaries of tools/profile_bench.sh:
bytes per launch = (FETCH_SIZE x 2 + WRITE_SIZE) KiB (gfx950 correction, MI355X_MICROARCH.md HBM section)."""
import csv
import json
import sys

fetch_csv, write_csv = sys.argv[1], sys.argv[2]


def read(path):
    with open(path, newline="") as fh:
        rows = list(csv.reader(fh))
    return {r[0]: float(r[2]) for r in rows[1:] if len(r) >= 3}


f, w = read(fetch_csv), read(write_csv)
kernels = {}
for name in f:
    kernels[name] = {"fetch_KiB_raw": f[name], "write_KiB_raw": w.get(name, 0.0),
                     "traffic_bytes_per_launch": int(round((2 * f[name] + w.get(name, 0.0)) * 1024))}
u8 = next((v for k, v in kernels.items() if "u8_to_unit" in k), None)
cal = ""
if u8:
    cal = (" Calibrated on th::u8_to_unit_kernel of the same run: 47.04 MB read / 188.16 MB written algorithmic vs "
           f"2*{u8['fetch_KiB_raw']:.0f} KiB = {2 * u8['fetch_KiB_raw'] * 1024 / 1e6:.2f} MB and {u8['write_KiB_raw']:.0f} KiB = "
           f"{u8['write_KiB_raw'] * 1024 / 1e6:.2f} MB measured.")
print(json.dumps({
    "command": "tools/profile_bench.sh: rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} (separate passes) -- python bench.py --gpus 1 --steps 192 --warmup 32 --no-cpu-baseline",
    "units": "rocprofv3 reports KiB; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE x2." + cal,
    "note": "memory-side (fabric) requests of the 8 XCD L2s; Infinity-Cache hits are counted, so this is an upper bound of HBM traffic.",
    "kernels": kernels}, indent=1))
