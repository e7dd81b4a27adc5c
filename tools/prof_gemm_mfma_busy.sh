#!/bin/bash
# Runs on the GPU box: the MFMA-busy figure of sgemm_tile<128> from the counters.  SQ_VALU_MFMA_BUSY_CYCLES reads 2 147 483 648 = 2^31 for every
# 4096^3 dispatch -- r03 / r05 took that for a saturated counter; it is the EXACT count: 4096^3 / (32 x 32 x 2) v_mfma_f32_32x32x2_f32 x 64 busy
# cycles each (MI355X_MICROARCH.md) = 2^25 x 2^6.  2048^3 and 4096 x 4096 x 512 (an eighth of the work, either way) read 2^28, as they must.
#   MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the SQ counter adds up the busy cycles of every SIMD;
#   GRBM_GUI_ACTIVE is the launch's duration in shader cycles, summed over the 8 XCDs' GRBMs (divide by 8).  That is a fraction of the cycles
#   the part actually ran (effective clock = cycles / duration: ~2.2 GHz under this load); the timing-derived fraction is against 2.4 GHz.
# Counters in their own passes (no --kernel-trace with --pmc).
# -> gpurun_out/gemm_mfma_busy/summary.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/gemm_mfma_busy
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/summary.txt"
for shape in "0 1 2048 2048 2048" "0 0 2048 2048 2048" "1 0 2048 2048 2048" "0 1 4096 4096 512" "0 1 4096 4096 4096" "0 0 4096 4096 4096" "1 0 4096 4096 4096"; do
    tag=$(echo $shape | tr ' ' '_')
    echo "## th_sgemm (ta tb m n k) = $shape" >> "$OUT/summary.txt"
    python $ROOT/tools/gemm_mnk.py $shape 150 > /tmp/gb_line.txt 2>&1; cat /tmp/gb_line.txt >> "$OUT/summary.txt"
    rm -rf /tmp/gb_$tag
    timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/gb_$tag -- python $ROOT/tools/gemm_mnk.py $shape 10 > /dev/null 2>&1
    python - "$tag" "$(grep -o '[0-9.]* us per launch' /tmp/gb_line.txt | cut -d' ' -f1)" >> "$OUT/summary.txt" <<'PY'
import csv, sys
from collections import defaultdict
from pathlib import Path
acc = defaultdict(lambda: defaultdict(list))
for f in Path("/tmp/gb_" + sys.argv[1]).rglob("*counter_collection.csv"):
    for row in csv.DictReader(open(f, newline="")):
        if "sgemm" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    mean = {n: sum(v) / len(v) for n, v in c.items()}
    busy, act = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), mean.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    us = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] else float("nan")
    print(f"{k[:70]}: dispatches {len(next(iter(c.values())))}, mean SQ_VALU_MFMA_BUSY_CYCLES {busy:.0f}, SQ_BUSY_CYCLES {mean.get('SQ_BUSY_CYCLES', 0.0):.0f}, "
          f"GRBM_GUI_ACTIVE / 8 XCDs {act:.0f} cycles (profiled pass) -> MFMA-busy fraction {busy / (act * 1024.0) if act else float('nan'):.3f} of the cycles run; "
          f"against the un-profiled {us:.1f} us at 2.4 GHz: {busy / (us * 1e-6 * 2.4e9 * 1024.0):.3f}")
PY
    rm -rf /tmp/gb_$tag
done
cat "$OUT/summary.txt"
