#!/bin/bash
# Runs on the GPU box: PMC counters of the simple CNN's chain launch (th_conv_chain_head_fwd) at 1 024 images, the one-workgroup-per-CU instance
# (TAPER_CHAIN_LEAN=0: conv_chain_simple_kernel<true, 10, false, false>, 244 registers) beside the two-to-a-CU instance (=1: <.., true>, 128
# registers, half-pass k loop), one --pmc pass per counter set (no kernel trace in the same run).  MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES /
# (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); LDS conflict ratio = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  -> gpurun_out/chain_lean_pmc/summary.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/chain_lean_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
N=${1:-1024}
for lean in 0 1; do
  i=0
  for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
    i=$((i+1))
    rm -rf /tmp/clp_${lean}_$i
    TAPER_CHAIN_LEAN=$lean timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d /tmp/clp_${lean}_$i -- python $ROOT/tools/chain_lean_ab.py --child $N /tmp/clp.npz > /dev/null 2>&1
  done
done
python - "$OUT" "$N" <<'PY'
import csv, sys, glob
from collections import defaultdict
out, n = sys.argv[1], sys.argv[2]
with open(out + "/summary.txt", "w") as fh:
    for lean in (0, 1):
        acc = defaultdict(list)
        for f in glob.glob(f"/tmp/clp_{lean}_*/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f, newline="")):
                if "conv_chain_simple" in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        m = {k: sum(v) / len(v) for k, v in acc.items()}
        fh.write(f"## th_conv_chain_head_fwd, {n} images, TAPER_CHAIN_LEAN={lean} ({'two workgroups per CU, 128 registers' if lean else 'one workgroup per CU, 244 registers'}); mean per dispatch\n")
        for k in sorted(m):
            fh.write(f"    {k:28s} {m[k]:16.0f}\n")
        if m.get("GRBM_GUI_ACTIVE"):
            fh.write(f"    MFMA-busy fraction (cycles run)   {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}\n")
        if m.get("SQ_LDS_IDX_ACTIVE"):
            fh.write(f"    LDS bank-conflict ratio           {m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:.3f}\n")
        if m.get("SQ_WAVE_CYCLES"):
            fh.write(f"    wait (parked) / wave cycles       {m.get('SQ_WAIT_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}   issue stall / wave cycles {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}\n")
print(open(out + "/summary.txt").read())
PY
