#!/bin/bash
# Runs on the GPU box: PMC counters of the conv chain kernels (tools/chain_time.py at batch 256), one --pmc pass per counter set.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/chain_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf /tmp/chpmc_$i
    timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d /tmp/chpmc_$i -- python $ROOT/tools/chain_time.py 256 > /dev/null 2>&1
    for c in $set; do
        python $ROOT/tools/summarize_pmc.py /tmp/chpmc_$i $c | grep -E "conv_chain|wide_grads|Kernel_Name" > "$OUT/$c.csv"
    done
done
python - "$OUT" <<'PY'
import csv, sys, glob, os
from collections import defaultdict
out=sys.argv[1]; tab=defaultdict(dict)
for f in sorted(glob.glob(out+'/*.csv')):
    c=os.path.basename(f)[:-4]
    for row in csv.DictReader(open(f)):
        tab[row['Kernel_Name']][c]=float(list(row.values())[2])
cols=sorted({c for v in tab.values() for c in v})
with open(out+'/summary.txt','w') as fh:
    for k,v in tab.items():
        fh.write(k+'\n')
        for c in cols: fh.write(f'    {c:32s} {v.get(c,float("nan")):16.1f}\n')
        if 'SQ_BUSY_CYCLES' in v and 'SQ_VALU_MFMA_BUSY_CYCLES' in v: fh.write(f'    mfma_busy/busy {v["SQ_VALU_MFMA_BUSY_CYCLES"]/v["SQ_BUSY_CYCLES"]:.3f}\n')
print(open(out+'/summary.txt').read())
PY
