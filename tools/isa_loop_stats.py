#!/usr/bin/env python3
"""Instruction mix of the loop that holds a kernel's MFMAs in a `hipcc -S` listing: the body is taken from the target
label of the first backward branch behind the last MFMA up to that branch.
usage: isa_loop_stats.py listing.s mangled-name-fragment..."""
import collections
import sys

lines = open(sys.argv[1]).read().split('\n')
for frag in sys.argv[2:]:
    start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and frag in l and ':' in l and not l.startswith('\t'))
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    body = [l.strip() for l in lines[start:end]]
    idx = [i for i, l in enumerate(body) if 'v_mfma' in l]
    labels = {l[:-1]: i for i, l in enumerate(body) if l.endswith(':')}
    br = next(i for i in range(idx[-1], len(body)) if (body[i].startswith('s_cbranch') or body[i].startswith('s_branch')) and labels.get(body[i].split()[1], 10**9) < idx[0])
    head = labels[body[br].split()[1]]
    seg = [l for l in body[head:br + 1] if l and not l.startswith(';') and not l.startswith('.') and not l.endswith(':')]
    c = collections.Counter(l.split()[0] for l in seg)
    valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
    print(f"{frag}: loop body {len(seg)} instructions: {c['v_mfma_f32_16x16x4_f32']} mfma, {valu} other VALU, "
          f"{sum(v for k, v in c.items() if k.startswith('ds_'))} LDS, {sum(v for k, v in c.items() if k.startswith('global_') or k.startswith('scratch_'))} global/scratch, "
          f"{sum(v for k, v in c.items() if k.startswith('s_'))} scalar; kernel scratch ops {sum('scratch_' in l for l in body)}")
    print('   ', sorted(((k, v) for k, v in c.items() if k.startswith('v_') and 'mfma' not in k), key=lambda kv: -kv[1])[:14])
