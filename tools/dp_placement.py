#!/usr/bin/env python3
"""Where the workgroups of a shared-device data-parallel run sat: reads the `[dp ring rank R] ...` lines the event-ring instrumentation prints
(experiments/dp_event_ring.patch; TAPER_DP_POSTMORTEM=1 dumps them when an exchange times out) and shows, for the exchange step at which a
rank fell behind: the CUs (XCC, shader engine, CU from HW_ID) that held the other ranks' waiting workgroups, the CUs left free per shader
engine, and which workgroups of the late rank's first launch were placed (and where) or found no CU until the others gave up.
usage: dp_placement.py RING_LINES.txt N_RANKS [EXCHANGE_STEP]     (default: the step with the largest spread between the ranks)"""
import collections
import re
import sys

W = int(sys.argv[2])
ev = []
for l in open(sys.argv[1], errors="replace"):
    m = re.search(r"\[dp ring rank (\d+)\] t (\d+) kind (\d+) xcc (\d+) a (\d+) step (\d+) hwid (\d+)", l)
    if m:
        ev.append(tuple(int(x) for x in m.groups()))          # (rank, t, kind, xcc, a, step, hwid)


def cu(h):
    return ((h >> 13) & 7, (h >> 8) & 15)                     # HW_ID: SE_ID, CU_ID


tails = [e for e in ev if e[2] == 3]
k1s = [e for e in ev if e[2] == 4]
first = {}
for e in tails:
    k = (e[0], e[5])
    first[k] = min(first.get(k, 1 << 62), e[1])
allpos = collections.defaultdict(set)
for e in ev:
    allpos[e[3]].add(cu(e[6]))
best = None
for s in sorted(set(s for _, s in first)):
    ts = {r: first[(r, s)] for r in range(W) if (r, s) in first}
    if len(ts) == W and (len(sys.argv) < 4 or s == int(sys.argv[3])):
        d = max(ts.values()) - min(ts.values())
        if best is None or d > best[0]:
            best = (d, s, ts)
if best is None:
    sys.exit("no exchange step with every rank's workgroups in the ring")
d, S, ts = best
X, T = max(ts, key=ts.get), min(ts.values())
occ = collections.defaultdict(list)
for e in tails:
    if e[0] != X and e[5] == S:
        occ[(e[3],) + cu(e[6])].append(e[0])
gave_up = d > 100_000_000
xs = sorted([e for e in k1s if e[0] == X and T < e[1] < min(ts[X], T + 400_000_000)], key=lambda e: e[1])
late = sorted([e for e in k1s if e[0] == X and e[1] >= T + 400_000_000], key=lambda e: e[1]) if gave_up else []
print(f"exchange step {S}: rank {X} entered its exchange {d / 100:.1f} us after the first rank" + (" -- after the others had given up" if gave_up else ""))
print(f"waiting workgroups of the other ranks: {sum(len(v) for v in occ.values())} on {len(occ)} CUs ({dict(collections.Counter(len(v) for v in occ.values()))} per CU)")
n_late = 0
for x in range(8):
    taken = set(k[1:] for k in occ if k[0] == x)
    free = collections.Counter(p[0] for p in allpos[x] - taken)
    placed = [f"{e[4]}@se{cu(e[6])[0]}cu{cu(e[6])[1]}{'*' if (e[3],) + cu(e[6]) in occ else ''}+{(e[1] - T) / 100:.1f}us" for e in xs if e[3] == x]
    n_placed = len(placed)
    waiting = [e[4] for e in late if e[3] == x][:max(0, 4 - n_placed)]
    n_late += len(waiting)
    print(f"  xcc {x}: {len(taken):2d} CUs hold a waiting workgroup; free CUs per shader engine {[free.get(s, 0) for s in range(4)]}; "
          f"late rank's first-launch workgroups placed: {' '.join(placed) or '-'}" + (f"; NOT placed until the others gave up: {waiting}" if waiting else ""))
print("(block@seNcuM+time after the first waiting workgroup; * = on a CU that already held a waiting workgroup)")
