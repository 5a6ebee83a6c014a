#!/usr/bin/env python3
"""Per-position statistics of a frame's launches from a rocprofv3 --kernel-trace CSV: does every one of the 2x net's eight
trunkw_kernel launches take the same time, or do some (the first behind head_kernel, the last in front of tail_kernel) pull the
average?   usage: tools/launch_index_stats.py <..._kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq, cur = [], []
for r in rows:
    name = r["Kernel_Name"]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "head_kernel" in name:
        cur = [("head", dur, int(r["Start_Timestamp"]), int(r["End_Timestamp"]))]
    elif cur and ("trunkw_kernel" in name or "trunk2_kernel" in name or "tail_kernel" in name or "tail4_kernel" in name):
        cur.append((name.split("(")[0].split("::")[-1][:22], dur, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        if "tail" in name:
            seq.append(cur)
            cur = []
full = [f for f in seq if len(f) == 10 and f[1][1] > 150]       # full-size frames only (the parity probe's launches are short)
print("%d frames of 10 launches" % len(full))
by = defaultdict(list)
gaps = defaultdict(list)
for f in full:
    for i, (n, d, s, e) in enumerate(f):
        by[i].append(d)
        if i:
            gaps[i].append((s - f[i - 1][3]) / 1e3)
for i in range(10):
    v = sorted(by[i])
    g = sorted(gaps[i]) if i else [0]
    print("launch %d %-22s mean %7.1f  median %7.1f  min %7.1f  max %7.1f us   gap before it (median) %5.2f us" %
          (i, full[0][i][0], sum(v) / len(v), v[len(v) // 2], v[0], v[-1], g[len(g) // 2]))
tot = sorted(f[-1][3] - f[0][2] for f in full)
print("head start -> tail end, median: %.1f us" % (tot[len(tot) // 2] / 1e3))

# slow launches: where are they, what ran beside them?
allk = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in rows]
tw = [(s, e, n) for s, e, n in allk if "trunkw_kernel" in n and (e - s) > 150e3]
med = sorted(e - s for s, e, n in tw)[len(tw) // 2]
t0 = tw[0][0]
slow = [(s, e) for s, e, n in tw if e - s > 1.12 * med]
print("trunkw launches: %d, median %.1f us, mean %.1f us, slower than 1.12 x median: %d (they add %.2f %% to the mean)" %
      (len(tw), med / 1e3, sum(e - s for s, e, n in tw) / len(tw) / 1e3, len(slow),
       100.0 * sum(e - s - med for s, e in slow) / sum(e - s for s, e, n in tw)))
idx = {s: i for i, (s, e, n) in enumerate(tw)}
for s, e in slow[:40]:
    beside = [n for s2, e2, n in allk if s2 < e and e2 > s and "trunkw" not in n]
    print("  launch #%d at %.3f ms: %.1f us %s" % (idx[s], (s - t0) / 1e6, (e - s) / 1e3, ("beside: " + ", ".join(beside)) if beside else ""))
import os
mc = sys.argv[1].replace("kernel_trace", "memory_copy_trace")
if os.path.exists(mc):
    cp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")) for r in csv.DictReader(open(mc))]
    n_over = 0
    for s, e in slow:
        o = [d for s2, e2, d in cp if s2 < e and e2 > s]
        n_over += bool(o)
    print("memory copies in the trace: %d; slow launches that overlap one: %d of %d" % (len(cp), n_over, len(slow)))
    over_all = sum(1 for s, e, n in tw if any(s2 < e and e2 > s for s2, e2, d in cp))
    print("all trunkw launches that overlap a copy: %d of %d" % (over_all, len(tw)))
