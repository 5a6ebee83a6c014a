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
