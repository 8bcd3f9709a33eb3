#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals for the LAST V-cycle-sized window
and a per-depth breakdown (depth inferred from the grid size).  usage: trace_summary.py <kernel_trace.csv> [n_last]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t = sorted(((r['Kernel_Name'].split('(')[0][:40], int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Grid_Size_X']), int(r['Grid_Size_Y'])) for r in rows), key=lambda x: x[1])
# one V-cycle = from one k_gradcc to the next
idx = [i for i, x in enumerate(t) if 'k_gradcc' in x[0] or 'k_bcoef_fused' in x[0]]
a, b = idx[-2], idx[-1]
win = t[a:b]
print("V-cycle window: %d kernels, span %.1f us, sum of durations %.1f us" % (len(win), (win[-1][2] - win[0][1]) / 1e3, sum(e - s for _, s, e, _, _ in win) / 1e3))
agg = collections.OrderedDict()
for k, s, e, gx, gy in win:
    key = (k, gx, gy)
    c = agg.setdefault(key, [0, 0.0])
    c[0] += 1; c[1] += (e - s) / 1e3
for (k, gx, gy), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%9.1f us  x%-3d %-42s grid %6d x %-5d" % (d, c, k, gx, gy))
