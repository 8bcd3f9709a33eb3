#!/usr/bin/env python3
"""Per kernel AND grid size statistics of a rocprofv3 --kernel-trace CSV (the per-kernel --stats table mixes the depths of the
V-cycle).  usage: stats_by_grid.py <kernel_trace.csv> <n_vcycles>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nv = int(sys.argv[2])
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    key = (name, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r.get("Grid_Size_Z", 1) or 1))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("# kernel | grid (work-items) | calls | avg us | total ms | ms per V-cycle")
tot = 0.0
for (name, grid), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if us / nv < 1.0: continue
    tot += us / nv
    print("%-48s %9d %5d %9.1f %8.2f %8.3f" % (name[:48], grid, c, us / c, us / 1e3, us / 1e3 / nv))
print("# sum of the rows above: %.3f ms per V-cycle" % (tot / 1e3))
