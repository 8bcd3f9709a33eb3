#!/usr/bin/env python3
"""The kernels of a rocprofv3 --kernel-trace CSV in launch order, run-length compressed, between the k-th and (k+1)-th occurrence of a marker
kernel: one cycle of a repeating pattern as a readable list.  (Trace hierarchies with SUHMO_GRAPH_MAX_CELLS=0: rocprofv3 7.2 crashes
inside hipGraph capture.)  usage: trace_sequence.py <kernel_trace.csv> <marker substring> [k = 20]"""
import csv, os, sys
if len(sys.argv) < 3 or not os.path.isfile(sys.argv[1]):
    sys.exit(__doc__)
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat, k = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20
nm = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:48]
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
a, b = idx[k], idx[k + 1]
seq, t0 = rows[a:b], int(rows[a]["Start_Timestamp"])
print("# %d launches, %.1f us from the start of the first to the start of the next '%s'" % (len(seq), (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, pat))
out, prev, cnt, dur = [], None, 0, 0.0
for r in seq + [None]:
    key = (nm(r), r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "")) if r else None
    if key != prev:
        if prev: out.append("%3d x %-48s grid %-9s %7.1f us" % (cnt, prev[0], prev[1], dur))
        prev, cnt, dur = key, 0, 0.0
    if r: cnt += 1; dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("\n".join(out))
