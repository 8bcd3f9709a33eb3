#!/usr/bin/env python3
"""How busy is the GPU in a rocprofv3 --kernel-trace CSV?  Over the last `frac` of the trace (the timed steps, not the set-up): wall time,
kernel time, idle time between consecutive kernels, and where the idle time sits (by the kernel that FOLLOWS the gap: the launch the
device had to wait for).
usage: trace_busy.py <kernel_trace.csv> [frac = 0.5]"""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - frac * (t1 - t0)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
nm = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:50]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
wall = max(int(r["End_Timestamp"]) for r in rows) - int(rows[0]["Start_Timestamp"])
gaps = collections.defaultdict(lambda: [0, 0])
idle, end = 0, int(rows[0]["End_Timestamp"])
for r in rows[1:]:
    g = int(r["Start_Timestamp"]) - end
    if g > 0:
        idle += g
        gaps[nm(r)][0] += 1; gaps[nm(r)][1] += g
    end = max(end, int(r["End_Timestamp"]))
print("last %.0f %% of the trace: %d kernels, wall %.2f ms, kernel time %.2f ms (%.1f %%), idle between kernels %.2f ms (%.1f %%), mean gap %.2f us"
      % (100 * frac, len(rows), wall / 1e6, busy / 1e6, 100.0 * busy / wall, idle / 1e6, 100.0 * idle / wall, idle / 1e3 / max(1, len(rows) - 1)))
print("# idle time by the kernel that follows the gap: kernel | gaps | total ms | mean us")
for n, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-50s %7d %9.2f %8.2f" % (n, c, g / 1e6, g / 1e3 / c))
