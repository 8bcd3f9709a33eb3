#!/usr/bin/env python3
"""Which kernels run right before / after a given kernel in a rocprofv3 --kernel-trace CSV (same queue, by start time)?
usage: trace_around.py <kernel_trace.csv> <name substring>   -- e.g. fillBufferAligned: who asks for the small memsets"""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2]
nm = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
before, after = collections.Counter(), collections.Counter()
for k, r in enumerate(rows):
    if pat in r["Kernel_Name"]:
        if k > 0: before[nm(rows[k - 1])] += 1
        if k + 1 < len(rows): after[nm(rows[k + 1])] += 1
print("# kernels launched right BEFORE '%s' (count)" % pat)
for n, c in before.most_common(12): print("%8d  %s" % (c, n))
print("# ... right AFTER")
for n, c in after.most_common(12): print("%8d  %s" % (c, n))
