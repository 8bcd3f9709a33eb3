#!/usr/bin/env python3
"""pmc summary (tools/pmc_any.sh output) -> profiles/r01_pmc_traffic_gsrb.json: HBM bytes per launch of the depth-0
k_gsrb_fused kernel.  FETCH_SIZE is doubled for this kernel's 16-byte loads as MI355X_MICROARCH.md prescribes for
gfx950 (cross-check: TCC_EA0_RDREQ_sum x 128 B).  usage: make_traffic_json.py <pmc_summary.txt> <cells> <sweeps_per_launch> <out.json>"""
import json, re, sys


def vcycle_mode(outdir, ka, kb, out):
    """--vcycle <dir of tools/pmc_vcycle.sh> <KA> <KB> <out.json>: HBM bytes of one whole V-cycle = (bytes of the run with KB timed cycles
    - bytes of the run with KA) / (KB - KA), per kernel and in total.  FETCH_SIZE (KB) x 2 as MI355X_MICROARCH.md prescribes for gfx950
    (every kernel of the library reads with wide coalesced loads); cross-check: TCC_EA0_RDREQ_sum x 128 B."""
    import collections, csv, glob
    tot = {}
    for k in (ka, kb):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for f in glob.glob("%s/k%d_p*/*/*counter_collection.csv" % (outdir, k)):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:56]
                agg[(name, row.get("Grid_Size", "?"))][row["Counter_Name"]] += float(row["Counter_Value"])
        tot[k] = agg
    dk = kb - ka
    rows, sums = [], collections.defaultdict(float)
    for key in sorted(set(tot[ka]) | set(tot[kb])):
        d = {c: (tot[kb][key].get(c, 0.0) - tot[ka][key].get(c, 0.0)) / dk for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum")}
        rd, wr = d["FETCH_SIZE"] * 1024 * 2, d["WRITE_SIZE"] * 1024
        if rd + wr <= 0:
            continue
        rows.append((rd + wr, key, rd, wr, d["TCC_EA0_RDREQ_sum"] * 128))
        sums["read"] += rd; sums["write"] += wr; sums["rdreq128"] += d["TCC_EA0_RDREQ_sum"] * 128
    print("# HBM bytes per V-cycle by kernel and grid (PMC; FETCH_SIZE x 2 + WRITE_SIZE; RDREQ x 128 B as the cross-check of the reads)")
    print("# %-56s %10s %12s %12s %12s" % ("kernel", "grid", "read MB", "write MB", "rdreq128 MB"))
    for t, key, rd, wr, rq in sorted(rows, reverse=True):
        print("%-58s %10s %12.2f %12.2f %12.2f" % (key[0], key[1], rd / 1e6, wr / 1e6, rq / 1e6))
    print("# total: read %.1f MB + written %.1f MB = %.1f MB per V-cycle (reads by RDREQ x 128 B: %.1f MB)"
          % (sums["read"] / 1e6, sums["write"] / 1e6, (sums["read"] + sums["write"]) / 1e6, sums["rdreq128"] / 1e6))
    j = {"what": "HBM bytes of one whole V-cycle of bench.py's default workload (4096^2, 6 depths), every kernel, from rocprofv3 --pmc passes "
                 "(tools/pmc_vcycle.sh: runs with %d and %d timed cycles, difference / %d)" % (ka, kb, dk),
         "correction": "FETCH_SIZE (KB) x 1024 x 2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE x 1024",
         "hbm_read_bytes_per_vcycle": sums["read"], "hbm_write_bytes_per_vcycle": sums["write"],
         "hbm_bytes_per_vcycle": sums["read"] + sums["write"], "rdreq_x_128B_per_vcycle": sums["rdreq128"], "cells": 16777216}
    json.dump(j, open(out, "w"), indent=1)


if len(sys.argv) > 1 and sys.argv[1] == "--vcycle":
    vcycle_mode(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    sys.exit(0)
txt = open(sys.argv[1]).read()
cells, spl, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
blocks = re.split(r"\n(?=\S)", txt)
best = None
for b in blocks:
    head = b.split("\n", 1)[0]
    if "k_gsrb_fused" not in head:
        continue
    m = re.search(r"grid=(\d+)", head)
    g = int(m.group(1)) if m else 0
    vals = {mm.group(1): float(mm.group(2)) for mm in re.finditer(r"^\s+(\S+)\s+n=\d+ mean=(\S+)", b, flags=re.M)}
    if "FETCH_SIZE" in vals and (best is None or g > best[0]):
        best = (g, head.strip(), vals)
g, head, v = best
rd = v["FETCH_SIZE"] * 1024 * 2
wr = v["WRITE_SIZE"] * 1024
d = {"kernel": head + " (%d sweeps per launch), %d cells" % (spl, cells),
     "source": "rocprofv3 --pmc, separate passes (tools/pmc_any.sh on bench.py --no-cpu); FETCH_SIZE doubled per MI355X_MICROARCH.md "
               "(gfx950 reports 1/2 of wide coalesced reads); cross-check TCC_EA0_RDREQ_sum x 128 B",
     "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
     "TCC_EA0_RDREQ_sum": v.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_WRREQ_sum": v.get("TCC_EA0_WRREQ_sum"),
     "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
     "rdreq_x_128B": (v.get("TCC_EA0_RDREQ_sum") or 0) * 128,
     "sweeps_per_launch": spl, "cells": cells, "algorithmic_bytes_per_launch": 72 * cells * spl,
     "hbm_bytes_per_cell_per_launch": (rd + wr) / cells}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d, indent=1))
