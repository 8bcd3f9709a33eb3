#!/usr/bin/env python3
"""pmc summary (tools/pmc_any.sh output) -> profiles/r01_pmc_traffic_gsrb.json: HBM bytes per launch of the depth-0
k_gsrb_fused kernel.  FETCH_SIZE is doubled for this kernel's 16-byte loads as MI355X_MICROARCH.md prescribes for
gfx950 (cross-check: TCC_EA0_RDREQ_sum x 128 B).  usage: make_traffic_json.py <pmc_summary.txt> <cells> <sweeps_per_launch> <out.json>"""
import json, re, sys
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
