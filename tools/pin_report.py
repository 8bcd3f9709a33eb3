#!/usr/bin/env python3
"""What the committed oracle tables reproduce of the reference's committed result tables: per suite and case the largest deviation of
every column, relative to the column's scale (tests/golden/shmip_<case>_oracle_pin_table.dat against shmip_<case>_postproc_reference.dat;
the settings of each pin: DESIGN.md section 4).  usage: python tools/pin_report.py > tests/golden/PIN_REPORT.txt"""
import json, os
import numpy as np
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
COLS = {"steady": ["x", "Ylength", "discharge", "dischargeEFF", "dischargeINEFF", "recharge_ext", "recharge_melt", "N"],
        "series": ["T_hrs", "T_days", "avgN", "N_LB", "N_MB", "N_HB", "recharge", "discharge"]}
print("# largest |oracle pin - reference| / max|reference column| per column; rows = all rows of the reference's table")
for suite, cases, kind in (("A", ["A%d" % k for k in range(1, 7)], "steady"), ("B", ["B%d" % k for k in range(1, 6)], "steady"),
                           ("E", ["E%d" % k for k in range(1, 6)], "steady"), ("F", ["F%d" % k for k in range(1, 6)], "series")):
    names = COLS[kind]
    print("\nSHMIP %s   (%s)" % (suite, "320 x 8 cross-section table after 10002 steps" if suite in "AB" else
                                  "256 x 8 cross-section table after 5002 steps" if suite == "E" else "1830 daily rows of a five-year seasonal cycle"))
    print("%-5s" % "case" + "".join("%16s" % n for n in names[2:]) + "   rows")
    for c in cases:
        ref = np.loadtxt(os.path.join(G, "shmip_%s_postproc_reference.dat" % c))
        got = np.loadtxt(os.path.join(G, "shmip_%s_oracle_pin_table.dat" % c))[:, :8]
        sel = slice(1, None) if kind == "steady" else slice(None)
        out = []
        for k in range(2, 8):
            rows = sel if (kind == "steady" and k in (2, 3, 4)) else slice(None)          # row 0 of the discharge columns: the outflow face
            sc = np.max(np.abs(ref[rows, k]))
            out.append(np.max(np.abs(got[rows, k] - ref[rows, k])) / sc if sc > 0 else 0.0)
        print("%-5s" % c + "".join("%16.1e" % v for v in out) + "   %d" % ref.shape[0])
