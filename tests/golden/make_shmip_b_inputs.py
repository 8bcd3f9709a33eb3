#!/usr/bin/env python3
"""Extracts the moulin tables (positions, fluxes, sigmas) and background input of the SHMIP suite-B cases from the
reference's input files exec/B_SHMIP/B<k>/input.hydro (lines 37-41: suhmo.n_moulins, moulin_position, moulin_flux,
moulin_sigma, distributed_input) into tests/golden/shmip_B_inputs.json, and copies the committed result tables
exec/B_SHMIP/B<k>/results/postproc.dat as data fixtures.  Run where /root/reference exists."""
import json, os, re, shutil
REF = "/root/reference/exec/B_SHMIP"
HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
for k in range(1, 6):
    txt = open(os.path.join(REF, "B%d" % k, "input.hydro")).read()
    def vals(key):
        m = re.search(r"^suhmo\.%s\s*=\s*([^#\n]*)" % key, txt, flags=re.M)
        return [float(v) for v in m.group(1).split()]
    n = int(vals("n_moulins")[0])
    out["B%d" % k] = {"n_moulins": n, "positions": vals("moulin_position")[: 2 * n], "flux": vals("moulin_flux")[:n],
                      "sigma": vals("moulin_sigma")[:n], "distributed_input": vals("distributed_input")[0],
                      "diffFactor": vals("diffFactor")[0]}
    shutil.copyfile(os.path.join(REF, "B%d" % k, "results", "postproc.dat"), os.path.join(HERE, "shmip_B%d_postproc_reference.dat" % k))
json.dump(out, open(os.path.join(HERE, "shmip_B_inputs.json"), "w"), indent=0)
print({k: v["n_moulins"] for k, v in out.items()})
