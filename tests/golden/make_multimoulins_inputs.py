#!/usr/bin/env python3
"""Extracts the moulin table (positions, fluxes, sigmas), the background input and the physics keys of cfg5 from the
reference's input file exec/AMR_multiMoulins/run_C_3lev/input.hydro (lines 13-44) into
tests/golden/multimoulins_inputs.json.  Data only.  Run where /root/reference exists."""
import json, os, re
SRC = "/root/reference/exec/AMR_multiMoulins/run_C_3lev/input.hydro"
HERE = os.path.dirname(os.path.abspath(__file__))
txt = open(SRC).read()
def vals(prefix, key):
    m = re.search(r"^%s\.%s\s*=\s*([^#\n]*)" % (prefix, key), txt, flags=re.M)
    return [float(v) for v in m.group(1).split()]
n = int(vals("suhmo", "n_moulins")[0])
out = {"n_moulins": n, "positions": vals("suhmo", "moulin_position")[: 2 * n], "flux": vals("suhmo", "moulin_flux")[:n],
       "sigma": vals("suhmo", "moulin_sigma")[:n]}
for k in ("GeoFlux", "LatHeat", "IceHeight", "WaterViscosity", "ct", "cw", "turbulentParam", "br", "lr", "A", "cutOffbr", "maxOffbr",
          "diffFactor", "slope", "GapInit", "distributed_input"):
    out[k] = vals("suhmo", k)[0]
out["SlidingVelocity"] = vals("suhmo", "SlidingVelocity")
out["domain_size"] = vals("main", "domain_size")
out["lo_bc"] = vals("bc", "lo_bc"); out["hi_bc"] = vals("bc", "hi_bc")
out["num_cells"] = vals("AmrHydro", "num_cells"); out["fixed_dt"] = vals("AmrHydro", "fixed_dt")[0]
out["eps_PicardIte"] = vals("solver", "eps_PicardIte")[0]
json.dump(out, open(os.path.join(HERE, "multimoulins_inputs.json"), "w"), indent=0)
print(n, len(out["positions"]), len(out["flux"]), len(out["sigma"]))
