#!/usr/bin/env python3
"""The one thing the reference says about iteration counts (docs/GettingStarted.md:140, the tutorial run exec/0_convergence_channelized/1lev,
32 x 8 cells, 3000 steps): "The first 50 timesteps exhibit from 2 to 3 Picard iterations and over 30 FASMG iterations while the initial state
gets settled.  Then the moulin input ramps up and as many as 7 Picard iterations are required for another 200-300 iterations ... Steady state
is reached soon after."  The solve loop is in the un-vendored Chombo fork (AMRFASMultiGrid); this sweeps, ON THE ORACLE (CPU), the ways it may
differ from upstream's solveNoInit and reports which reproduce the statement:
    stop rule: exit on normThresh as upstream / only on eps x initial norm;  imin: postpones the hang test (upstream) / a hard minimum of cycles;
    iterMin, imin per solve / per time step (first solve only);  bottom: numBottom relaxes / followed by RelaxSolver (imax 40).
usage: python tools/stopping_rule_sweep.py [steps]   (one oracle run of 3000 steps per variant, a few seconds each)"""
import itertools, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import test_oracle_timeloop as t
    pv = t.tutorial_iteration_counts("oracle")
    p, v = pv[:, 0], pv[:, 1]
    first, ramp, late = slice(1, 49), slice(49, 600), slice(1000, 3000)
    print("%d %d | %d %d %.0f %d %.1f | %d %d | %d %.1f %.1f" % (
        p[0], v[0], p[first].min(), p[first].max(), np.median(v[first]), v[first].min(), np.mean(v[first] / p[first]),
        p[ramp].max(), int(np.sum(p[ramp] >= 2)), p[late].max(), np.mean(p[late]), np.mean(v[late])))
    sys.exit(0)

print("# stop  imin-hard  min-first-solve-only  bottom | step 1: Picard V-cycles | steps 2-49: Picard min max, V-cycles per step median min, per solve mean |"
      " steps 50-600: max Picard, steps with >= 2 | steps 1001-3000: max Picard, mean Picard, mean V-cycles per step | matches the statement")
for no_thresh, imin_hard, first_only, bottom in itertools.product((0, 1), (0, 1), (0, 1), (0, 1)):
    env = dict(os.environ, SUHMO_ORACLE_STOP=str(no_thresh | (imin_hard << 1)), SUHMO_ORACLE_BOTTOM=str(bottom))
    env.pop("SUHMO_ORACLE_MIN_FIRST_SOLVE_ONLY", None)
    if first_only:
        env["SUHMO_ORACLE_MIN_FIRST_SOLVE_ONLY"] = "1"
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out = r.stdout.decode().strip().splitlines()
    if r.returncode or not out:
        print("%d %d %d %d | run failed: %s" % (no_thresh, imin_hard, first_only, bottom, r.stderr.decode().strip().splitlines()[-1:]))
        continue
    q = out[-1].replace("|", " ").split()
    pmin, pmax, vmed = int(q[2]), int(q[3]), float(q[4])
    ramp_max = int(q[7])
    ok = pmin >= 1 and pmax <= 3 and vmed > 30 and 5 <= ramp_max <= 8 and int(q[8]) >= 200
    print("%d %d %d %d | %s | %s" % (no_thresh, imin_hard, first_only, bottom, out[-1], "YES" if ok else "no"), flush=True)
