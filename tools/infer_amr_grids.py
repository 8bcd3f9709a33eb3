#!/usr/bin/env python3
"""Which refined regions did the reference's AMR convergence runs have?  exec/0_convergence_channelized/{k}lev_base (one AMR
level) and {k}lev_base2levs (two) regrid by tagging the melt rate (input.hydro:77-82), so their grids are not in the inputs; but
the RHS_moulin column of CONV_ANA/results/convergence_data_{2Levels,3Levels}.dat -- the composite L2 difference between the
hierarchy's moulin source term and the single-level run two (three) refinements finer -- depends on nothing but the grids
(Calc_moulin_integral / Calc_moulin_source_term_distributed, src/AmrHydro.cpp:1866-2066, need no solve).  This script scans
the rectangles the block factor allows (x from the outflow boundary to a few blocks past the moulin, y around it) and prints
those that reproduce the column to its 5 printed digits.  numpy restatement of the moulin quadrature (search only; the tests
use the oracle and the device on the grids found here, tests/golden/convergence_channelized_amr_grids.json)."""
import itertools
import json
import os
import sys
import numpy as np

LX, LY = 64.0, 16.0
MX, MY, SIG, FLUX = 16.015625, 8.015625, 1.0, 30.0
V = np.array([0.5555555555, 0.8888888888, 0.5555555555])
LQ = np.array([-0.77459666924 / 2.0, 0.0, 0.77459666924 / 2.0])


def ms_level(nx, ny):
    """Gauss-Legendre sample of the Gaussian per cell of a nx x ny level (whole domain)"""
    dx, dy = LX / nx, LY / ny
    x = (np.arange(nx)[:, None] + 0.5 + LQ[None, :]) * dx - MX          # nx x 3
    y = (np.arange(ny)[:, None] + 0.5 + LQ[None, :]) * dy - MY
    gx = np.exp(-0.5 / (SIG * SIG) * x * x) @ V                          # separable: sum_a v_a exp(-ex_a^2 / 2 s^2)
    gy = np.exp(-0.5 / (SIG * SIG) * y * y) @ V
    return (1.0 / (SIG * np.sqrt(2.0 * 3.14))) * np.outer(gy, gx)


_cache = {}
def ms(nx):
    if nx not in _cache:
        _cache[nx] = ms_level(nx, nx // 4)
    return _cache[nx]


def composite_error(nx0, boxes, nx_exact):
    """boxes: one physical rectangle (x0, x1, y0, y1) per AMR level (nested); returns the composite L2 error vs the single level"""
    nlev = 1 + len(boxes)
    exact = ms(nx_exact)
    exact = exact * FLUX / (exact.sum() * (LX / nx_exact) ** 2)
    # index boxes per level, coverage masks
    idx = []
    for l in range(1, nlev):
        nx = nx0 << l
        dx = LX / nx
        x0, x1, y0, y1 = boxes[l - 1]
        idx.append((int(round(x0 / dx)), int(round(x1 / dx)), int(round(y0 / dx)), int(round(y1 / dx))))
    integ, fields = 0.0, []
    for l in range(nlev - 1, -1, -1):
        nx = nx0 << l
        dx = LX / nx
        m = ms(nx).copy()
        valid = np.ones(m.shape, bool)
        if l > 0:
            i0, i1, j0, j1 = idx[l - 1]
            valid[:] = False; valid[j0:j1, i0:i1] = True
        if l < nlev - 1:
            i0, i1, j0, j1 = idx[l]
            valid[j0 // 2:j1 // 2, i0 // 2:i1 // 2] = False
        integ += m[valid].sum() * dx * dx
        fields.append((l, m, valid, dx))
    tot = 0.0
    for l, m, valid, dx in fields:
        r = nx_exact // (nx0 << l)
        avg = exact.reshape(exact.shape[0] // r, r, exact.shape[1] // r, r).mean(axis=(1, 3))
        e = m * FLUX / integ - avg
        tot += np.sum(e[valid] ** 2) * dx * dx
    return float(np.sqrt(tot))


def close(a, b):
    return abs(a - b) <= 0.6 * 10 ** (np.floor(np.log10(b)) - 4)      # 5 printed digits


def scan_one(nx0, block, target, nx_exact):
    out = []
    for x1 in np.arange(16.0 + block, 16.0 + 12.0 + 1e-9, block):
        for lo in np.arange(block, 8.0 + 1e-9, block):
            for hi in np.arange(block, 8.0 + 1e-9, block):
                e = composite_error(nx0, [(0.0, x1, 8.0 - lo, 8.0 + hi)], nx_exact)
                if close(e, target):
                    out.append(((0.0, float(x1), float(8.0 - lo), float(8.0 + hi)), e))
    return out


def scan_two(nx0, block1, block2, target, nx_exact):
    out = []
    for x1 in np.arange(16.0 + block1, 16.0 + 12.0 + 1e-9, block1):
        for w1 in np.arange(block1, 8.0 + 1e-9, block1):
            b1 = (0.0, float(x1), 8.0 - w1, 8.0 + w1)
            dxc = LX / (nx0 << 1)
            margin = 2 * dxc                                  # nestingRadius 2 cells of level 1
            for x2 in np.arange(16.0 + block2, x1 - margin + 1e-9, block2):
                for lo in np.arange(block2, w1 - margin + 1e-9, block2):
                    for hi in np.arange(block2, w1 - margin + 1e-9, block2):
                        e = composite_error(nx0, [b1, (0.0, float(x2), float(8.0 - lo), float(8.0 + hi))], nx_exact)
                        if close(e, target):
                            out.append(((b1, (0.0, float(x2), float(8.0 - lo), float(8.0 + hi))), e))
    return out


if __name__ == "__main__":
    ref = "/root/reference/exec/0_convergence_channelized/CONV_ANA/results/"
    t2 = {int(float(r[0])): r[5] for r in np.loadtxt(ref + "convergence_data_2Levels.dat")}
    t3 = {int(float(r[0])): r[5] for r in np.loadtxt(ref + "convergence_data_3Levels.dat")}
    # block factor (fine cells) of {k}lev_base / {k}lev_base2levs -> physical block of each AMR level
    bf1 = {32: 2, 64: 8, 128: 16, 256: 32, 512: 64}
    bf2 = {32: 2, 64: 8, 128: 8, 256: 16}
    for nx0, target in sorted(t2.items()):
        blk = bf1[nx0] * LX / (2 * nx0)
        print("2Levels %4d  block %.3g m  target %.5g" % (nx0, blk, target), scan_one(nx0, blk, target, 4 * nx0), flush=True)
    for nx0, target in sorted(t3.items()):
        b1, b2 = bf2[nx0] * LX / (2 * nx0), bf2[nx0] * LX / (4 * nx0)
        print("3Levels %4d  blocks %.3g / %.3g m  target %.5g" % (nx0, b1, b2, target), scan_two(nx0, b1, b2, target, 8 * nx0), flush=True)
