"""Second, independent restatement of the hot-path arithmetic in vectorised numpy on
GLOBAL arrays (no boxes, no exchange).  Used only to cross-check the C oracle: numpy
evaluates elementwise IEEE double ops without contraction, so with the same expression
association the two must agree bit for bit, and agreement also proves the box
decomposition / exchange logic of the oracle's level shim is value-neutral.

Formulas: SURVEY.md Appendix C (reference src/VCAMRNonLinearPoissonOpF.ChF:119-158,
257-279, 591-598; src/AmrHydroF.ChF:40-63, 94-107, 214-226; src/AmrHydro.cpp:248-309).
Arrays are [j][i]; ghosted arrays have one ghost layer.
"""
import numpy as np


def fill_ghosts(phi, bc, dx, dy, homogeneous=False):
    """valid (ny,nx) -> ghosted (ny+2,nx+2) with periodic wrap / DiriBC(order 1) / NeumBC."""
    ny, nx = phi.shape
    g = np.zeros((ny + 2, nx + 2))
    g[1:-1, 1:-1] = phi
    d = (dx, dy)
    for direction in range(2):
        for side in range(2):
            if direction == 0:
                near = phi[:, 0] if side == 0 else phi[:, -1]
                wrap = phi[:, -1] if side == 0 else phi[:, 0]
            else:
                near = phi[0, :] if side == 0 else phi[-1, :]
                wrap = phi[-1, :] if side == 0 else phi[0, :]
            if bc["periodic"][direction]:
                val = wrap
            else:
                v = 0.0 if homogeneous else bc["value"][direction][side]
                if bc["type"][direction][side] == 0:
                    val = 2.0 * v - near
                else:
                    sgn = -1.0 if side == 0 else 1.0
                    val = near if homogeneous else near + sgn * d[direction] * v
            if direction == 0:
                g[1:-1, 0 if side == 0 else -1] = val
            else:
                g[0 if side == 0 else -1, 1:-1] = val
    return g


def nl_terms(phi, B, Pi, zb, mask, ph):
    """COMPUTENONLINEARTERMS on valid cells; B, Pi, zb, mask given on valid cells."""
    N = Pi - ph["rho_w_g"] * (phi - zb)
    nl = -ph["A"] * B * N * N * N
    dnl = 3.0 * ph["A"] * B * 1000.0 * ph["grav"] * N * N
    br, brmax = ph["cutOffbr"], ph["maxOffbr"]
    if br != 0.0:
        m = br > B
        nl = np.where(m, nl * (1.0 - (br - B) / br), nl)
        dnl = np.where(m, dnl * B / br, dnl)
    m = brmax < B
    nl = np.where(m, nl * (1.0 - (brmax - B) / brmax), nl)
    dnl = np.where(m, dnl * B / brmax, dnl)
    neg = mask < 0.0
    nl = np.where(neg, 0.0, nl)
    dnl = np.where(neg, 0.0, dnl)
    if not ph.get("use_NL", 1):
        nl, dnl = np.zeros_like(nl), np.zeros_like(dnl)
    return nl, dnl


def op(pg, a, bx, by, nl, alpha, beta, dx, dy):
    """L(phi) on valid cells from ghosted phi pg."""
    rdx, rdy = 1.0 / (dx * dx), 1.0 / (dy * dy)
    c = pg[1:-1, 1:-1]
    e, w = pg[1:-1, 2:], pg[1:-1, :-2]
    n, s = pg[2:, 1:-1], pg[:-2, 1:-1]
    return (alpha * a * c
            - beta * (bx[:, 1:] * (e - c) * rdx - bx[:, :-1] * (c - w) * rdx
                      + by[1:, :] * (n - c) * rdy - by[:-1, :] * (c - s) * rdy)
            + nl)


def lam(a, bx, by, alpha, beta, dx, dy):
    out = a * alpha
    out = out + (1.0 / (dx * dx)) * beta * (bx[:, 1:] + bx[:, :-1])
    out = out + (1.0 / (dy * dy)) * beta * (by[1:, :] + by[:-1, :])
    return out


def gsrb_sweep(phi, rhs, a, bx, by, B, Pi, zb, mask, ph, bc, alpha, beta, dx, dy):
    """one levelGSRB: red pass (i+j even) then black pass, BC + NL refreshed per pass."""
    ny, nx = phi.shape
    phi = phi.copy()
    lm = lam(a, bx, by, alpha, beta, dx, dy)
    jj, ii = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    v = slice(1, -1)
    for p in range(2):
        pg = fill_ghosts(phi, bc, dx, dy, False)
        nl, dnl = nl_terms(phi, B[v, v], Pi[v, v], zb[v, v], mask[v, v], ph)
        L = op(pg, a, bx, by, nl, alpha, beta, dx, dy)
        new = phi + (rhs - L) / (1.0e-16 + lm + dnl)
        sel = ((ii + jj + p) % 2) == 0
        phi = np.where(sel, new, phi)
    return phi


def restrict_sum4(f):
    """res(I,J) = ((((0 + f00/4) + f10/4) + f01/4) + f11/4), fine index (2I+a, 2J+b)."""
    out = np.zeros((f.shape[0] // 2, f.shape[1] // 2))
    out = out + f[0::2, 0::2] / 4.0
    out = out + f[0::2, 1::2] / 4.0
    out = out + f[1::2, 0::2] / 4.0
    out = out + f[1::2, 1::2] / 4.0
    return out


def bcoef_update(phi, Bg, maskg, ph, bc, dx, dy):
    """WFlx_level on a single level; returns bx (ny,nx+1), by (ny+1,nx)."""
    ny, nx = phi.shape
    pg = fill_ghosts(phi, bc, dx, dy, False)
    hm = ph.get("use_mask_gradients", 0)
    # MAC normal gradients on valid faces
    gx = (1.0 / dx) * (pg[1:-1, 1:] - pg[1:-1, :-1])          # (ny, nx+1)
    gy = (1.0 / dy) * (pg[1:, 1:-1] - pg[:-1, 1:-1])          # (ny+1, nx)
    if hm:
        mx = (maskg[1:-1, 1:] < 1e-6) | (maskg[1:-1, :-1] < 1e-6)
        my = (maskg[1:, 1:-1] < 1e-6) | (maskg[:-1, 1:-1] < 1e-6)
        gx = np.where(mx, 0.0, gx)
        gy = np.where(my, 0.0, gy)
    G = np.zeros((2, ny + 2, nx + 2))
    G[0, 1:-1, 1:-1] = 0.5 * (gx[:, :-1] + gx[:, 1:])
    G[1, 1:-1, 1:-1] = 0.5 * (gy[:-1, :] + gy[1:, :])
    for c in range(2):
        g = G[c]
        # x ghosts: periodic wrap or linear extrapolation
        if bc["periodic"][0]:
            g[1:-1, 0], g[1:-1, -1] = g[1:-1, -2], g[1:-1, 1]
        else:
            g[1:-1, 0] = 2.0 * g[1:-1, 1] - g[1:-1, 2]
            g[1:-1, -1] = 2.0 * g[1:-1, -2] - g[1:-1, -3]
        if bc["periodic"][1]:
            g[0, 1:-1], g[-1, 1:-1] = g[-2, 1:-1], g[1, 1:-1]
        else:
            g[0, 1:-1] = 2.0 * g[1, 1:-1] - g[2, 1:-1]
            g[-1, 1:-1] = 2.0 * g[-2, 1:-1] - g[-3, 1:-1]
    s = np.sqrt(G[0] * G[0] + G[1] * G[1])
    om, nu = ph["omega"], ph["nu"]
    disc = 1.0 + 4.0 * om * (Bg * Bg * Bg * ph["grav"] * s) / (12.0 * nu * nu)
    Re = (-1.0 + np.sqrt(disc)) / (2.0 * om)
    out = []
    for direction in range(2):
        if direction == 0:
            Ref = 0.5 * (Re[1:-1, 1:] + Re[1:-1, :-1])
            Bf = 0.5 * (Bg[1:-1, 1:] + Bg[1:-1, :-1])
            m, mm1 = maskg[1:-1, 1:], maskg[1:-1, :-1]
        else:
            Ref = 0.5 * (Re[1:, 1:-1] + Re[:-1, 1:-1])
            Bf = 0.5 * (Bg[1:, 1:-1] + Bg[:-1, 1:-1])
            m, mm1 = maskg[1:, 1:-1], maskg[:-1, 1:-1]
        mec = np.where(np.abs(m - mm1) < 1e-10, np.where(m > 0.0, 1.0, -1.0), 0.0)
        if direction == 0:
            mec[:, 0] = 0.0
            mec[:, -1] = 0.0
        else:
            mec[0, :] = 0.0
            mec[-1, :] = 0.0
        num = -(Bf * Bf * Bf * ph["grav"])
        den = 12.0 * nu * (1.0 + om * Ref)
        b = num / den
        if ph.get("cutOffB", 0) > 0:
            b = np.where(mec < 0.0, 0.0, b)
        out.append(b)
    return out[0], out[1]
