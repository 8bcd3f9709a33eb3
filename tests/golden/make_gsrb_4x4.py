"""Generates tests/golden/gsrb_4x4.json: a hand-derived single-sweep example.

Scalar Python floats only (IEEE double, no fused ops), written directly from the
formulas in SURVEY.md Appendix C -- independent of both the C oracle and numpy
vector code.  Run:  python tests/golden/make_gsrb_4x4.py
"""
import json
import os

nx = ny = 4
dx, dy = 2.0, 1.5
alpha, beta = 0.25, -1.0
phys = dict(A=5e-25, omega=1e-3, nu=1.787e-6, cutOffbr=0.0125, maxOffbr=0.03, rho_w_g=9800.0,
            grav=9.8, cutOffB=0, use_NL=1, use_mask_gradients=0)
bc = dict(type=[[0, 1], [1, 0]], value=[[2.0, 0.125], [-0.25, 5.0]], periodic=[0, 0])


def val(k, a, b):  # deterministic "random-looking" numbers
    return a + (b - a) * (((k * 2654435761) % 1000003) / 1000003.0)


phi = [[val(11 + j * nx + i, 5.0, 40.0) for i in range(nx)] for j in range(ny)]
rhs = [[val(97 + j * nx + i, -1e-5, 1e-5) for i in range(nx)] for j in range(ny)]
aC = [[val(211 + j * nx + i, 0.0, 1.0) for i in range(nx)] for j in range(ny)]
B = [[val(307 + j * (nx + 2) + i, 0.005, 0.04) for i in range(nx + 2)] for j in range(ny + 2)]
Pi = [[val(401 + j * (nx + 2) + i, 1e5, 9e6) for i in range(nx + 2)] for j in range(ny + 2)]
zb = [[val(503 + j * (nx + 2) + i, 0.0, 20.0) for i in range(nx + 2)] for j in range(ny + 2)]
mask = [[1.0 for i in range(nx + 2)] for j in range(ny + 2)]
mask[2][3] = -1.0  # cell (i=2, j=1) masked out
bx = [[-val(601 + j * (nx + 1) + i, 0.1, 1.0) for i in range(nx + 1)] for j in range(ny)]
by = [[-val(701 + j * nx + i, 0.1, 1.0) for i in range(nx)] for j in range(ny + 1)]


def ghost(p, i, j, homogeneous=False):
    """phi at (i,j) with physical BC ghosts (DiriBC order 1 / NeumBC)."""
    if 0 <= i < nx and 0 <= j < ny:
        return p[j][i]
    d, side = (0, 0 if i < 0 else 1) if not (0 <= i < nx) else (1, 0 if j < 0 else 1)
    ni, nj = min(max(i, 0), nx - 1), min(max(j, 0), ny - 1)
    near = p[nj][ni]
    v = 0.0 if homogeneous else bc["value"][d][side]
    if bc["type"][d][side] == 0:
        return 2.0 * v - near
    sgn = -1.0 if side == 0 else 1.0
    return near + sgn * (dx if d == 0 else dy) * v


def nl_terms(p, i, j):
    if mask[j + 1][i + 1] < 0.0:
        return 0.0, 0.0
    b = B[j + 1][i + 1]
    N = Pi[j + 1][i + 1] - phys["rho_w_g"] * (p[j][i] - zb[j + 1][i + 1])
    nl = -phys["A"] * b * N * N * N
    dnl = 3.0 * phys["A"] * b * 1000.0 * phys["grav"] * N * N
    br, brm = phys["cutOffbr"], phys["maxOffbr"]
    if br > b:
        nl = nl * (1.0 - (br - b) / br)
        dnl = dnl * b / br
    if brm < b:
        nl = nl * (1.0 - (brm - b) / brm)
        dnl = dnl * b / brm
    return nl, dnl


rdx, rdy = 1.0 / (dx * dx), 1.0 / (dy * dy)


def L_of(p, i, j):
    c = p[j][i]
    nl, dnl = nl_terms(p, i, j)
    lap = (bx[j][i + 1] * (ghost(p, i + 1, j) - c) * rdx - bx[j][i] * (c - ghost(p, i - 1, j)) * rdx
           + by[j + 1][i] * (ghost(p, i, j + 1) - c) * rdy - by[j][i] * (c - ghost(p, i, j - 1)) * rdy)
    return alpha * aC[j][i] * c - beta * lap + nl, dnl


def lam(i, j):
    l = aC[j][i] * alpha
    l = l + rdx * beta * (bx[j][i + 1] + bx[j][i])
    l = l + rdy * beta * (by[j + 1][i] + by[j][i])
    return l


res0 = [[rhs[j][i] - L_of(phi, i, j)[0] for i in range(nx)] for j in range(ny)]
p = [row[:] for row in phi]
for rb in (0, 1):
    old = [row[:] for row in p]  # colour-Jacobi: same-colour cells do not see each other
    for j in range(ny):
        for i in range(nx):
            if (i + j + rb) % 2 == 0:
                Lv, dnl = L_of(old, i, j)
                p[j][i] = old[j][i] + (rhs[j][i] - Lv) / (1.0e-16 + lam(i, j) + dnl)

out = dict(inputs=dict(dx=dx, dy=dy, phi=phi, rhs=rhs, aCoef=aC, B=B, Pi=Pi, zb=zb, mask=mask, bx=bx, by=by),
           bc=bc, phys=phys, alpha=alpha, beta=beta,
           expected=dict(phi_after_one_sweep=p, residual_before=res0))
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gsrb_4x4.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote gsrb_4x4.json")
