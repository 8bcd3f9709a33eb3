"""Deterministic synthetic inputs for the head-solve hot path.

Formulas follow the reference's initial-condition code so that the fields have the
magnitudes the kernels see in production:
  * geometry / ice thickness / overburden: SqrtIBC::initializeData
    (src/SqrtIBC.cpp:219-262), SHMIP-A domain (exec/A_SHMIP/A3/input.hydro:2)
  * ice mask: SqrtIBC::setup_iceMask (src/SqrtIBC.cpp:145-167)
  * physics constants: exec/A_SHMIP/A3/input.hydro:15-36, src/suhmo_params.cpp:51-53
All arrays are float64, C order [j][i] (i fastest).  "ghosted" arrays carry one ghost
layer: shape (ny+2, nx+2), cell (i,j) at [j+1, i+1].
"""
import numpy as np

RHO_I, RHO_W, GRAV = 910.0, 1000.0, 9.8

A3_PHYS = dict(A=5e-25, omega=1e-3, nu=1.787e-6, cutOffbr=0.0, maxOffbr=1.0e4,
               rho_w_g=9800.0, grav=9.8, cutOffB=0, use_NL=1, use_mask_gradients=0)

# bc.lo_bc = 0 1 / bc.hi_bc = 1 1, all values 0 (exec/A_SHMIP/A3/input.hydro:8-13)
A3_BC = dict(type=[[0, 1], [1, 1]], value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 0])
# exec/1_convergence_distributed/256x64/input.hydro:8-13,76: y periodic
CONV_BC = dict(type=[[0, 1], [1, 1]], value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 1])

SOLVER_DEFAULT = dict(num_smooth=4, num_bottom=16, max_iter=100, iter_min=2, imin=5,
                      eps=1e-7, hang=0.01, norm_thresh=1e-7, bcoeff_otf=1, max_depth=-1)


def shmip_fields(nx, ny, lx=1.0e5, ly=2.0e4, vary_B=True, seed=12345, ice_height=5000.0,
                 slope=0.0, gap_init=0.01, source=5.79e-9, j0=0, ny_total=None):
    """K-bench inputs.  j0 / ny_total select a strip of rows [j0, j0+ny) of a taller
    domain (used by the multi-GPU weak-scaling bench; random phi is seeded per row
    block so any partition sees the same global field)."""
    ny_total = ny if ny_total is None else ny_total
    dx, dy = lx / nx, ly / ny_total
    i = np.arange(-1, nx + 1, dtype=np.float64)
    j = np.arange(j0 - 1, j0 + ny + 1, dtype=np.float64)
    x = (i + 0.5) * dx
    y = (j + 0.5) * dy
    X, Y = np.meshgrid(x, y)  # (ny+2, nx+2)
    zb = slope * X
    H = np.maximum(6.0 * (np.sqrt(np.maximum(X + ice_height, 0.0)) - np.sqrt(ice_height)) + 1.0, 0.0)
    Pi = np.maximum(RHO_I * GRAV * H, 0.0)
    mask = np.where(Pi > 0.0, 1.0, -1.0)
    B = np.full_like(X, gap_init)
    if vary_B:
        B = B * (1.0 + 0.5 * np.sin(2.0 * np.pi * X / 1.0e4) * np.cos(2.0 * np.pi * Y / 5.0e3))
    # random perturbation of the head, reproducible per global row
    phi = np.empty((ny, nx))
    r = 0
    while r < ny:
        blk_id, off = divmod(j0 + r, 64)
        blk = np.random.default_rng([seed, blk_id]).uniform(-1.0, 1.0, size=(64, nx))
        n = min(64 - off, ny - r)
        phi[r:r + n] = blk[off:off + n]
        r += n
    phi = 101325.0 / (RHO_W * GRAV) + zb[1:-1, 1:-1] + 1.0e-3 * phi
    return dict(nx=nx, ny=ny, dx=dx, dy=dy,
                phi=phi, rhs=np.full((ny, nx), source), aCoef=np.zeros((ny, nx)),
                B=np.ascontiguousarray(B), Pi=np.ascontiguousarray(Pi),
                zb=np.ascontiguousarray(zb), mask=np.ascontiguousarray(mask))


def random_fields(nx, ny, dx=3.0, dy=2.0, seed=7, with_mask_holes=True):
    """Adversarial inputs for kernel parity: random coefficients of realistic magnitude,
    some masked-out cells (mask < 0), gap heights on both sides of cutOffbr / maxOffbr
    so that every branch of COMPUTENONLINEARTERMS (src/AmrHydroF.ChF:40-62) is taken."""
    rng = np.random.default_rng(seed)
    g = (ny + 2, nx + 2)
    B = rng.uniform(0.002, 0.05, size=g)
    Pi = rng.uniform(1.0e5, 1.3e7, size=g)
    zb = rng.uniform(0.0, 50.0, size=g)
    mask = np.ones(g)
    if with_mask_holes:
        mask[rng.uniform(size=g) < 0.1] = -1.0
    phi = rng.uniform(5.0, 900.0, size=(ny, nx))
    rhs = rng.uniform(-1e-5, 1e-5, size=(ny, nx))
    aCoef = rng.uniform(0.0, 1.0, size=(ny, nx))
    bx = -rng.uniform(0.05, 1.0, size=(ny, nx + 1))
    by = -rng.uniform(0.05, 1.0, size=(ny + 1, nx))
    return dict(nx=nx, ny=ny, dx=dx, dy=dy, phi=phi, rhs=rhs, aCoef=aCoef, B=B, Pi=Pi, zb=zb,
                mask=mask, bx=bx, by=by)


RANDOM_PHYS = dict(A=5e-25, omega=1e-3, nu=1.787e-6, cutOffbr=0.01, maxOffbr=0.03,
                   rho_w_g=9800.0, grav=9.8, cutOffB=1, use_NL=1, use_mask_gradients=1)
RANDOM_BC = dict(type=[[0, 1], [1, 0]], value=[[3.0, -0.02], [0.01, 7.0]], periodic=[0, 0])


# ---- whole-model inputs (the caller of the head solve): SHMIP suite A
# exec/A_SHMIP/A3/input.hydro:15-60, src/suhmo_params.cpp:51-106
A3_MODEL = dict(rho_i=910.0, rho_w=1000.0, gravity=9.8, G=0.0, L=3.34e5, ct=7.5e-8, cw=4.22e3, ub=(1.0e-6, 0.0),
                br=0.1, lr=2.0, diffFactor=0.0, distributed_input=5.79e-9, eps_picard=1.0e-4, basal_friction=1,
                use_mask_rhs_b=0, dt=3600.0, max_step=10000, nx=320, ny=64, lx=1.0e5, ly=2.0e4)


# exec/A_SHMIP/A{1..6}/input.hydro:41 -- the suite-A cases differ in the distributed input only
SHMIP_A_INPUT = dict(A1=7.93e-11, A2=1.59e-9, A3=5.79e-9, A4=2.5e-8, A5=4.5e-8, A6=5.79e-7)


def shmip_a_model(case):
    return dict(A3_MODEL, distributed_input=SHMIP_A_INPUT[case])


def shmip_b_model(case, inputs):
    """exec/B_SHMIP/B<k>/input.hydro: suite A physics + moulins (inputs = tests/golden/shmip_B_inputs.json[case]),
    diffFactor 1 and the implicit gap-height solve"""
    b = inputs
    return dict(A3_MODEL, distributed_input=b["distributed_input"], diffFactor=b["diffFactor"], use_impl_diff=1,
                use_moulin_source=1, ramp=1.0)


def shmip_initial_state(nx, ny, lx=1.0e5, ly=2.0e4, ice_height=5000.0, slope=0.0, gap_init=0.01):
    """SqrtIBC::initializeData (src/SqrtIBC.cpp:219-262) evaluated over the ghosted level:
    zb = slope x, H = max(6(sqrt(x+IceHeight)-sqrt(IceHeight))+1, 0), Pi = rho_i g H,
    B = GapInit (1e-16 where Pi < 2), head = 101325/(rho_w g) + zb; mask: SqrtIBC::setup_iceMask."""
    dx, dy = lx / nx, ly / ny
    i = np.arange(-1, nx + 1, dtype=np.float64)
    j = np.arange(-1, ny + 1, dtype=np.float64)
    X, _ = np.meshgrid((i + 0.5) * dx, (j + 0.5) * dy)
    zb = slope * X
    H = np.maximum(6.0 * (np.sqrt(X + ice_height) - np.sqrt(ice_height)) + 1.0, 0.0)
    Pi = np.maximum(RHO_I * GRAV * H, 0.0)
    B = np.where(Pi < 2.0, 1.0e-16, gap_init)
    head = 101325.0 * (1.0 / (RHO_W * GRAV)) + zb
    mask = np.where(Pi > 0.0, 1.0, -1.0)
    return dict(nx=nx, ny=ny, dx=dx, dy=dy, head=head, B=B, Pi=Pi, zb=zb, mask=mask)


def shmip_amr_states(nx0, ny0, patches, lx=1.0e5, ly=2.0e4, ice_height=5000.0, slope=0.0, gap_init=0.01, rough=0.0):
    """SqrtIBC state (shmip_initial_state) sampled on every level of a hierarchy: level 0 = nx0 x ny0 over the domain,
    patches[k] = box (ci0, cj0, ci1, cj1) of level k+1 in the cells of level k.  rough > 0 modulates gap height, head and bed
    with smooth analytic functions of (x, y), the same on every level (tests)."""
    def level(nx, ny, i0, j0, nxg, nyg):
        dx, dy = lx / nxg, ly / nyg
        i = np.arange(i0 - 1, i0 + nx + 1, dtype=np.float64)
        j = np.arange(j0 - 1, j0 + ny + 1, dtype=np.float64)
        X, Y = np.meshgrid((i + 0.5) * dx, (j + 0.5) * dy)
        zb = slope * X + rough * 2.0 * np.sin(2.0 * np.pi * X / (0.37 * lx)) * np.cos(2.0 * np.pi * Y / (0.61 * ly))
        H = np.maximum(6.0 * (np.sqrt(np.maximum(X + ice_height, 0.0)) - np.sqrt(ice_height)) + 1.0, 0.0)
        Pi = np.maximum(RHO_I * GRAV * H, 0.0)
        B = np.where(Pi < 2.0, 1.0e-16, gap_init) * (1.0 + rough * 4.0 * (1.0 + np.sin(2.0 * np.pi * X / (0.23 * lx)) * np.sin(2.0 * np.pi * Y / (0.41 * ly))))
        head = 101325.0 * (1.0 / (RHO_W * GRAV)) + zb + rough * 20.0 * (1.0 + np.cos(2.0 * np.pi * X / (0.53 * lx)) * np.sin(2.0 * np.pi * Y / (0.77 * ly)))
        mask = np.where(Pi > 0.0, 1.0, -1.0)
        return dict(nx=nx, ny=ny, dx=dx, dy=dy, i0=i0, j0=j0, nxg=nxg, nyg=nyg, head=np.ascontiguousarray(head), B=np.ascontiguousarray(B),
                    Pi=np.ascontiguousarray(Pi), zb=np.ascontiguousarray(zb), mask=np.ascontiguousarray(mask))
    out = [level(nx0, ny0, 0, 0, nx0, ny0)]
    nxg, nyg = nx0, ny0
    for ci0, cj0, ci1, cj1 in patches:
        nxg, nyg = 2 * nxg, 2 * nyg
        out.append(level(2 * (ci1 - ci0 + 1), 2 * (cj1 - cj0 + 1), 2 * ci0, 2 * cj0, nxg, nyg))
    return out


def shmip_amrm_states(nx0, ny0, boxes, **kw):
    """shmip_amr_states for a hierarchy whose levels >= 1 are unions of boxes (boxes[l-1] = list of (lo0, lo1, hi0, hi1) in
    the index space of level l): [[level-0 state], [states of the boxes of level 1], ...]; the fields are analytic functions
    of (x, y), so a ghost cell of a box holds the neighbour's value or the function's own."""
    out = [[shmip_amr_states(nx0, ny0, (), **kw)[0]]]
    for l, bl in enumerate(boxes, start=1):
        lev = []
        for (lo0, lo1, hi0, hi1) in bl:
            # a one-patch hierarchy whose patch is this box: patch boxes are given in the cells of the level below
            chain = [(0, 0, (nx0 << k) - 1, (ny0 << k) - 1) for k in range(l - 1)] + [(lo0 // 2, lo1 // 2, hi0 // 2, hi1 // 2)]
            lev.append(shmip_amr_states(nx0, ny0, chain, **kw)[-1])
        out.append(lev)
    return out


def wrap_ghosts(f, bc):
    """periodic directions: caller-side ghost data must be the periodic image (what Chombo's
    exchange would have put there), otherwise 'one box' and 'several ranks' see different input"""
    for k in ("B", "Pi", "zb", "mask"):
        a = f[k]
        if bc["periodic"][1]:
            a[0, :], a[-1, :] = a[-2, :].copy(), a[1, :].copy()
        if bc["periodic"][0]:
            a[:, 0], a[:, -1] = a[:, -2].copy(), a[:, 1].copy()
    if "by" in f and bc["periodic"][1]:
        f["by"][-1, :] = f["by"][0, :]        # the same physical face
    if "bx" in f and bc["periodic"][0]:
        f["bx"][:, -1] = f["bx"][:, 0]
    return f


def shmip_postproc_table(dx, dy, qwx, cd, src, mR, Pw, Pi, mask, rho_w=1000.0):
    """The SHMIP cross-section table of AmrHydro::timeStepFAS (src/AmrHydro.cpp:3647-4102), columns of
    exec/*_SHMIP/*/results/postproc.dat: x[km], Ylength, discharge, dischargeEFF, dischargeINEFF,
    recharge(ext), recharge(melt), mean effective pressure [MPa].  cell arrays are VALID cells (ny, nx),
    cd is ghosted (ny+2, nx+2) (channelisation degree, domain ghosts as stored), qwx is (ny, nx+1)."""
    ny, nx = mR.shape
    cd_ec = 0.5 * (cd[1:-1, 1:] + cd[1:-1, :-1])                       # CellToEdge, x-faces
    q_tot = (qwx * dy).sum(axis=0)
    q_chan = (qwx * dy * cd_ec).sum(axis=0)
    q_dist = (qwx * dy * (1.0 - cd_ec)).sum(axis=0)
    ice = mask > 0.0
    ext = np.where(ice, src * dy * dx, 0.0).sum(axis=0)
    mr = np.where(ice, (mR / rho_w) * dy * dx, 0.0).sum(axis=0)
    ylen = np.where(ice, dy, 0.0).sum(axis=0)
    ok = ice & (Pi > 0.0)
    avp = np.where(ok, Pi - Pw, 0.0).sum(axis=0)
    cnt = ok.sum(axis=0)
    ext = np.cumsum(ext[::-1])[::-1]
    mr = np.cumsum(mr[::-1])[::-1]
    x = (np.arange(nx) + 0.5) * dx / 1e3
    return np.stack([x, ylen, -q_tot[:nx], -q_chan[:nx], -q_dist[:nx], ext, mr, avp / np.maximum(cnt, 1.0) / 1e6], axis=1)


# ---- two-level AMR inputs: cfg3 of BASELINE.json = exec/0_convergence_channelized/2lev_base/input.hydro
# (64 m x 16 m, 64 x 16 base cells, y periodic, slope 0.02, IceHeight 500, A = 2.5e-25, one moulin at
# (16.015625, 8.015625), sigma 1, flux 30) with a FIXED refined box around the moulin (the reference tags on the
# melt rate, which needs the time loop; AmrHydro.grids_file is the reference's own way to fix the grids)
CFG3_PHYS = dict(A3_PHYS, A=2.5e-25)
CFG3_PATCH = (8, 4, 23, 11)          # coarse cells [8..23] x [4..11]: 32 x 16 fine cells around the moulin


def amr_fields(nx0=64, ny0=16, patches=(CFG3_PATCH,), lx=64.0, ly=16.0, slope=0.02, ice_height=500.0, gap_init=0.01,
               moulin=(16.015625, 8.015625, 1.0, 30.0), background=1.0e-11, seed=2024, vary_B=True):
    """List of input dicts, one per AMR level (level 0 = base; patches[k] = box of level k+1 in the index space of
    level k).  Every field is the same analytic function sampled at each resolution (zb = slope x; H, Pi as SqrtIBC;
    rhs = background + a Gaussian moulin, value at the cell centre); the head gets an independent random
    perturbation per level."""
    def level(nx, ny, i0, j0, nxg, nyg, rng):
        dx, dy = lx / nxg, ly / nyg
        i = np.arange(i0 - 1, i0 + nx + 1, dtype=np.float64)
        j = np.arange(j0 - 1, j0 + ny + 1, dtype=np.float64)
        X, Y = np.meshgrid((i + 0.5) * dx, (j + 0.5) * dy)
        zb = slope * X
        H = np.maximum(6.0 * (np.sqrt(np.maximum(X + ice_height, 0.0)) - np.sqrt(ice_height)) + 1.0, 0.0)
        Pi = np.maximum(RHO_I * GRAV * H, 0.0)
        mask = np.where(Pi > 0.0, 1.0, -1.0)
        B = np.full_like(X, gap_init)
        if vary_B:
            B = B * (1.0 + 0.3 * np.sin(2.0 * np.pi * X / (0.5 * lx)) * np.cos(2.0 * np.pi * Y / ly))
        mx, my, sig, flux = moulin
        src = background + flux / (2.0 * np.pi * sig * sig) * np.exp(-((X - mx) ** 2 + (Y - my) ** 2) / (2.0 * sig * sig)) * 1.0e-6
        phi = 101325.0 / (RHO_W * GRAV) + zb + 1.0e-3 * rng.uniform(-1.0, 1.0, size=X.shape)
        v = (slice(1, -1), slice(1, -1))
        return dict(nx=nx, ny=ny, dx=dx, dy=dy, phi=np.ascontiguousarray(phi[v]), rhs=np.ascontiguousarray(src[v]),
                    aCoef=np.zeros((ny, nx)), B=np.ascontiguousarray(B), Pi=np.ascontiguousarray(Pi),
                    zb=np.ascontiguousarray(zb), mask=np.ascontiguousarray(mask))
    out = [level(nx0, ny0, 0, 0, nx0, ny0, np.random.default_rng([seed, 0]))]
    nxg, nyg = nx0, ny0
    for k, (ci0, cj0, ci1, cj1) in enumerate(patches):
        nxg, nyg = 2 * nxg, 2 * nyg
        out.append(level(2 * (ci1 - ci0 + 1), 2 * (cj1 - cj0 + 1), 2 * ci0, 2 * cj0, nxg, nyg, np.random.default_rng([seed, k + 1])))
    return out


def amrm_fields(nx0=64, ny0=16, boxes=(), lx=64.0, ly=16.0, slope=0.02, ice_height=500.0, gap_init=0.01,
                moulin=(16.015625, 8.015625, 1.0, 30.0), background=1.0e-11, seed=2024, vary_B=True):
    """Inputs of a hierarchy whose levels >= 1 are unions of boxes: boxes[l-1] = list of (lo0, lo1, hi0, hi1) in the index
    space of level l.  Returns [level0 dict, [box dicts of level 1], ...].  The analytic functions of amr_fields; the head
    perturbation is drawn once per level over the level's whole domain, so that the same level cut into different boxes
    gets the same data (ghost cells of a box = the neighbour's valid cells or the function's own values)."""
    def whole(nxg, nyg, rng):
        dx, dy = lx / nxg, ly / nyg
        i = np.arange(-1, nxg + 1, dtype=np.float64)
        j = np.arange(-1, nyg + 1, dtype=np.float64)
        X, Y = np.meshgrid((i + 0.5) * dx, (j + 0.5) * dy)
        zb = slope * X
        H = np.maximum(6.0 * (np.sqrt(np.maximum(X + ice_height, 0.0)) - np.sqrt(ice_height)) + 1.0, 0.0)
        Pi = np.maximum(RHO_I * GRAV * H, 0.0)
        mask = np.where(Pi > 0.0, 1.0, -1.0)
        B = np.full_like(X, gap_init)
        if vary_B:
            B = B * (1.0 + 0.3 * np.sin(2.0 * np.pi * X / (0.5 * lx)) * np.cos(2.0 * np.pi * Y / ly))
        mx, my, sig, flux = moulin
        src = background + flux / (2.0 * np.pi * sig * sig) * np.exp(-((X - mx) ** 2 + (Y - my) ** 2) / (2.0 * sig * sig)) * 1.0e-6
        phi = 101325.0 / (RHO_W * GRAV) + zb + 1.0e-3 * rng.uniform(-1.0, 1.0, size=X.shape)
        return dict(dx=dx, dy=dy, phi=phi, rhs=src, B=B, Pi=Pi, zb=zb, mask=mask)

    def cut(w, lo0, lo1, hi0, hi1):
        nx, ny = hi0 - lo0 + 1, hi1 - lo1 + 1
        g = (slice(lo1, hi1 + 3), slice(lo0, hi0 + 3))          # ghosted window (the arrays start at index -1)
        v = (slice(lo1 + 1, hi1 + 2), slice(lo0 + 1, hi0 + 2))
        return dict(nx=nx, ny=ny, dx=w["dx"], dy=w["dy"], box=(lo0, lo1, hi0, hi1), phi=np.ascontiguousarray(w["phi"][v]),
                    rhs=np.ascontiguousarray(w["rhs"][v]), aCoef=np.zeros((ny, nx)), B=np.ascontiguousarray(w["B"][g]),
                    Pi=np.ascontiguousarray(w["Pi"][g]), zb=np.ascontiguousarray(w["zb"][g]), mask=np.ascontiguousarray(w["mask"][g]))
    out = [cut(whole(nx0, ny0, np.random.default_rng([seed, 0])), 0, 0, nx0 - 1, ny0 - 1)]
    nxg, nyg = nx0, ny0
    for l, bl in enumerate(boxes):
        nxg, nyg = 2 * nxg, 2 * nyg
        w = whole(nxg, nyg, np.random.default_rng([seed, l + 1]))
        out.append([cut(w, *b) for b in bl])
    return out


def amr2_fields(nxc=64, nyc=16, patch=CFG3_PATCH, **kw):
    """(coarse, fine): the two-level case of amr_fields"""
    c, f = amr_fields(nxc, nyc, (patch,), **kw)
    return c, f


# ---- SHMIP suite E: valley glacier (exec/E_SHMIP/E<k>/input.hydro, src/ValleyIBC.cpp:120-330)
E_GAMMA = dict(E1=0.05, E2=0.0, E3=-0.1, E4=-0.5, E5=-0.7)
E_PHYS = dict(A3_PHYS, use_mask_gradients=1)            # + cutOffB = 1 for E1, E4, E5 (solver.cut_solve_outside_domain)
E_CUTOFFB = dict(E1=1, E2=0, E3=0, E4=1, E5=1)
E_MODEL = dict(A3_MODEL, G=0.05, ct=0.0, diffFactor=1.0, distributed_input=1.158e-6, use_impl_diff=1, use_mask_rhs_b=1,
               max_step=5000, nx=256, ny=64, lx=6000.0, ly=1500.0)


def shmip_e_model(case):
    return dict(E_MODEL, eps_picard=5.0e-4 if case == "E5" else 1.0e-4)


def valley_initial_state(nx, ny, gamma, lx=6000.0, ly=1500.0, gap_init=0.01):
    """ValleyIBC::initializeData (src/ValleyIBC.cpp:236-330) over the ghosted level: surface
    100 (x+200)^(1/4) + x/60 - (2e10)^(1/4) + 1, bed f(x) + g(y) h(x), Pi = rho_i g max(surface - bed, 0),
    head = Pi / (2 rho_w g) + zb, gap = GapInit (1e-16 where Pi < 1e-10); mask: ValleyIBC::setup_iceMask"""
    dx, dy = lx / nx, ly / ny
    i = np.arange(-1, nx + 1, dtype=np.float64)
    j = np.arange(-1, ny + 1, dtype=np.float64)
    X, Y = np.meshgrid((i + 0.5) * dx, (j + 0.5) * dy - 750.0)
    surf = 100.0 * np.power(X + 200.0, 0.25) + X / 60.0 - np.power(2.0e10, 0.25) + 1.0
    gamma_b = 0.05
    H6 = 100.0 * np.power(6000.0 + 200.0, 0.25) + 6000.0 / 60.0 - np.power(2.0e10, 0.25) + 1.0
    fx = (H6 - 6000.0 * gamma) * X * X / (6000.0 * 6000.0) + gamma * X
    fxg = (H6 - 6000.0 * gamma_b) * X * X / (6000.0 * 6000.0) + gamma_b * X
    gy = 0.5e-6 * np.abs(Y * Y * Y)
    hx = (-4.5 * X / 6000.0 + 5.0) * (surf - fx) / (surf - fxg + 1.0e-16)
    zb = fx + gy * hx
    Pi = RHO_I * GRAV * np.maximum(surf - zb, 0.0)
    B = np.where(Pi < 1.0e-10, 1.0e-16, gap_init)
    head = (Pi * 0.5) * (1.0 / (RHO_W * GRAV)) + zb
    head = np.where(head < 0.0, 0.0, head)                 # ValleyIBC::resetCovered
    mask = np.where(Pi > 0.0, 1.0, -1.0)
    return dict(nx=nx, ny=ny, dx=dx, dy=dy, head=head, B=B, Pi=Pi, zb=zb, mask=mask)


# ---- cfg5: exec/AMR_multiMoulins (problem_type = dino: src/MountainSetupIBC.cpp), 63 moulins on 100 km x 100 km
def multimoulins_inputs():
    """moulin table and physics keys of exec/AMR_multiMoulins/run_C_3lev/input.hydro (tests/golden/multimoulins_inputs.json)"""
    import json, os
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "multimoulins_inputs.json")
    return json.load(open(here))


def multimoulins_setup():
    """(bc, phys, model, moulins) of cfg5 as the reference's input file states them"""
    c = multimoulins_inputs()
    bc = dict(type=[[int(c["lo_bc"][0]), int(c["hi_bc"][0])], [int(c["lo_bc"][1]), int(c["hi_bc"][1])]],
              value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 0])
    phys = dict(A3_PHYS, A=c["A"], omega=c["turbulentParam"], nu=c["WaterViscosity"], cutOffbr=c["cutOffbr"], maxOffbr=c["maxOffbr"])
    model = dict(A3_MODEL, G=c["GeoFlux"], L=c["LatHeat"], ct=c["ct"], cw=c["cw"], ub=tuple(c["SlidingVelocity"]), br=c["br"], lr=c["lr"],
                 diffFactor=c["diffFactor"], distributed_input=c["distributed_input"], eps_picard=c["eps_PicardIte"], basal_friction=1,
                 use_moulin_source=1, use_impl_diff=1, dt=c["fixed_dt"])
    n = c["n_moulins"]
    moulins = dict(positions=np.array(c["positions"]).reshape(n, 2), sigma=np.array(c["sigma"]), flux=np.array(c["flux"]))
    return bc, phys, model, moulins


def mountain_state(nx, ny, i0, j0, nxg, nyg, lx=1.0e5, ly=1.0e5, gap_init=0.01):
    """MountainIBC::initializeData (src/MountainSetupIBC.cpp:151-323) on the ghosted box [i0-1, i0+nx] x [j0-1, j0+ny] of a level of
    nxg x nyg cells, WITHOUT the bed noise: the reference adds dist2(generator) from an unseeded std::default_random_engine
    drawn in per-box iteration order (:174, :270), which is not reproducible across decompositions."""
    dx, dy = lx / nxg, ly / nyg
    i = np.arange(i0 - 1, i0 + nx + 1, dtype=np.float64)
    j = np.arange(j0 - 1, j0 + ny + 1, dtype=np.float64)
    X, Y = np.meshgrid((i + 0.5) * dx, (j + 0.5) * dy)
    ax, by, cst = 1.5e-3, -1.5e-3, 100.0
    step_1 = np.maximum(ax * X + by * Y + cst, 0.0)
    iceH = 2.0 * (ax * X + by * Y + cst + 100.0)

    def finger(angle, x0, y0, a_max, s_gauss, sigma_fn):
        ca, sa = np.cos(angle * 3.14159 / 180.0), np.sin(angle * 3.14159 / 180.0)
        xb, yb = X - x0, Y - y0
        xt, yt = xb * ca + yb * sa, -xb * sa + yb * ca
        a_g = a_max * np.exp(-0.5 / (s_gauss * s_gauss) * yt * yt)
        sg = sigma_fn(yt)
        return a_g * np.exp(-0.5 / (sg * sg) * xt * xt)
    step_2 = finger(35.0, 100000.0, 0.0, 250.0, 50000.0, lambda yt: 12000.0 - 3000.0 * np.minimum(1.0 - (50000.0 - yt) / 50000.0, 1.0))
    step_3 = finger(90.0, 85000.0, 0.0, 250.0, 30000.0, lambda yt: 10000.0)
    step_4 = finger(60.0, 55000.0, 0.0, 100.0, 20000.0, lambda yt: 5000.0)
    step_5 = finger(2.0, 100000.0, 20000.0, 300.0, 35000.0, lambda yt: 6000.0)
    zb = np.maximum(step_1 + step_2 + step_3 + step_4 + step_5, 0.0)
    Pi = RHO_I * GRAV * np.maximum(iceH, 0.0)
    B = np.where(Pi == 0.0, 1.0e-16, gap_init)
    head = Pi * 0.5 * (1.0 / (RHO_W * GRAV)) + zb
    mask = np.where(Pi > 0.0, 1.0, -1.0)
    return dict(nx=nx, ny=ny, dx=dx, dy=dy, i0=i0, j0=j0, nxg=nxg, nyg=nyg, head=np.ascontiguousarray(head), B=np.ascontiguousarray(B),
                Pi=np.ascontiguousarray(Pi), zb=np.ascontiguousarray(zb), mask=np.ascontiguousarray(mask))


def mountain_amrm_states(nx0, ny0, boxes, **kw):
    """mountain_state on every box of a hierarchy (boxes[l-1] = boxes of level l in its own index space)"""
    out = [[mountain_state(nx0, ny0, 0, 0, nx0, ny0, **kw)]]
    for l, bl in enumerate(boxes, start=1):
        out.append([mountain_state(hi0 - lo0 + 1, hi1 - lo1 + 1, lo0, lo1, nx0 << l, ny0 << l, **kw) for (lo0, lo1, hi0, hi1) in bl])
    return out


def boxes_around(points, nx0, ny0, nlev, lx, ly, radius_cells=(6, 5, 4), block=4, max_box=64, nest=2):
    """A fixed multi-box hierarchy around `points` (the reference regrids by tagging melt rate / gap height, i.e. around the
    moulins and the channels that leave them, src/AmrHydro.cpp:4176-4604; regridding is out of scope, a grids file is an input:
    AmrHydro.grids_file :1119-1122).  Level l >= 1 covers the cells of level l-1 within radius_cells[l-1] cells of a point,
    restricted to what level l-1 holds minus `nest` cells (proper nesting), snapped to blocks of `block` cells of level l-1 and cut
    into disjoint rectangles of at most max_box fine cells per side.  Returns boxes[l-1] = list of (lo0, lo1, hi0, hi1) in the
    index space of level l."""
    out = []
    nx, ny = nx0, ny0
    allowed = np.ones((ny, nx), dtype=bool)                  # cells of level l-1 that may be refined
    for l in range(1, nlev):
        dx, dy = lx / nx, ly / ny
        tag = np.zeros((ny, nx), dtype=bool)
        r = radius_cells[min(l - 1, len(radius_cells) - 1)]
        for (px, py) in points:
            ic, jc = int(px / dx), int(py / dy)
            tag[max(jc - r, 0):min(jc + r + 1, ny), max(ic - r, 0):min(ic + r + 1, nx)] = True
        # snap to blocks; a block is refined only if all of it may be
        nbx, nby = nx // block, ny // block
        tb = tag[:nby * block, :nbx * block].reshape(nby, block, nbx, block).any(axis=(1, 3))
        ab = allowed[:nby * block, :nbx * block].reshape(nby, block, nbx, block).all(axis=(1, 3))
        tb &= ab
        mb = max(1, max_box // (2 * block))                    # blocks per box side
        boxes, used = [], np.zeros_like(tb)
        for J in range(nby):
            for I in range(nbx):
                if not tb[J, I] or used[J, I]:
                    continue
                w = 1
                while I + w < nbx and w < mb and tb[J, I + w] and not used[J, I + w]:
                    w += 1
                h = 1
                while J + h < nby and h < mb and tb[J + h, I:I + w].all() and not used[J + h, I:I + w].any():
                    h += 1
                used[J:J + h, I:I + w] = True
                boxes.append((2 * I * block, 2 * J * block, 2 * (I + w) * block - 1, 2 * (J + h) * block - 1))
        if not boxes:
            break
        out.append(boxes)
        # cells of level l that may be refined further: held by level l and `nest` + 1 cells away from its edge
        nx, ny = 2 * nx, 2 * ny
        held = np.zeros((ny, nx), dtype=bool)
        for (lo0, lo1, hi0, hi1) in boxes:
            held[lo1:hi1 + 1, lo0:hi0 + 1] = True
        allowed = held.copy()
        m = 2 * nest + 2                                       # in cells of level l: 2 coarse cells of margin + the interpolation stencil
        pad = np.pad(held, m, constant_values=True)            # the domain boundary needs no margin
        for dj in range(-m, m + 1):
            for di in range(-m, m + 1):
                allowed &= pad[m + dj:m + dj + ny, m + di:m + di + nx]
    return out
