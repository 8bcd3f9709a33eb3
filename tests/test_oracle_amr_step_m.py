"""oracle/amr_step_m.c (test infrastructure): the time step on hierarchies whose levels are unions of boxes.  With one box
per level it IS oracle/amr_step.c bit for bit; a level cut into more boxes gives the same bits (fine-fine exchange after
every ghost fill); on unions with a re-entrant corner, a disjoint box and a box on the domain side the step runs with
moulins, diffusion and the implicit gap-height solve and conserves what the single-level loop conserves."""
import numpy as np
import pytest

from oracle import pyoracle as po
from suhmo_amd import synthetic as sy

ONE = ([(32, 16, 95, 47)], [(80, 44, 159, 83)])
ONE_PATCHES = ((16, 8, 47, 23), (40, 22, 79, 41))
CUT = ([(32, 16, 63, 47), (64, 16, 95, 31), (64, 32, 95, 47)], [(80, 44, 119, 83), (120, 44, 159, 83)])
UNION = ([(32, 16, 63, 47), (64, 16, 95, 31), (0, 4, 23, 27)],
         [(72, 40, 119, 55), (72, 56, 103, 87), (8, 16, 31, 39)],
         [(160, 88, 207, 103), (24, 40, 47, 63)])
MOULINS = dict(positions=[(30.0e3, 9.0e3), (42.0e3, 5.5e3), (8.0e3, 4.0e3)], sigma=[900.0, 700.0, 800.0], flux=[8.0, 5.0, 3.0])


def make(boxes, m=None, rough=0.5, nx0=64, ny0=32):
    m = dict(sy.A3_MODEL) if m is None else m
    sts = sy.shmip_amrm_states(nx0, ny0, boxes, rough=rough)
    A = po.OracleAmrMModel(nx0, ny0, sts[0][0]["dx"], sts[0][0]["dy"], sy.A3_BC, sy.A3_PHYS, m, boxes, max_box=16, nthreads=2)
    A.set_states(sts)
    return A, sts, m


@pytest.mark.parametrize("variant", ["explicit", "moulins-diffusion-implicit"])
def test_one_box_per_level_is_the_nested_patch_step(variant):
    m = dict(sy.A3_MODEL)
    if variant != "explicit":
        m.update(diffFactor=1.0, use_impl_diff=1, use_moulin_source=1, distributed_input=7.93e-11)
    A, sts, _ = make(ONE, m)
    B = po.OracleAmrModel(64, 32, sts[0][0]["dx"], sts[0][0]["dy"], sy.A3_BC, sy.A3_PHYS, m, ONE_PATCHES, max_box=16, nthreads=2)
    for l in range(3):
        B.set_state(l, sts[l][0])
    if variant != "explicit":
        ia, ib = A.moulin_source(**MOULINS), B.moulin_source(**MOULINS)
        assert np.array_equal(ia, ib)
    for step in range(2):
        assert A.timestep(m["dt"]) == B.timestep(m["dt"])
        for l in range(3):
            for fid in (po.OM_H, po.OM_B, po.OM_MR, po.OM_QWX, po.OM_RE):
                assert np.array_equal(A.field(l, 0, fid), B.field(l, fid)), (step, l, fid)
    A.close(); B.close()


def level_valid(A, l, fid):
    nx, ny = 64 << l, 32 << l
    out = np.full((ny, nx), np.nan)
    for k, (lo0, lo1, hi0, hi1) in enumerate(A.boxes[l]):
        out[lo1:hi1 + 1, lo0:hi0 + 1] = A.field(l, k, fid)[1:-1, 1:-1]
    return out


def test_cutting_a_level_into_boxes_changes_no_bit():
    m = dict(sy.A3_MODEL, diffFactor=1.0, use_impl_diff=1, use_moulin_source=1, distributed_input=7.93e-11)
    A1, _, _ = make(ONE, m)
    A3, _, _ = make(CUT, m)
    assert np.array_equal(A1.moulin_source(**MOULINS), A3.moulin_source(**MOULINS))
    for step in range(2):
        assert A1.timestep(m["dt"]) == A3.timestep(m["dt"])
        for l in (1, 2):
            for fid in (po.OM_H, po.OM_B, po.OM_MR):
                assert np.array_equal(level_valid(A1, l, fid), level_valid(A3, l, fid), equal_nan=True), (step, l, fid)
        assert np.array_equal(A1.field(0, 0, po.OM_H), A3.field(0, 0, po.OM_H))
    A1.close(); A3.close()


def test_union_hierarchy_steps_and_delivers_the_moulin_flux():
    m = dict(sy.A3_MODEL, diffFactor=1.0, use_impl_diff=1, use_moulin_source=1, distributed_input=7.93e-11)
    A, sts, _ = make(UNION, m)
    A.moulin_source(**MOULINS)
    # composite integral of the source term = sum of the moulin fluxes (cells under a finer level do not count)
    total = 0.0
    for l in range(A.nlev):
        src = level_valid(A, l, po.OM_MSRC)
        if l + 1 < A.nlev:
            fine = level_valid(A, l + 1, po.OM_MSRC)
            cov = ~np.isnan(fine[0::2, 0::2])
            src = np.where(cov, 0.0, src)
        dx, dy = sts[l][0]["dx"], sts[l][0]["dy"]
        total += np.nansum(src) * dx * dy
    assert abs(total - sum(MOULINS["flux"])) < 1e-9 * sum(MOULINS["flux"])
    for step in range(2):
        pi, nv = A.timestep(m["dt"])
        assert 1 <= pi <= 20 and nv > 0
    for l in range(A.nlev):
        for k in range(len(A.boxes[l])):
            for fid in (po.OM_H, po.OM_B, po.OM_MR):
                assert np.all(np.isfinite(A.field(l, k, fid)))
    # covered cells hold the average of the finer level's head (CoarseAverage :3138-3141)
    for l in (3, 2, 1):
        fine, coarse = level_valid(A, l, po.OM_H), level_valid(A, l - 1, po.OM_H)
        avg = 0.25 * (fine[0::2, 0::2] + fine[0::2, 1::2] + fine[1::2, 0::2] + fine[1::2, 1::2])
        msk = ~np.isnan(avg)
        assert msk.sum() > 0 and np.max(np.abs(avg[msk] - coarse[msk])) <= 1e-12 * np.max(np.abs(avg[msk]))
    A.close()
