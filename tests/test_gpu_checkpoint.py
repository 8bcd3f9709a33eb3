"""AmrHydro::writeCheckpointFile / restart (src/AmrHydro.cpp:5670-5842, 6213-6246) on the device models: a run that is
interrupted by a checkpoint and restarted from the file continues BIT FOR BIT -- on one level (SHMIP A3), on nested patches and
on a hierarchy of box unions with moulins, diffusion and the implicit gap-height solve."""
import numpy as np
import pytest

from suhmo_amd import synthetic as sy

pytestmark = pytest.mark.gpu
UNION = ([(32, 16, 63, 47), (64, 16, 95, 31), (0, 4, 23, 27)], [(72, 40, 119, 55), (72, 56, 103, 87), (8, 16, 31, 39)])
MOULINS = dict(positions=[(30.0e3, 9.0e3), (42.0e3, 5.5e3), (8.0e3, 4.0e3)], sigma=[900.0, 700.0, 800.0], flux=[8.0, 5.0, 3.0])


def snapshot(M, kind):
    from suhmo_amd import checkpoint
    tree = checkpoint._boxes_of(M)
    from suhmo_amd.model import HipModel
    return [[{nm: L.get(HipModel.FIELDS[nm]) for nm in ("head", "B", "mR", "Pw", "qwx")} for (L, _) in bl] for bl in tree]


@pytest.mark.parametrize("kind", ["single", "nested", "union"])
def test_restart_continues_bit_for_bit(tmp_path, kind):
    from suhmo_amd import model, checkpoint
    import os
    if checkpoint.hdf5_prefix() is None and not os.path.exists(checkpoint.LIB_PATH):
        pytest.skip("no HDF5 C library on this box: the (optional) checkpoint library cannot be built")
    checkpoint.build()
    m = dict(sy.A3_MODEL)
    if kind != "single":
        m.update(diffFactor=1.0, use_impl_diff=1, use_moulin_source=1, distributed_input=7.93e-11)

    def make():
        if kind == "single":
            st = sy.shmip_initial_state(m["nx"], m["ny"], m["lx"], m["ly"])
            M = model.HipModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, max_box=64)
            M.set_state(st)
        elif kind == "nested":
            patches = ((16, 8, 47, 23), (40, 22, 79, 41))
            sts = sy.shmip_amr_states(64, 32, patches, rough=0.5)
            M = model.HipAmrModel(64, 32, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, m, patches, max_box=16)
            for l, st in enumerate(sts):
                M.set_state(l, st)
            M.moulin_source(**MOULINS)
        else:
            sts = sy.shmip_amrm_states(64, 32, UNION, rough=0.5)
            M = model.HipHierModel(64, 32, sts[0][0]["dx"], sts[0][0]["dy"], sy.A3_BC, sy.A3_PHYS, m, UNION, max_box=16)
            M.set_states(sts)
            M.moulin_source(**MOULINS)
        return M
    A = make()
    for _ in range(3):
        A.timestep(m["dt"])
    path = str(tmp_path / ("chk_%s.2d.hdf5" % kind))
    checkpoint.write(path, A, time=3 * m["dt"], dt=m["dt"])
    counts_a = [A.timestep(m["dt"]) for _ in range(3)]
    ref = snapshot(A, kind)
    A.close()
    B = make()                                        # a fresh model on the same grids, state from the file
    hdr = checkpoint.restart(path, B)
    assert hdr["current_step"] == 3 and B.cur_step == 3 and hdr["time"] == 3 * m["dt"]
    counts_b = [B.timestep(m["dt"]) for _ in range(3)]
    assert counts_a == counts_b
    got = snapshot(B, kind)
    for la, lb in zip(ref, got):
        for ba, bb in zip(la, lb):
            for nm in ba:
                assert np.array_equal(ba[nm], bb[nm], equal_nan=True), (kind, nm)
    B.close()
