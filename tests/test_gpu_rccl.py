"""The native RCCL transport (suhmo_amd/csrc/suhmo_rccl.hip) on a 1-GPU box: a strip that is its OWN
periodic neighbour (world = 1, lo = hi = rank 0) sends its edge rows to itself through
ncclSend/ncclRecv.  A level of ny rows declared as the lower strip of a 2 ny periodic domain then sees,
in its ghost rows, exactly the periodic image -- so every result must equal the whole periodic level of
ny rows BIT FOR BIT.  (Pack/unpack kernels, message layout, send/recv order when lo == hi, the stream
ordering with the relax kernels and the MAX all-reduce are all on this path; only the peer differs on
a real multi-GPU node.)"""
import numpy as np
import pytest

from suhmo_amd import synthetic as sy

pytestmark = pytest.mark.gpu


def make_pair(nx, ny, f, ph, halo, transport="rccl"):
    """transport "ipc": the strip's halo rows go through the peer-direct transport (suhmo_ipc.hip: the pack kernel stores into the receive
    slots -- the strip's own, being its own neighbour -- and publishes a number the unpack kernel waits for), reductions through RCCL"""
    from suhmo_amd import level, multigpu
    bc = sy.CONV_BC
    W = level.HipLevel(nx, ny, f["dx"], f["dy"], bc, ph, 0.0, -1.0, 32)
    W.set_inputs(f)
    S = level.HipLevel(nx, ny, f["dx"], f["dy"], bc, ph, 0.0, -1.0, 32, j0=0, ny_global=2 * ny, halo_rows=halo)
    S.set_inputs(f)
    multigpu.attach_rccl(S, 0, 1, periodic_y=True)
    if transport.startswith("ipc"):
        multigpu.ipc_attach(S, 0, 1, True, [multigpu.ipc_export(S)])
    return W, S


@pytest.mark.parametrize("nx,ny,variant,halo", [(64, 32, "simple", 4), (2048, 1024, "fused", 4), (128, 64, "simple", 16), (2048, 1024, "fused", 16), (256, 128, "simple", 24), (2048, 1024, "fused", 24)])
@pytest.mark.parametrize("transport", ["rccl", "ipc", "ipc-eight-workgroups"])
def test_self_neighbour_vcycle_bitwise(nx, ny, variant, halo, transport, monkeypatch):
    from suhmo_amd import capi
    from suhmo_amd.level import F_PHI, F_RES, F_BX, F_BY
    from test_gpu_strips import wrap_ghosts
    if variant == "fused":
        monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "100000")
    if transport == "ipc-eight-workgroups":           # exchange launches of at most 8 workgroups: every workgroup owns several pieces of a message, and messages
        monkeypatch.setenv("SUHMO_IPC_BLOCKS", "8")   # of fewer pieces than that alternate with fuller ones on a channel (the acknowledgements of BOTH counts are waited for)
    f = wrap_ghosts(sy.shmip_fields(nx, ny, ly=2.0e4 * ny / nx * 5), sy.CONV_BC)
    W, S = make_pair(nx, ny, f, sy.A3_PHYS, halo, transport)
    sp = dict(sy.SOLVER_DEFAULT)
    for L in (W, S):
        L.build_mg_coefficients()
        L.vcycle(sp)
        L.vcycle(sp)
        L.residual()
    assert (capi.lib().suhmo_level_ipc_exchanges(S.h) if transport.startswith("ipc") else capi.lib().suhmo_level_rccl_exchanges(S.h)) > 10
    for fid in (F_PHI, F_RES, F_BX):
        assert np.array_equal(W.get(fid), S.get(fid)), fid
    assert np.array_equal(W.get(F_BY)[:-1], S.get(F_BY)[:-1])
    assert W.norm(F_RES, 0) == S.norm(F_RES, 0)          # goes through ncclAllReduce(MAX) on the device value, read back through the pinned slot
    assert W.norm(F_RES, 2) == S.norm(F_RES, 2) and W.dot(F_RES, F_PHI) == S.dot(F_RES, F_PHI)     # ncclAllReduce(SUM); world = 1: the same bits
    n1, h1 = W.solve(dict(sp, max_iter=6))
    n2, h2 = S.solve(dict(sp, max_iter=6))
    assert n1 == n2 and np.array_equal(h1, h2)
    assert np.array_equal(W.get(F_PHI), S.get(F_PHI))
    W.close()
    S.close()


def test_detaching_the_peer_direct_transport_gives_the_level_back_to_rccl():
    """suhmo_level_detach_ipc (what multigpu.attach's probe does when a rank reports a failure): the halo rows travel over RCCL again, the cycles stay
    the single level's bit for bit, and the level can be attached once more"""
    from suhmo_amd import capi, multigpu
    from suhmo_amd.level import F_PHI
    from test_gpu_strips import wrap_ghosts
    f = wrap_ghosts(sy.shmip_fields(256, 128, ly=2.0e4 * 128 / 256 * 5), sy.CONV_BC)
    W, S = make_pair(256, 128, f, sy.A3_PHYS, 8, "ipc")
    sp = dict(sy.SOLVER_DEFAULT)
    lib = capi.lib()
    for L in (W, S):
        L.build_mg_coefficients()
        L.vcycle(sp)
    n_ipc, n_rccl = lib.suhmo_level_ipc_exchanges(S.h), lib.suhmo_level_rccl_exchanges(S.h)
    assert n_ipc > 0
    S.synchronize()
    assert lib.suhmo_level_detach_ipc(S.h) == 0 and lib.suhmo_level_ipc_exchanges(S.h) == -1
    W.vcycle(sp); S.vcycle(sp)
    assert lib.suhmo_level_rccl_exchanges(S.h) > n_rccl
    assert np.array_equal(W.get(F_PHI), S.get(F_PHI))
    multigpu.ipc_attach(S, 0, 1, True, [multigpu.ipc_export(S)])
    W.vcycle(sp); S.vcycle(sp)
    assert lib.suhmo_level_ipc_exchanges(S.h) > 0 and np.array_equal(W.get(F_PHI), S.get(F_PHI))
    W.close(); S.close()


@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("direct", [0, 1, "ipc"])
def test_self_neighbour_timestep_bitwise(impl, direct, monkeypatch):
    """suhmo_level_timestep on a strip coupled through the native RCCL hooks (its own periodic neighbour): gap-height,
    melt-rate, gradient and RHS halos, the MAX all-reduced Picard test and -- impl = 1 -- the implicit gap-height solver
    sharing the strip's communicator; must equal the whole periodic level bit for bit."""
    ipc = direct == "ipc"                                     # the halo rows peer-direct (suhmo_ipc.hip), the implicit gap solver's handle sharing the arena
    monkeypatch.setenv("SUHMO_RCCL_DIRECT", "0" if ipc else str(direct))      # 1: halo rows sent from / received into the canvas itself
    from suhmo_amd import model, multigpu
    from test_gpu_timestep import perturbed_state
    from test_gpu_timestep_strips import wrap, NAMES
    from test_gpu_moulin import moulins
    nx, ny, bc = 128, 64, sy.CONV_BC
    m = dict(sy.A3_MODEL, diffFactor=1.0, use_impl_diff=impl, use_moulin_source=1, distributed_input=7.93e-11)
    st = wrap(perturbed_state(nx, ny, 23), bc)
    pos, sg, fl = moulins(5, 9)
    W = model.HipModel(nx, ny, st["dx"], st["dy"], bc, sy.A3_PHYS, m, max_box=16)
    S = model.HipModel(nx, ny, st["dx"], st["dy"], bc, sy.A3_PHYS, m, max_box=16, j0=0, ny_global=2 * ny, halo_rows=4)
    for G in (W, S):
        G.set_state(st)
    multigpu.attach_rccl(S.level, 0, 1, periodic_y=True)
    if ipc:
        multigpu.ipc_attach(S.level, 0, 1, True, [multigpu.ipc_export(S.level)])
    # the strip believes the level has 2 ny rows, so its own moulin integrals would differ: it gets W's source term
    W.moulin_source(pos, sg, fl, 1.0)
    from suhmo_amd import level as lv
    S.level.set(lv.F_MSRC, W.get("msrc"))
    for k in range(2):
        assert W.timestep(m["dt"]) == S.timestep(m["dt"])
        for nm in NAMES:
            a, b = W.get(nm), S.get(nm)
            if nm == "qwy":
                a, b = a[:-1], b[:-1]
            assert np.array_equal(a, b, equal_nan=True), (k, nm)
    W.close()
    S.close()
