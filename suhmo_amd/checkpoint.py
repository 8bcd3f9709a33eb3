"""Checkpoint files of the hydrology state in Chombo's HDF5 layout (include/suhmo_chk.h, suhmo_amd/csrc/suhmo_chk.cpp):
AmrHydro::writeCheckpointFile / readCheckpointFile / restart (src/AmrHydro.cpp:5670-5842, 5845-6246).  Host side: box data cross
PCIe once per checkpoint.  Works on every model class of suhmo_amd.model (one level, nested patches, unions of boxes)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libsuhmo_chk.so")
_LIB = None
SYMBOLS = ["suhmo_chk_last_error", "suhmo_chk_create", "suhmo_chk_write_level", "suhmo_chk_write_field", "suhmo_chk_close",
           "suhmo_chk_open", "suhmo_chk_read_level", "suhmo_chk_field_ghost", "suhmo_chk_read_field"]
# dataset of the file -> field of the model (None: not a device field, written from `extra` or as a constant)
FIELDS = [("headData", "head"), ("gapHeightData", "B"), ("overburdenPressData", "Pi"), ("velMagData", None), ("bedelevationData", "zb"),
          ("ReData", "Re"), ("iceHeightData", None), ("bumpHeightData", None), ("bumpSpacingData", None), ("meltRateData", "mR"),
          ("iceMaskData", "mask")]


class Header(C.Structure):
    _fields_ = [("max_level", C.c_int), ("finest_level", C.c_int), ("current_step", C.c_int), ("time", C.c_double), ("dt", C.c_double),
                ("cfl", C.c_double), ("is_periodic", C.c_int * 2)]


def hdf5_prefix():
    """installation prefix of an HDF5 C library (include/hdf5.h + lib/libhdf5.so): $HDF5_ROOT / $HDF5_DIR, h5cc, pkg-config, then the
    image's /opt/conda and the usual system prefixes; None if there is none"""
    import shutil
    cands = [os.environ.get("HDF5_ROOT"), os.environ.get("HDF5_DIR")]
    h5cc = shutil.which("h5cc")
    if h5cc:
        cands.append(os.path.dirname(os.path.dirname(os.path.realpath(h5cc))))
    try:
        out = subprocess.run(["pkg-config", "--variable=prefix", "hdf5"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=10).stdout.decode().strip()
        cands.append(out or None)
    except Exception:
        pass
    cands += ["/opt/conda", "/usr", "/usr/local"]
    for c in cands:
        if c and os.path.exists(os.path.join(c, "include", "hdf5.h")) and any(os.path.exists(os.path.join(c, "lib", n)) for n in ("libhdf5.so", "libhdf5.a")):
            return c
    return None


def build(force=False):
    """g++ against an HDF5 C library; raises RuntimeError when there is none (the checkpoint file is optional: the caller decides)"""
    src = [os.path.join(CSRC, "suhmo_chk.cpp"), os.path.join(os.path.dirname(_HERE), "include", "suhmo_chk.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        prefix = hdf5_prefix()
        if prefix is None:
            raise RuntimeError("no HDF5 C library found (set HDF5_ROOT): libsuhmo_chk.so not built")
        subprocess.check_call(["make", "-C", CSRC, "-B", "libsuhmo_chk.so", "HDF5=" + prefix], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libsuhmo_chk.so not built (%s); run __graft_entry__.build()" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, ci, dp, ip = C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.suhmo_chk_last_error.restype = C.c_char_p
        L.suhmo_chk_create.argtypes = [C.POINTER(vp), C.c_char_p, C.POINTER(Header)]
        L.suhmo_chk_write_level.argtypes = [vp, ci, C.c_double, C.c_double, ci, ip, ci, ip]
        L.suhmo_chk_write_field.argtypes = [vp, ci, C.c_char_p, ci, C.POINTER(dp)]
        L.suhmo_chk_close.argtypes = [vp]
        L.suhmo_chk_open.argtypes = [C.POINTER(vp), C.c_char_p, C.POINTER(Header)]
        L.suhmo_chk_read_level.argtypes = [vp, ci, dp, dp, ip, ip, ip, ip, ci]
        L.suhmo_chk_field_ghost.argtypes = [vp, ci, C.c_char_p, ip]
        L.suhmo_chk_read_field.argtypes = [vp, ci, C.c_char_p, C.POINTER(dp)]
        _LIB = L
    return _LIB


def _check(rc):
    if rc:
        raise RuntimeError("libsuhmo_chk: " + lib().suhmo_chk_last_error().decode())


def _ptrs(arrs):
    dp = C.POINTER(C.c_double)
    return (dp * len(arrs))(*[a.ctypes.data_as(dp) for a in arrs])


def write_levels(path, levels, step, time, dt, periodic=(0, 0), max_level=None, cfl=0.5):
    """levels[l] = dict(dx, dy, domain=(lo0, lo1, hi0, hi1), boxes=[(lo0, lo1, hi0, hi1), ...], data={dataset: [ghosted (ny+2, nx+2)
    array per box]}).  The eleven datasets of src/AmrHydro.cpp:5826-5836 are written with one ghost layer."""
    nlev = len(levels)
    hdr = Header(nlev - 1 if max_level is None else max_level, nlev - 1, int(step), float(time), float(dt), float(cfl), (C.c_int * 2)(*[int(p) for p in periodic]))
    h = C.c_void_p()
    _check(lib().suhmo_chk_create(C.byref(h), path.encode(), C.byref(hdr)))
    try:
        for l, lv in enumerate(levels):
            bx = np.ascontiguousarray(np.array(lv["boxes"], dtype=np.int32).reshape(-1, 4))
            dom = (C.c_int * 4)(*[int(v) for v in lv["domain"]])
            _check(lib().suhmo_chk_write_level(h, l, lv["dx"], lv["dy"], 2 if l < hdr.max_level else 0, dom, len(bx), bx.ctypes.data_as(C.POINTER(C.c_int))))
            for name, _ in FIELDS:
                arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in lv["data"][name]]
                for a, b in zip(arrs, bx):
                    assert a.shape == (b[3] - b[1] + 3, b[2] - b[0] + 3), (name, a.shape, tuple(b))
                _check(lib().suhmo_chk_write_field(h, l, name.encode(), 1, _ptrs(arrs)))
        # levels the run allows but has not defined yet: the header of every level <= max_level (dx, prob_domain, ref_ratio), as
        # AmrHydro::writeCheckpointFile writes them and readCheckpointFile expects them (src/AmrHydro.cpp:5798-5821)
        for l in range(nlev, hdr.max_level + 1):
            lv = levels[-1]
            r = 2 ** (l - (nlev - 1))
            d0 = [int(v) for v in lv["domain"]]
            dom = (C.c_int * 4)(d0[0] * r, d0[1] * r, (d0[2] + 1) * r - 1, (d0[3] + 1) * r - 1)
            _check(lib().suhmo_chk_write_level(h, l, lv["dx"] / r, lv["dy"] / r, 2 if l < hdr.max_level else 0, dom, 0, None))
    finally:
        lib().suhmo_chk_close(h)


def read_levels(path):
    """-> (header dict, levels as write_levels takes them)"""
    hdr = Header()
    h = C.c_void_p()
    _check(lib().suhmo_chk_open(C.byref(h), path.encode(), C.byref(hdr)))
    levels = []
    try:
        for l in range(hdr.finest_level + 1):
            dx, dy, ref, nb = C.c_double(), C.c_double(), C.c_int(), C.c_int()
            dom = (C.c_int * 4)()
            _check(lib().suhmo_chk_read_level(h, l, C.byref(dx), C.byref(dy), C.byref(ref), dom, C.byref(nb), None, 0))
            bx = np.zeros((nb.value, 4), dtype=np.int32)
            _check(lib().suhmo_chk_read_level(h, l, None, None, None, None, C.byref(nb), bx.ctypes.data_as(C.POINTER(C.c_int)), nb.value))
            data = {}
            for name, _ in FIELDS:
                g = C.c_int()
                _check(lib().suhmo_chk_field_ghost(h, l, name.encode(), C.byref(g)))
                arrs = [np.zeros((b[3] - b[1] + 1 + 2 * g.value, b[2] - b[0] + 1 + 2 * g.value)) for b in bx]
                _check(lib().suhmo_chk_read_field(h, l, name.encode(), _ptrs(arrs)))
                data[name] = arrs
            levels.append(dict(dx=dx.value, dy=dy.value, ref_ratio=ref.value, domain=tuple(dom), boxes=[tuple(int(v) for v in b) for b in bx], data=data))
    finally:
        lib().suhmo_chk_close(h)
    return dict(max_level=hdr.max_level, finest_level=hdr.finest_level, current_step=hdr.current_step, time=hdr.time, dt=hdr.dt, cfl=hdr.cfl,
                is_periodic=tuple(hdr.is_periodic)), levels


def _boxes_of(model):
    """[(level, box index, HipLevel-like, (lo0, lo1, hi0, hi1))] and per-level (dx, dy, domain) of a model of suhmo_amd.model"""
    from . import model as md
    if isinstance(model, md.HipHierModel):
        lv = model.level
        allb = [[(0, 0, lv[0][0].nx - 1, lv[0][0].ny - 1)]] + model.hier.boxes
        return [[(lv[l][k], allb[l][k]) for k in range(len(lv[l]))] for l in range(len(lv))]
    if isinstance(model, md.HipAmrModel):
        out = []
        for l, L in enumerate(model.levels):
            d = L._desc
            out.append([(L, (d.i0, d.j0, d.i0 + d.nx - 1, d.j0 + d.ny - 1))])
        return out
    return [[(model.level, (0, 0, model.nx - 1, model.ny - 1))]]


def write(path, model, time, dt, periodic=(0, 0), extra=None):
    """AmrHydro::writeCheckpointFile of a device-resident model.  extra: dataset -> constant or list (per level) of lists (per
    box) of ghosted arrays for velMagData / iceHeightData / bumpHeightData / bumpSpacingData (defaults: |ub|, 0, br, lr)."""
    from . import model as md
    m = model.model
    ub = m.get("ub", (0.0, 0.0))
    const = {"velMagData": float(np.hypot(ub[0], ub[1])), "iceHeightData": 0.0, "bumpHeightData": float(m["br"]), "bumpSpacingData": float(m["lr"])}
    const.update(extra or {})
    tree = _boxes_of(model)
    nx0 = tree[0][0][1][2] + 1
    ny0 = tree[0][0][1][3] + 1
    levels = []
    for l, bl in enumerate(tree):
        L0 = bl[0][0]
        data = {}
        for name, fld in FIELDS:
            arrs = []
            for k, (L, b) in enumerate(bl):
                shape = (b[3] - b[1] + 3, b[2] - b[0] + 3)
                if fld is not None:
                    arrs.append(L.get(md.HipModel.FIELDS[fld], ghosted=True))
                elif isinstance(const[name], (int, float)):
                    arrs.append(np.full(shape, float(const[name])))
                else:
                    arrs.append(np.asarray(const[name][l][k], dtype=np.float64))
            data[name] = arrs
        levels.append(dict(dx=L0.dx, dy=L0.dy, domain=(0, 0, (nx0 << l) - 1, (ny0 << l) - 1), boxes=[b for _, b in bl], data=data))
    write_levels(path, levels, model.cur_step, time, dt, periodic)


def restart(path, model):
    """AmrHydro::restart: the state of `path` into a model created on the same grids (the box lists must agree).  Returns the header."""
    from . import level as lv
    hdr, levels = read_levels(path)
    tree = _boxes_of(model)
    assert len(tree) == len(levels), "the checkpoint holds %d levels, the model %d" % (len(levels), len(tree))
    for l, bl in enumerate(tree):
        assert [b for _, b in bl] == levels[l]["boxes"], "level %d: the boxes of the checkpoint differ from the model's" % l
        for k, (L, b) in enumerate(bl):
            d = levels[l]["data"]
            L.set(lv.F_PHI, d["headData"][k][1:-1, 1:-1])
            L.set(lv.F_ACOEF, np.zeros((L.ny, L.nx)))
            for name, fid in (("gapHeightData", lv.F_B), ("overburdenPressData", lv.F_PI), ("bedelevationData", lv.F_ZB), ("iceMaskData", lv.F_MASK)):
                L.set(fid, d[name][k], ghosted=True)
            L.set(lv.F_MR, d["meltRateData"][k][1:-1, 1:-1])
            L.set(lv.F_RE, d["ReData"][k][1:-1, 1:-1])
    model.cur_step = hdr["current_step"]
    return hdr
