// suhmo_chk.cpp -- Chombo-HDF5 checkpoint files of the hydrology state (include/suhmo_chk.h).  Host code, HDF5 C library.
#include "../../include/suhmo_chk.h"
#include <hdf5.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern "C" const char *const suhmo_chk_field_names[SUHMO_CHK_NFIELDS] = {
    "headData", "gapHeightData", "overburdenPressData", "velMagData", "bedelevationData", "ReData",
    "iceHeightData", "bumpHeightData", "bumpSpacingData", "meltRateData", "iceMaskData"};
// component names of the root header, src/AmrHydro.cpp:5736-5795
static const char *const k_comp_names[SUHMO_CHK_NFIELDS] = {
    "head", "gapHeight", "overburdenPress", "magVel", "bedelevation", "Re", "iceHeight", "bumpHeight", "bumpSpacing", "meltRate", "iceMask"};

static thread_local char g_err[512] = "";
static int fail(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return -1;
}
extern "C" const char *suhmo_chk_last_error(void) { return g_err; }

struct LevelInfo { int nbox = 0; std::vector<int> boxes; };
struct suhmo_chk {
    hid_t file = -1;
    bool writing = false;
    hid_t box_t = -1, iv_t = -1;
    std::vector<LevelInfo> lev;
};

namespace {
struct Box2 { int lo_i, lo_j, hi_i, hi_j; };
struct IV2 { int intvecti, intvectj; };
hid_t make_box_type()
{
    hid_t t = H5Tcreate(H5T_COMPOUND, sizeof(Box2));
    H5Tinsert(t, "lo_i", HOFFSET(Box2, lo_i), H5T_NATIVE_INT); H5Tinsert(t, "lo_j", HOFFSET(Box2, lo_j), H5T_NATIVE_INT);
    H5Tinsert(t, "hi_i", HOFFSET(Box2, hi_i), H5T_NATIVE_INT); H5Tinsert(t, "hi_j", HOFFSET(Box2, hi_j), H5T_NATIVE_INT);
    return t;
}
hid_t make_iv_type()
{
    hid_t t = H5Tcreate(H5T_COMPOUND, sizeof(IV2));
    H5Tinsert(t, "intvecti", HOFFSET(IV2, intvecti), H5T_NATIVE_INT); H5Tinsert(t, "intvectj", HOFFSET(IV2, intvectj), H5T_NATIVE_INT);
    return t;
}
int put_attr(hid_t loc, const char *name, hid_t type, const void *val)
{
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(loc, name, type, sp, H5P_DEFAULT, H5P_DEFAULT);
    herr_t e = a >= 0 ? H5Awrite(a, type, val) : -1;
    if (a >= 0) H5Aclose(a);
    H5Sclose(sp);
    return e < 0 ? fail("cannot write attribute %s", name) : 0;
}
int put_str(hid_t loc, const char *name, const char *val)
{
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, strlen(val) > 0 ? strlen(val) : 1);
    int rc = put_attr(loc, name, t, val);
    H5Tclose(t);
    return rc;
}
int get_attr(hid_t loc, const char *name, hid_t type, void *val)
{
    if (H5Aexists(loc, name) <= 0) return fail("attribute %s missing", name);
    hid_t a = H5Aopen(loc, name, H5P_DEFAULT);
    herr_t e = a >= 0 ? H5Aread(a, type, val) : -1;
    if (a >= 0) H5Aclose(a);
    return e < 0 ? fail("cannot read attribute %s", name) : 0;
}
std::string level_name(int l) { char b[32]; snprintf(b, sizeof(b), "level_%d", l); return b; }
long box_pts(const int *b, int g) { return (long)(b[2] - b[0] + 1 + 2 * g) * (long)(b[3] - b[1] + 1 + 2 * g); }
}  // namespace

extern "C" int suhmo_chk_create(suhmo_chk_t **out, const char *path, const suhmo_chk_header_t *hdr)
{
    if (!out || !path || !hdr) return fail("bad argument");
    H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
    suhmo_chk *h = new suhmo_chk();
    h->writing = true;
    h->file = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (h->file < 0) { delete h; return fail("cannot create %s", path); }
    h->box_t = make_box_type(); h->iv_t = make_iv_type();
    // HDF5Handle(CREATE): the group Chombo_global with SpaceDim and testReal
    hid_t g = H5Gcreate2(h->file, "Chombo_global", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    int sd = 2; double tr = 0.0;
    int rc = put_attr(g, "SpaceDim", H5T_NATIVE_INT, &sd) | put_attr(g, "testReal", H5T_NATIVE_DOUBLE, &tr);
    H5Gclose(g);
    hid_t root = H5Gopen2(h->file, "/", H5P_DEFAULT);
    int ncomp = 10;                                         // src/AmrHydro.cpp:5716 ("H/B/Pice/Zb/Re/Hice/BH/BL/MR/MS")
    rc |= put_attr(root, "max_level", H5T_NATIVE_INT, &hdr->max_level) | put_attr(root, "finest_level", H5T_NATIVE_INT, &hdr->finest_level)
        | put_attr(root, "current_step", H5T_NATIVE_INT, &hdr->current_step) | put_attr(root, "time", H5T_NATIVE_DOUBLE, &hdr->time)
        | put_attr(root, "dt", H5T_NATIVE_DOUBLE, &hdr->dt) | put_attr(root, "num_comps", H5T_NATIVE_INT, &ncomp)
        | put_attr(root, "cfl", H5T_NATIVE_DOUBLE, &hdr->cfl) | put_attr(root, "is_periodic_0", H5T_NATIVE_INT, &hdr->is_periodic[0])
        | put_attr(root, "is_periodic_1", H5T_NATIVE_INT, &hdr->is_periodic[1]);
    for (int c = 0; c < SUHMO_CHK_NFIELDS && !rc; c++) {
        char key[32]; snprintf(key, sizeof(key), "component_%04d", c);
        rc |= put_str(root, key, k_comp_names[c]);
    }
    H5Gclose(root);
    if (rc) { suhmo_chk_close(h); return -1; }
    h->lev.resize(hdr->max_level + 1);
    *out = h;
    return 0;
}

extern "C" int suhmo_chk_write_level(suhmo_chk_t *h, int level, double dx, double dy, int ref_ratio, const int domain[4], int nbox, const int *boxes)
{
    if (!h || !h->writing || level < 0 || level >= (int)h->lev.size() || nbox < 0 || (nbox && !boxes)) return fail("bad argument");
    hid_t g = H5Gcreate2(h->file, level_name(level).c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (g < 0) return fail("cannot create group %s", level_name(level).c_str());
    int rc = 0;
    if (ref_ratio > 0) rc |= put_attr(g, "ref_ratio", H5T_NATIVE_INT, &ref_ratio);
    rc |= put_attr(g, "dx", H5T_NATIVE_DOUBLE, &dx) | put_attr(g, "dy", H5T_NATIVE_DOUBLE, &dy);
    Box2 dom{domain[0], domain[1], domain[2], domain[3]};
    rc |= put_attr(g, "prob_domain", h->box_t, &dom);
    if (nbox > 0 && !rc) {
        std::vector<Box2> bx(nbox);
        for (int k = 0; k < nbox; k++) bx[k] = Box2{boxes[4 * k], boxes[4 * k + 1], boxes[4 * k + 2], boxes[4 * k + 3]};
        hsize_t n = (hsize_t)nbox;
        hid_t sp = H5Screate_simple(1, &n, nullptr);
        hid_t d = H5Dcreate2(g, "boxes", h->box_t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        if (d < 0 || H5Dwrite(d, h->box_t, H5S_ALL, H5S_ALL, H5P_DEFAULT, bx.data()) < 0) rc = fail("cannot write the boxes of level %d", level);
        if (d >= 0) H5Dclose(d);
        std::vector<int> procs(nbox, 0);
        hid_t dp = H5Dcreate2(g, "Processors", H5T_NATIVE_INT, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        if (dp >= 0) { H5Dwrite(dp, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, procs.data()); H5Dclose(dp); }
        H5Sclose(sp);
    }
    H5Gclose(g);
    h->lev[level].nbox = nbox;
    h->lev[level].boxes.assign(boxes, boxes + 4 * (size_t)nbox);
    return rc;
}

extern "C" int suhmo_chk_write_field(suhmo_chk_t *h, int level, const char *name, int ghost, const double *const *fabs)
{
    if (!h || !h->writing || level < 0 || level >= (int)h->lev.size() || !name || ghost < 0 || !fabs) return fail("bad argument");
    const LevelInfo &L = h->lev[level];
    if (L.nbox <= 0) return fail("level %d has no boxes (suhmo_chk_write_level first)", level);
    hid_t g = H5Gopen2(h->file, level_name(level).c_str(), H5P_DEFAULT);
    if (g < 0) return fail("level %d not written", level);
    std::vector<long long> off(L.nbox + 1, 0);
    for (int k = 0; k < L.nbox; k++) off[k + 1] = off[k] + box_pts(&L.boxes[4 * k], ghost);
    std::vector<double> flat((size_t)off[L.nbox]);
    for (int k = 0; k < L.nbox; k++) memcpy(&flat[(size_t)off[k]], fabs[k], sizeof(double) * (size_t)(off[k + 1] - off[k]));
    int rc = 0;
    std::string base(name);
    {
        hsize_t n = (hsize_t)flat.size();
        hid_t sp = H5Screate_simple(1, &n, nullptr);
        hid_t d = H5Dcreate2(g, (base + ":datatype=0").c_str(), H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        if (d < 0 || H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, flat.data()) < 0) rc = fail("cannot write %s", name);
        if (d >= 0) H5Dclose(d);
        H5Sclose(sp);
    }
    {
        hsize_t n = (hsize_t)off.size();
        hid_t sp = H5Screate_simple(1, &n, nullptr);
        hid_t d = H5Dcreate2(g, (base + ":offsets=0").c_str(), H5T_NATIVE_LLONG, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        if (d < 0 || H5Dwrite(d, H5T_NATIVE_LLONG, H5S_ALL, H5S_ALL, H5P_DEFAULT, off.data()) < 0) rc = fail("cannot write the offsets of %s", name);
        if (d >= 0) H5Dclose(d);
        H5Sclose(sp);
    }
    {
        hid_t a = H5Gcreate2(g, (base + "_attributes").c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        int comps = 1; IV2 gv{ghost, ghost};
        rc |= put_attr(a, "comps", H5T_NATIVE_INT, &comps) | put_attr(a, "ghost", h->iv_t, &gv) | put_attr(a, "outputGhost", h->iv_t, &gv)
            | put_str(a, "objectType", "FArrayBox");
        H5Gclose(a);
    }
    H5Gclose(g);
    return rc;
}

extern "C" int suhmo_chk_close(suhmo_chk_t *h)
{
    if (!h) return 0;
    if (h->box_t >= 0) H5Tclose(h->box_t);
    if (h->iv_t >= 0) H5Tclose(h->iv_t);
    if (h->file >= 0) H5Fclose(h->file);
    delete h;
    return 0;
}

extern "C" int suhmo_chk_open(suhmo_chk_t **out, const char *path, suhmo_chk_header_t *hdr)
{
    if (!out || !path || !hdr) return fail("bad argument");
    H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
    suhmo_chk *h = new suhmo_chk();
    h->file = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (h->file < 0) { delete h; return fail("cannot open %s", path); }
    h->box_t = make_box_type(); h->iv_t = make_iv_type();
    hid_t root = H5Gopen2(h->file, "/", H5P_DEFAULT);
    int rc = get_attr(root, "max_level", H5T_NATIVE_INT, &hdr->max_level) | get_attr(root, "finest_level", H5T_NATIVE_INT, &hdr->finest_level)
           | get_attr(root, "current_step", H5T_NATIVE_INT, &hdr->current_step) | get_attr(root, "time", H5T_NATIVE_DOUBLE, &hdr->time)
           | get_attr(root, "dt", H5T_NATIVE_DOUBLE, &hdr->dt) | get_attr(root, "cfl", H5T_NATIVE_DOUBLE, &hdr->cfl)
           | get_attr(root, "is_periodic_0", H5T_NATIVE_INT, &hdr->is_periodic[0]) | get_attr(root, "is_periodic_1", H5T_NATIVE_INT, &hdr->is_periodic[1]);
    H5Gclose(root);
    if (rc) { suhmo_chk_close(h); return -1; }
    h->lev.resize(hdr->max_level + 1);
    *out = h;
    return 0;
}

extern "C" int suhmo_chk_read_level(suhmo_chk_t *h, int level, double *dx, double *dy, int *ref_ratio, int domain[4], int *nbox, int *boxes, int max_boxes)
{
    if (!h || h->writing || level < 0 || level >= (int)h->lev.size() || !nbox) return fail("bad argument");
    hid_t g = H5Gopen2(h->file, level_name(level).c_str(), H5P_DEFAULT);
    if (g < 0) return fail("checkpoint file does not contain %s", level_name(level).c_str());
    int rc = 0;
    if (dx) rc |= get_attr(g, "dx", H5T_NATIVE_DOUBLE, dx);
    if (dy) rc |= get_attr(g, "dy", H5T_NATIVE_DOUBLE, dy);
    if (ref_ratio) { *ref_ratio = 0; if (H5Aexists(g, "ref_ratio") > 0) rc |= get_attr(g, "ref_ratio", H5T_NATIVE_INT, ref_ratio); }
    if (domain) { Box2 d; rc |= get_attr(g, "prob_domain", h->box_t, &d); domain[0] = d.lo_i; domain[1] = d.lo_j; domain[2] = d.hi_i; domain[3] = d.hi_j; }
    *nbox = 0;
    if (H5Lexists(g, "boxes", H5P_DEFAULT) > 0) {
        hid_t d = H5Dopen2(g, "boxes", H5P_DEFAULT);
        hid_t sp = H5Dget_space(d);
        hsize_t n = 0;
        H5Sget_simple_extent_dims(sp, &n, nullptr);
        *nbox = (int)n;
        std::vector<Box2> bx(n);
        if (n && H5Dread(d, h->box_t, H5S_ALL, H5S_ALL, H5P_DEFAULT, bx.data()) < 0) rc = fail("cannot read the boxes of level %d", level);
        H5Sclose(sp); H5Dclose(d);
        h->lev[level].nbox = (int)n;
        h->lev[level].boxes.resize(4 * n);
        for (size_t k = 0; k < n; k++) { int *q = &h->lev[level].boxes[4 * k]; q[0] = bx[k].lo_i; q[1] = bx[k].lo_j; q[2] = bx[k].hi_i; q[3] = bx[k].hi_j; }
        if (boxes) {
            if ((int)n > max_boxes) rc = fail("level %d has %d boxes, room for %d", level, (int)n, max_boxes);
            else memcpy(boxes, h->lev[level].boxes.data(), sizeof(int) * 4 * n);
        }
    }
    H5Gclose(g);
    return rc;
}

extern "C" int suhmo_chk_field_ghost(suhmo_chk_t *h, int level, const char *name, int *ghost)
{
    if (!h || h->writing || level < 0 || level >= (int)h->lev.size() || !name || !ghost) return fail("bad argument");
    hid_t g = H5Gopen2(h->file, level_name(level).c_str(), H5P_DEFAULT);
    if (g < 0) return fail("checkpoint file does not contain %s", level_name(level).c_str());
    hid_t a = H5Gopen2(g, (std::string(name) + "_attributes").c_str(), H5P_DEFAULT);
    int rc = 0;
    if (a < 0) rc = fail("checkpoint file does not contain %s", name);
    else { IV2 gv{0, 0}; rc = get_attr(a, "outputGhost", h->iv_t, &gv); *ghost = gv.intvecti; H5Gclose(a); }
    H5Gclose(g);
    return rc;
}

extern "C" int suhmo_chk_read_field(suhmo_chk_t *h, int level, const char *name, double *const *fabs)
{
    if (!h || h->writing || level < 0 || level >= (int)h->lev.size() || !name || !fabs) return fail("bad argument");
    int ghost = 0;
    if (suhmo_chk_field_ghost(h, level, name, &ghost)) return -1;
    const LevelInfo &L = h->lev[level];
    if (L.nbox <= 0) return fail("read the level first (suhmo_chk_read_level)");
    hid_t g = H5Gopen2(h->file, level_name(level).c_str(), H5P_DEFAULT);
    std::string base(name);
    std::vector<long long> off(L.nbox + 1, 0);
    int rc = 0;
    // number of elements a dataset really holds (a truncated or foreign file must not overrun the buffers sized from the box list)
    auto extent = [](hid_t d) -> long long { hid_t sp = H5Dget_space(d); long long n = sp >= 0 ? (long long)H5Sget_simple_extent_npoints(sp) : -1; if (sp >= 0) H5Sclose(sp); return n; };
    {
        hid_t d = H5Dopen2(g, (base + ":offsets=0").c_str(), H5P_DEFAULT);
        if (d < 0) rc = fail("checkpoint file does not contain %s data", name);
        else if (extent(d) != (long long)L.nbox + 1) rc = fail("%s: the offsets hold %lld entries, the level has %d boxes", name, extent(d), L.nbox);
        else if (H5Dread(d, H5T_NATIVE_LLONG, H5S_ALL, H5S_ALL, H5P_DEFAULT, off.data()) < 0) rc = fail("checkpoint file does not contain %s data", name);
        if (d >= 0) H5Dclose(d);
    }
    if (!rc && off[0] != 0) rc = fail("%s: the offsets do not start at 0", name);
    for (int k = 0; k < L.nbox && !rc; k++)
        if (off[k + 1] - off[k] != box_pts(&L.boxes[4 * k], ghost)) rc = fail("%s: box %d does not have the size its ghost width implies", name, k);   // (also: monotone, non-negative)
    if (!rc) {
        std::vector<double> flat((size_t)off[L.nbox]);
        hid_t d = H5Dopen2(g, (base + ":datatype=0").c_str(), H5P_DEFAULT);
        if (d < 0) rc = fail("checkpoint file does not contain %s data", name);
        else if (extent(d) != off[L.nbox]) rc = fail("%s: the data set holds %lld values, the boxes need %lld", name, extent(d), off[L.nbox]);
        else if (H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, flat.data()) < 0) rc = fail("checkpoint file does not contain %s data", name);
        if (d >= 0) H5Dclose(d);
        for (int k = 0; k < L.nbox && !rc; k++) memcpy(fabs[k], &flat[(size_t)off[k]], sizeof(double) * (size_t)(off[k + 1] - off[k]));
    }
    H5Gclose(g);
    return rc;
}
