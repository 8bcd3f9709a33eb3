/*
 * suhmo_chk.h -- C-ABI of the checkpoint reader / writer (libsuhmo_chk.so, host side, HDF5 C library).
 *
 * Replaces AmrHydro::writeCheckpointFile / readCheckpointFile (src/AmrHydro.cpp:5670-5842, 5845-6208): a Chombo HDF5
 * checkpoint = root attributes (header: max_level, finest_level, current_step, time, dt, num_comps, cfl, is_periodic_<d>,
 * component_<nnnn> names, :5710-5795), and per level a group level_<l> with attributes ref_ratio, dx, dy, prob_domain
 * (:5803-5819), the box list (write(handle, grids) :5824) and eleven LevelData<FArrayBox> written WITH their ghost cells
 * (:5826-5836): headData, gapHeightData, overburdenPressData, velMagData, bedelevationData, ReData, iceHeightData,
 * bumpHeightData, bumpSpacingData, meltRateData, iceMaskData.
 *
 * The container layout is Chombo's (lib/src/BoxTools/CH_HDF5.cpp of Chombo 3.2: "<name>:datatype=0" holds every box's
 * ghosted fab one after the other in Fortran order, "<name>:offsets=0" the start of each, group "<name>_attributes" the
 * component count and ghost vectors; boxes are the compound {lo_i, lo_j, hi_i, hi_j}).  The fork is not vendored and the
 * reference ships no checkpoint file: UNPINNED; the round trip and the bit-for-bit restart are what the tests hold it to.
 *
 * Plain C, host pointers, int return codes (0 ok), text via suhmo_chk_last_error().
 */
#ifndef SUHMO_CHK_H
#define SUHMO_CHK_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct suhmo_chk suhmo_chk_t;

typedef struct suhmo_chk_header {
    int max_level, finest_level, current_step;
    double time, dt, cfl;
    int is_periodic[2];
} suhmo_chk_header_t;

#define SUHMO_CHK_NFIELDS 11
/* dataset names in the order of src/AmrHydro.cpp:5826-5836 */
extern const char *const suhmo_chk_field_names[SUHMO_CHK_NFIELDS];

const char *suhmo_chk_last_error(void);

/* ---- writing */
int suhmo_chk_create(suhmo_chk_t **out, const char *path, const suhmo_chk_header_t *hdr);
/* level group: attributes + box list (boxes: nbox x {lo0, lo1, hi0, hi1}); ref_ratio <= 0: not written (finest allowed level) */
int suhmo_chk_write_level(suhmo_chk_t *h, int level, double dx, double dy, int ref_ratio, const int domain[4], int nbox, const int *boxes);
/* one LevelData<FArrayBox> of that level: fabs[k] = box k grown by `ghost`, Fortran order (i fastest), one component */
int suhmo_chk_write_field(suhmo_chk_t *h, int level, const char *name, int ghost, const double *const *fabs);
int suhmo_chk_close(suhmo_chk_t *h);

/* ---- reading */
int suhmo_chk_open(suhmo_chk_t **out, const char *path, suhmo_chk_header_t *hdr);
/* boxes == NULL: only the count */
int suhmo_chk_read_level(suhmo_chk_t *h, int level, double *dx, double *dy, int *ref_ratio, int domain[4], int *nbox, int *boxes, int max_boxes);
/* fabs[k] receives box k grown by the ghost width the data were written with (*ghost) */
int suhmo_chk_field_ghost(suhmo_chk_t *h, int level, const char *name, int *ghost);
int suhmo_chk_read_field(suhmo_chk_t *h, int level, const char *name, double *const *fabs);

#ifdef __cplusplus
}
#endif
#endif
