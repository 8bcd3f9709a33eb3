// suhmo_level_int.h -- what the translation units of the level share (suhmo_level.hip: lifecycle, options, LevelData traffic, strip halos;
// suhmo_ops.hip: operator / residual / restriction / prolongation / vector kernels; suhmo_bcoef.hip: WFlx_level, AverageOperator, MGnewOp's coefficients)
#pragma once
#include "suhmo_hier.h"
#include <initializer_list>
#define BLK2D dim3(64, 4)
static inline dim3 grid2d(int nx, int ny) { return dim3((nx + 63) / 64, (ny + 3) / 4); }
static inline dim3 grid_m(const suhmo_multi &m, int ex = 0, int ey = 0) { return dim3((m.maxnx + ex + 63) / 64, (m.maxny + ey + 3) / 4, m.nbox); }
static inline void phi_changed(suhmo_level *L, int depth) { L->d[depth].phi_fresh = 0; }   // see suhmo_ensure_phi_halo
static inline bool is_xface(int f) { return f == SUHMO_F_BX || f == SUHMO_F_QWX || f == SUHMO_F_DCX; }
static inline bool is_yface(int f) { return f == SUHMO_F_BY || f == SUHMO_F_QWY || f == SUHMO_F_DCY; }
static inline bool is_face(int f) { return is_xface(f) || is_yface(f); }
static inline bool on_strip(const suhmo_level *L) { const DV &v = L->d[0].v; return v.rk[0] || v.rk[1]; }
#define CHECK_DF(L, depth, field) ARG(L); ARG(depth >= 0 && depth < L->ndepth); ARG(field >= 0 && field < SUHMO_F_COUNT); \
    if (L->stub) { suhmo_set_error("this box of a partitioned AMR level is held by another rank (suhmo_hier_box_owner)"); return -7; }
int suhmo_exchange_fields(suhmo_level *L, int depth, std::initializer_list<int> fields, hipStream_t st);   // LevelData::exchange across rank boundaries (suhmo_ops.hip)
