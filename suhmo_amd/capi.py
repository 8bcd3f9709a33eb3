"""ctypes binding of libsuhmo_hip.so (the C-ABI declared in include/suhmo_hip.h).

There is NO fallback: if the HIP library is missing or no GPU is visible, calls raise.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("SUHMO_LIB") or os.path.join(CSRC, "libsuhmo_hip.so")     # SUHMO_LIB: another build of the same library (A/B runs)
_LIB = None


class SuhmoError(RuntimeError):
    pass


class Phys(C.Structure):
    _fields_ = [("A", C.c_double), ("omega", C.c_double), ("nu", C.c_double),
                ("cutOffbr", C.c_double), ("maxOffbr", C.c_double),
                ("rho_w_g", C.c_double), ("grav", C.c_double),
                ("cutOffB", C.c_int), ("use_NL", C.c_int), ("use_mask_gradients", C.c_int)]


class BC(C.Structure):
    _fields_ = [("type", (C.c_int * 2) * 2), ("value", (C.c_double * 2) * 2), ("periodic", C.c_int * 2)]


class SolverParams(C.Structure):
    _fields_ = [("num_smooth", C.c_int), ("num_bottom", C.c_int), ("max_iter", C.c_int),
                ("iter_min", C.c_int), ("imin", C.c_int), ("eps", C.c_double), ("hang", C.c_double),
                ("norm_thresh", C.c_double), ("bcoeff_otf", C.c_int), ("max_depth", C.c_int)]


class ModelParams(C.Structure):
    _fields_ = [("rho_i", C.c_double), ("rho_w", C.c_double), ("gravity", C.c_double), ("G", C.c_double),
                ("L", C.c_double), ("ct", C.c_double), ("cw", C.c_double), ("ub0", C.c_double), ("ub1", C.c_double),
                ("br", C.c_double), ("lr", C.c_double), ("diffFactor", C.c_double),
                ("distributed_input", C.c_double), ("eps_picard", C.c_double),
                ("basal_friction", C.c_int), ("use_mask_rhs_b", C.c_int), ("use_moulin_source", C.c_int),
                ("ramp", C.c_double), ("use_impl_diff", C.c_int),
                ("head_melt_off", C.c_int), ("freeze_icefree_gap", C.c_int)]


class LevelDesc(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("j0", C.c_int), ("ny_global", C.c_int),
                ("dx", C.c_double), ("dy", C.c_double), ("nbox", C.c_int), ("boxes", C.POINTER(C.c_int)),
                ("max_box", C.c_int), ("alpha", C.c_double), ("beta", C.c_double),
                ("bc", BC), ("phys", Phys), ("device", C.c_int), ("halo_rows", C.c_int),
                ("i0", C.c_int), ("nx_global", C.c_int), ("patch_j0", C.c_int), ("patch_ny", C.c_int)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double))
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)      # values, n, op (0 MAX, 1 SUM)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p)

# every symbol include/suhmo_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "suhmo_last_error", "suhmo_device_count", "suhmo_level_create", "suhmo_level_destroy",
    "suhmo_level_num_depths", "suhmo_level_synchronize", "suhmo_level_put_box", "suhmo_level_get_box",
    "suhmo_level_set_field", "suhmo_level_get_field", "suhmo_level_field_view", "suhmo_level_gsrb",
    "suhmo_level_apply_op", "suhmo_level_residual", "suhmo_level_restrict_residual",
    "suhmo_level_restrict_r", "suhmo_level_prolong_increment", "suhmo_level_prolong_bilinear",
    "suhmo_level_update_operator", "suhmo_level_average_operator", "suhmo_level_build_mg_coefficients",
    "suhmo_level_nonlinear", "suhmo_level_compute_lambda", "suhmo_level_fill_ghosts",
    "suhmo_level_divergence", "suhmo_level_get_flux", "suhmo_level_norm", "suhmo_level_axby",
    "suhmo_level_set_value", "suhmo_level_vcycle", "suhmo_level_solve", "suhmo_level_pack_rows",
    "suhmo_level_unpack_rows", "suhmo_level_set_hooks", "suhmo_level_exchange", "suhmo_level_halo_info", "suhmo_level_profile_reset",
    "suhmo_level_profile_enable", "suhmo_level_profile_read", "suhmo_level_profile_read_restricting", "suhmo_level_timestep",
    "suhmo_rccl_load", "suhmo_rccl_unique_id", "suhmo_level_attach_rccl", "suhmo_level_detach_rccl",
    "suhmo_level_rccl_exchanges", "suhmo_level_rccl_comm_count", "suhmo_level_ipc_export", "suhmo_level_attach_ipc", "suhmo_level_ipc_exchanges", "suhmo_level_detach_ipc",
    "suhmo_amr2_cf_interp", "suhmo_amr2_average", "suhmo_amr2_fine_update_operator", "suhmo_amr2_residual",
    "suhmo_amr2_vcycle", "suhmo_amr2_solve", "suhmo_level_moulin_source", "suhmo_amr2_prolong2", "suhmo_amr2_set_covered",
    "suhmo_amr_residual", "suhmo_amr_vcycle", "suhmo_amr_solve", "suhmo_level_postproc_table", "suhmo_level_postproc_partial", "suhmo_postproc_finish", "suhmo_postproc_temporal", "suhmo_level_postproc_temporal",
    "suhmo_level_set_alpha_beta", "suhmo_level_set_bc", "suhmo_amr2_reflux", "suhmo_amr2_pwl_fill", "suhmo_amr_timestep", "suhmo_level_time_varying_recharge", "suhmo_amr_moulin_source", "suhmo_amr2_prolong_pc", "suhmo_amr2_finer_operator_changed",
    "suhmo_hier_create", "suhmo_hier_destroy", "suhmo_hier_num_levels", "suhmo_hier_num_boxes", "suhmo_hier_box", "suhmo_hier_box_owner", "suhmo_hier_exchange",
    "suhmo_hier_cf_interp", "suhmo_hier_pwl_fill", "suhmo_hier_average", "suhmo_hier_gsrb", "suhmo_hier_update_operator",
    "suhmo_hier_residual", "suhmo_hier_vcycle", "suhmo_hier_solve", "suhmo_hier_timestep", "suhmo_hier_moulin_source",
    "suhmo_hier_set_allgather", "suhmo_hier_attach_rccl", "suhmo_hier_gathers", "suhmo_level_set_option", "suhmo_level_get_option", "suhmo_timers_enable", "suhmo_timers_reset", "suhmo_timers_report",
    "suhmo_level_set_reduce_hook", "suhmo_level_dot", "suhmo_level_set_allgather", "suhmo_level_agglomerated_depth",
    "suhmo_hier_create_opts", "suhmo_hier_set_option", "suhmo_hier_get_option",
]


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU).  With SUHMO_LIB set the library in use is
    somebody else's build: nothing is compiled (the in-tree file would not be the one lib() loads)."""
    if os.environ.get("SUHMO_LIB"):
        if not os.path.exists(LIB_PATH):
            raise SuhmoError("SUHMO_LIB = %s does not exist" % LIB_PATH)
        return LIB_PATH
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "suhmo_hip.h"))
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", CSRC, "-B", "libsuhmo_hip.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise SuhmoError("libsuhmo_hip.so not built (%s); run __graft_entry__.build(). "
                         "There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    dp, vp, ci = C.POINTER(C.c_double), C.c_void_p, C.c_int
    L.suhmo_last_error.restype = C.c_char_p
    L.suhmo_level_create.argtypes = [C.POINTER(vp), C.POINTER(LevelDesc)]
    L.suhmo_level_destroy.argtypes = [vp]
    L.suhmo_level_num_depths.argtypes = [vp]
    L.suhmo_level_synchronize.argtypes = [vp, vp]
    L.suhmo_level_put_box.argtypes = [vp, ci, ci, ci, dp, ci, ci, ci, ci, ci, vp]
    L.suhmo_level_get_box.argtypes = [vp, ci, ci, ci, dp, ci, ci, ci, ci, vp]
    L.suhmo_level_set_field.argtypes = [vp, ci, ci, vp, ci, ci, vp]
    L.suhmo_level_get_field.argtypes = [vp, ci, ci, vp, ci, ci, vp]
    L.suhmo_level_field_view.argtypes = [vp, ci, ci, C.POINTER(vp), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.suhmo_level_gsrb.argtypes = [vp, ci, ci, vp]
    L.suhmo_level_apply_op.argtypes = [vp, ci, ci, vp]
    for n in ("residual", "restrict_residual", "restrict_r", "prolong_increment", "prolong_bilinear",
              "update_operator", "average_operator", "nonlinear", "compute_lambda"):
        getattr(L, "suhmo_level_" + n).argtypes = [vp, ci, vp]
    L.suhmo_level_build_mg_coefficients.argtypes = [vp, vp]
    L.suhmo_level_fill_ghosts.argtypes = [vp, ci, ci, ci, vp]
    L.suhmo_level_divergence.argtypes = [vp, ci, ci, vp]
    L.suhmo_level_get_flux.argtypes = [vp, ci, ci, ci, dp, vp]
    L.suhmo_level_norm.argtypes = [vp, ci, ci, ci, dp, vp]
    L.suhmo_level_axby.argtypes = [vp, ci, ci, ci, ci, C.c_double, C.c_double, vp]
    L.suhmo_level_set_value.argtypes = [vp, ci, ci, C.c_double, vp]
    L.suhmo_level_vcycle.argtypes = [vp, C.POINTER(SolverParams), vp]
    L.suhmo_level_solve.argtypes = [vp, C.POINTER(SolverParams), C.POINTER(ci), dp, vp]
    L.suhmo_level_pack_rows.argtypes = [vp, ci, ci, ci, ci, vp, vp]
    L.suhmo_level_unpack_rows.argtypes = [vp, ci, ci, ci, ci, vp, vp]
    L.suhmo_level_set_hooks.argtypes = [vp, EXCHANGE_FN, ALLREDUCE_FN, vp]
    L.suhmo_level_set_reduce_hook.argtypes = [vp, REDUCE_FN]
    L.suhmo_level_set_allgather.argtypes = [vp, ALLGATHER_FN, vp]
    L.suhmo_level_agglomerated_depth.argtypes = [vp]
    L.suhmo_level_dot.argtypes = [vp, ci, ci, ci, dp, vp]
    L.suhmo_level_exchange.argtypes = [vp, ci, ci, vp]
    L.suhmo_level_halo_info.argtypes = [vp, ci] + [C.POINTER(ci)] * 5
    L.suhmo_level_timestep.argtypes = [vp, C.POINTER(ModelParams), C.c_double, ci, C.POINTER(ci), C.POINTER(ci), vp]
    L.suhmo_rccl_load.argtypes = [C.c_char_p]
    L.suhmo_rccl_unique_id.argtypes = [vp]
    L.suhmo_level_attach_rccl.argtypes = [vp, vp, ci, ci, ci, vp]
    L.suhmo_level_detach_rccl.argtypes = [vp]
    L.suhmo_level_rccl_exchanges.argtypes = [vp]
    L.suhmo_level_rccl_exchanges.restype = C.c_long
    L.suhmo_level_ipc_export.argtypes = [vp, vp]
    L.suhmo_level_attach_ipc.argtypes = [vp, ci, ci, ci, vp, vp]
    L.suhmo_level_ipc_exchanges.argtypes = [vp]
    L.suhmo_level_detach_ipc.argtypes = [vp]
    L.suhmo_level_ipc_exchanges.restype = C.c_long
    L.suhmo_level_moulin_source.argtypes = [vp, ci, dp, dp, dp, C.c_double, dp, vp]
    L.suhmo_level_postproc_table.argtypes = [vp, C.POINTER(ModelParams), dp, vp]
    L.suhmo_level_postproc_partial.argtypes = [vp, C.POINTER(ModelParams), dp, vp]
    L.suhmo_postproc_finish.argtypes = [dp, C.c_int, C.c_double, dp]
    L.suhmo_postproc_temporal.argtypes = [dp, C.c_int, C.c_double, dp]
    L.suhmo_level_postproc_temporal.argtypes = [vp, C.POINTER(ModelParams), dp, vp]
    L.suhmo_level_set_alpha_beta.argtypes = [vp, C.c_double, C.c_double]
    L.suhmo_level_set_bc.argtypes = [vp, C.POINTER(BC)]
    L.suhmo_amr2_reflux.argtypes = [vp, vp, C.c_int, vp]
    L.suhmo_amr2_pwl_fill.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.suhmo_level_time_varying_recharge.argtypes = [vp, C.c_double, C.c_double, vp]
    L.suhmo_amr_moulin_source.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int), C.c_int, dp, dp, dp, C.c_double, dp, vp]
    L.suhmo_amr_timestep.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(ModelParams), C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), vp]
    L.suhmo_amr2_prolong_pc.argtypes = [vp, vp, C.c_int, vp]
    L.suhmo_amr2_finer_operator_changed.argtypes = [vp, vp, vp]
    L.suhmo_amr_residual.argtypes = [C.POINTER(vp), ci, dp, vp]
    L.suhmo_amr_vcycle.argtypes = [C.POINTER(vp), ci, C.POINTER(SolverParams), vp]
    L.suhmo_amr_solve.argtypes = [C.POINTER(vp), ci, C.POINTER(SolverParams), C.POINTER(ci), dp, vp]
    L.suhmo_amr2_prolong2.argtypes = [vp, vp, ci, vp]
    L.suhmo_amr2_set_covered.argtypes = [vp, vp, ci, C.c_double, vp]
    L.suhmo_amr2_cf_interp.argtypes = [vp, vp, ci, ci, vp]
    L.suhmo_amr2_average.argtypes = [vp, vp, ci, ci, vp]
    L.suhmo_amr2_fine_update_operator.argtypes = [vp, vp, vp]
    L.suhmo_amr2_residual.argtypes = [vp, vp, dp, vp]
    L.suhmo_amr2_vcycle.argtypes = [vp, vp, C.POINTER(SolverParams), vp]
    L.suhmo_amr2_solve.argtypes = [vp, vp, C.POINTER(SolverParams), C.POINTER(ci), dp, vp]
    ip = C.POINTER(ci)
    L.suhmo_hier_create.argtypes = [C.POINTER(vp), C.POINTER(LevelDesc), ci, ip, ip]
    L.suhmo_hier_create_opts.argtypes = [C.POINTER(vp), C.POINTER(LevelDesc), ci, ip, ip, C.c_char_p]
    L.suhmo_hier_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    L.suhmo_hier_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_long)]
    L.suhmo_hier_destroy.argtypes = [vp]
    L.suhmo_hier_num_levels.argtypes = [vp]
    L.suhmo_hier_num_boxes.argtypes = [vp, ci]
    L.suhmo_hier_box.argtypes = [vp, ci, ci]
    L.suhmo_hier_box.restype = vp
    L.suhmo_hier_box_owner.argtypes = [vp, ci, ci, ip]
    L.suhmo_hier_exchange.argtypes = [vp, ci, ci, ci, vp]
    L.suhmo_hier_cf_interp.argtypes = [vp, ci, ci, ci, vp]
    L.suhmo_hier_pwl_fill.argtypes = [vp, ci, ci, ci, vp]
    L.suhmo_hier_average.argtypes = [vp, ci, ci, ci, vp]
    L.suhmo_hier_gsrb.argtypes = [vp, ci, ci, vp]
    L.suhmo_hier_update_operator.argtypes = [vp, ci, vp]
    L.suhmo_hier_residual.argtypes = [vp, dp, vp]
    L.suhmo_hier_vcycle.argtypes = [vp, C.POINTER(SolverParams), vp]
    L.suhmo_hier_solve.argtypes = [vp, C.POINTER(SolverParams), ip, dp, vp]
    L.suhmo_hier_timestep.argtypes = [vp, C.POINTER(ModelParams), C.c_double, ci, ip, ip, vp]
    L.suhmo_hier_moulin_source.argtypes = [vp, ci, dp, dp, dp, C.c_double, dp, vp]
    L.suhmo_hier_set_allgather.argtypes = [vp, ALLGATHER_FN, vp]
    L.suhmo_hier_attach_rccl.argtypes = [vp]
    L.suhmo_hier_gathers.argtypes = [vp]
    L.suhmo_hier_gathers.restype = C.c_long
    L.suhmo_timers_enable.argtypes = [ci]
    L.suhmo_timers_report.argtypes = [C.c_char_p, C.c_long]
    L.suhmo_timers_report.restype = C.c_long
    L.suhmo_level_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    L.suhmo_level_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_long)]
    L.suhmo_level_profile_reset.argtypes = [vp]
    L.suhmo_level_profile_enable.argtypes = [vp, ci]
    L.suhmo_level_profile_read.argtypes = [vp, vp, dp, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.suhmo_level_profile_read_restricting.argtypes = [vp, vp, dp, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    _LIB = L
    return L


def check(rc):
    if rc != 0:
        raise SuhmoError("libsuhmo_hip: rc=%d: %s" % (rc, lib().suhmo_last_error().decode()))


def timers_report():
    """CH_TIMER_REPORT: the named timers as text (enable with lib().suhmo_timers_enable(1 | 2) or SUHMO_TIMERS)"""
    n = lib().suhmo_timers_report(None, 0)
    buf = C.create_string_buffer(n)
    lib().suhmo_timers_report(buf, n)
    return buf.value.decode()
