"""ctypes binding of ``lib/libnlbac_hip.so`` (the C ABI in ``include/nlbac_hip.h``).

There is no fallback: if the library is missing or an entry point is absent,
importing/using this module raises.  ``build()`` compiles it in-tree with
hipcc for gfx950 (cross-compiles without a GPU).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NLBAC_HIP_LIB") or os.path.join(_HERE, "lib", "libnlbac_hip.so")   # env: kernel experiments
CSRC = os.path.join(_HERE, "csrc")

MAX_LAYERS, MAX_NETS, MLP_TILE = 6, 8, 32
MLP_TILE_MIN = 16      # rows per workgroup of the finest MLP kernels: what per-tile partial buffers (nlbac_dy_head) are sized by
SC_SIZE, DOPRI_CTL = 128, 16

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)


class NlbacError(RuntimeError):
    pass


class Mlp(C.Structure):
    """``struct nlbac_mlp``"""
    _fields_ = [("n_layers", C.c_int), ("in_dim", C.c_int), ("hid", C.c_int), ("out_dim", C.c_int),
                ("params", C.c_void_p),
                ("w_off", C.c_int * MAX_LAYERS), ("b_off", C.c_int * MAX_LAYERS),
                ("packed", C.c_void_p),
                ("pf_off", C.c_int * MAX_LAYERS), ("pb_off", C.c_int * MAX_LAYERS),
                ("rr_fwd_off", C.c_int), ("rr_bwd_off", C.c_int), ("packed_floats", C.c_int), ("rr_kind", C.c_int)]


class MlpIO(C.Structure):
    """``struct nlbac_mlp_io``"""
    _fields_ = [("x0", C.c_void_p), ("x0_dim", C.c_int), ("x0_ld", C.c_int),
                ("x1", C.c_void_p), ("x1_dim", C.c_int), ("x1_ld", C.c_int),
                ("y", C.c_void_p), ("y_ld", C.c_int),
                ("acts", C.c_void_p), ("acts_ls", C.c_long),
                ("dy", C.c_void_p), ("dy_ld", C.c_int), ("dz_first", C.c_int),
                ("dz", C.c_void_p),
                ("dx", C.c_void_p), ("dx_ld", C.c_int), ("dx_first", C.c_int),
                ("grad", C.c_void_p),
                ("skinny_ws", C.c_void_p),
                ("masks", C.c_void_p)]


class AuglagArgs(C.Structure):
    """``struct nlbac_auglag_args``"""
    _fields_ = [("n_cbf", C.c_int), ("n_clf", C.c_int), ("batch_size", C.c_float), ("do_lambda_update", C.c_int),
                ("do_backup_lambda_update", C.c_int), ("ratio_mode", C.c_int), ("backup_mode", C.c_int),
                ("lam_lo", C.c_float), ("lam_hi", C.c_float)]


class ActorScalarArgs(C.Structure):
    """``struct nlbac_actor_scalar_args``"""
    _fields_ = [("target_entropy", C.c_float), ("log_alpha", C.c_void_p * 2), ("g_log_alpha", C.c_void_p * 2),
                ("sc", C.c_void_p)]


class GaussHead(C.Structure):
    """``struct nlbac_gauss_head``"""
    _fields_ = [("eps", C.c_void_p), ("scale", C.c_void_p), ("bias", C.c_void_p), ("n_u", C.c_int),
                ("action", C.c_void_p), ("action_ld", C.c_int), ("logp", C.c_void_p),
                ("cf_kind", C.c_int), ("cf_net", C.c_int), ("cf_nh", C.c_int),
                ("cf_ps", C.c_void_p), ("cf_ps_next", C.c_void_p), ("cf_V", C.c_void_p), ("cf_hazards", C.c_void_p),
                ("cf_r2", C.c_float), ("cf_dt", C.c_float), ("cf_gamma_b", C.c_float), ("cf_gamma_l", C.c_float),
                ("cf_matr", C.c_void_p), ("cf_bmatr", C.c_void_p), ("cf_partials", C.c_void_p), ("cf_tickets", C.c_void_p),
                ("cf_n_cbf", C.c_int), ("cf_n_clf", C.c_int), ("cf_batch_size", C.c_float),
                ("cf_do_lambda_update", C.c_int), ("cf_do_backup_lambda_update", C.c_int), ("cf_ratio_mode", C.c_int),
                ("cf_backup_mode", C.c_int), ("cf_lam_lo", C.c_float), ("cf_lam_hi", C.c_float), ("cf_sc", C.c_void_p),
                ("cf_defer", C.c_int), ("cf_tiles", C.c_void_p)]


class HeadSums(C.Structure):
    """``struct nlbac_head_sums``"""
    _fields_ = [("kind", C.c_int), ("n_nets", C.c_int), ("partials", C.c_void_p), ("n_tiles", C.c_void_p),
                ("mul", C.c_float), ("out", C.c_void_p), ("out_x", C.c_void_p), ("B_norm", C.c_int),
                ("actor", ActorScalarArgs), ("sc", C.c_void_p)]


class DyHead(C.Structure):
    """``struct nlbac_dy_head``"""
    _fields_ = [("kind", C.c_int), ("B_norm", C.c_int),
                ("heads", C.c_void_p), ("heads_ld", C.c_int), ("eps", C.c_void_p), ("scale", C.c_void_p), ("n_u", C.c_int),
                ("da", C.c_void_p * 3), ("da_ld", C.c_int * 3),
                ("alpha", C.c_void_p), ("dlogp_mul", C.c_float), ("dheads", C.c_void_p), ("dheads_ld", C.c_int),
                ("q1t", C.c_void_p), ("q2t", C.c_void_p), ("lt", C.c_void_p), ("nlogp", C.c_void_p), ("reward", C.c_void_p),
                ("constraint", C.c_void_p), ("mask", C.c_void_p), ("rcm_ld", C.c_int),
                ("q", C.c_void_p * 3), ("gamma", C.c_float), ("dq", C.c_void_p * 3), ("next_q", C.c_void_p),
                ("next_l", C.c_void_p),
                ("xt", C.c_void_p), ("xsig", C.c_void_p), ("xsig_ld", C.c_int), ("xq", C.c_void_p), ("dxq", C.c_void_p),
                ("out_x", C.c_void_p),
                ("qa", C.c_void_p), ("qb", C.c_void_p), ("logp", C.c_void_p), ("dqa", C.c_void_p), ("dqb", C.c_void_p),
                ("n_prob", C.c_int), ("actor", ActorScalarArgs),
                ("partials", C.c_void_p), ("ticket", C.c_void_p), ("mul", C.c_float), ("out", C.c_void_p),
                ("cb_kind", C.c_int), ("cb_nh", C.c_int), ("cb_ps_next", C.c_void_p), ("cb_matr", C.c_void_p),
                ("cb_bmatr", C.c_void_p), ("cb_hazards", C.c_void_p), ("cb_sc", C.c_void_p), ("cb_dt", C.c_float),
                ("cb_batch", C.c_float), ("cb_dps_next", C.c_void_p), ("cb_dV", C.c_void_p),
                ("sums_defer", C.c_int), ("sums_tiles", C.c_void_p), ("finish", HeadSums * 3),
                ("cb_defer", C.c_int), ("cb_partials", C.c_void_p), ("cb_tiles", C.c_void_p), ("cb_auglag", AuglagArgs),
                ("cb_stage", C.c_void_p)]


class RkChain(C.Structure):
    """``struct nlbac_rk_chain``"""
    _fields_ = [("ctl", C.c_void_p), ("slot_floats", C.c_long), ("norm_mode", C.c_int), ("n_slots", C.c_int),
                ("rtol", C.c_float), ("atol", C.c_float), ("t_end", C.c_double), ("partials", C.c_void_p),
                ("tickets", C.c_void_p), ("ctl_w", C.c_void_p), ("hslots", C.c_void_p), ("alog", C.c_void_p),
                ("alog_cap", C.c_int), ("ctl_host", C.c_void_p),
                ("interp_out", C.c_void_p), ("interp_kind", C.c_int), ("interp_l", C.c_float), ("interp_p", C.c_void_p),
                ("interp_bwd", C.c_int), ("interp_dout", C.c_void_p), ("interp_dp", C.c_void_p),
                ("interp_dp2", C.c_void_p), ("interp_x", C.c_void_p), ("ctl_seq", C.c_double),
                ("norm_defer", C.c_int), ("norm_pre", C.c_int), ("partials_pre", C.c_void_p)]


class InMap(C.Structure):
    """``struct nlbac_in_map``"""
    _fields_ = [("kind", C.c_int), ("obs", C.c_void_p), ("obs_ld", C.c_int), ("l", C.c_float), ("ps", C.c_void_p)]


class OutMap(C.Structure):
    """``struct nlbac_out_map``"""
    _fields_ = [("kind", C.c_int), ("l", C.c_float), ("p", C.c_void_p), ("dp", C.c_void_p), ("dp2", C.c_void_p),
                ("x", C.c_void_p)]


_P, _I, _F, _D, _L = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_long

# name -> argtypes (return type is always int unless listed in _RESTYPE)
_PROTOS = {
    "nlbac_abi_version": [],
    "nlbac_last_error": [],
    "nlbac_elect_selftest": [_P, _P, _P, _I, _I, C.c_uint, _P],
    "nlbac_mlp_pack_layout": [C.POINTER(Mlp)],
    "nlbac_mlp_pack": [C.POINTER(Mlp), _I, _P],
    "nlbac_mlp_fwd": [C.POINTER(Mlp), C.POINTER(MlpIO), _I, _I, _P],
    "nlbac_mlp_fwd_gauss": [C.POINTER(Mlp), C.POINTER(MlpIO), _I, _I, C.POINTER(GaussHead), _P],
    "nlbac_mlp_masks_ok": [C.POINTER(Mlp), _I],
    "nlbac_mlp_fwd_head_ok": [C.POINTER(Mlp), _I],
    "nlbac_mlp_fwd_head": [C.POINTER(Mlp), C.POINTER(MlpIO), _I, _I, C.POINTER(GaussHead), _P],
    "nlbac_mlp_bwd_data": [C.POINTER(Mlp), C.POINTER(MlpIO), _I, _I, _P],
    "nlbac_mlp_bwd_data_head": [C.POINTER(Mlp), C.POINTER(MlpIO), _I, _I, C.POINTER(DyHead), _P],
    "nlbac_mlp_bwd_weights_ws_floats": [C.POINTER(Mlp), _I, _I],
    "nlbac_mlp_bwd_weights": [C.POINTER(Mlp), C.POINTER(MlpIO), _I, _I, _I, _L, _P, _L, _P],
    "nlbac_adam_prepare": [_P, _D, _P],
    "nlbac_adam_step": [_P, _P, _P, _P, _I, _L, _L, _P, _P, _F, _P],
    "nlbac_adam_fused": [_P, _P, _P, _P, _I, _L, _L, _P, _D, _P, _F, _P, _P, _I, _I, C.POINTER(C.c_long),
                         C.POINTER(C.c_void_p), _P, _P, _I, _P],
    "nlbac_node_rk_mask_words": [C.POINTER(Mlp), C.POINTER(Mlp), _I],
    "nlbac_concat_adj_in": [_P, _I, _P, _I, _I, _P, _I, _P, _P, _P],
    "nlbac_concat_adj_out": [_P, _P, _I, _I, _P, _I, _I, _P, _P],
    "nlbac_concat_adj_step_ok": [C.POINTER(Mlp)],
    "nlbac_rk_interp_ok": [C.POINTER(Mlp), C.POINTER(Mlp)],
    "nlbac_node_rk_fwd_begin_ok": [C.POINTER(Mlp), C.POINTER(Mlp), _I, _I],
    "nlbac_node_rk_fwd_begin": [C.POINTER(Mlp), C.POINTER(Mlp), _P, _P, _I, _I, c_float_p, c_float_p, _I, _P, _P, _P, _P, _L,
                                _P, _L, _P, C.POINTER(RkChain), C.POINTER(InMap), _P, C.c_uint, _P],
    "nlbac_concat_adj_step": [C.POINTER(Mlp), _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P,
                              _P, _P, _P, _P, _L, _P, _P, _D, _P],
    "nlbac_node_adj_interp_ok": [C.POINTER(Mlp), C.POINTER(Mlp)],
    "nlbac_reduce_slabs": [_P, _P, _I, _L, _L, _P],
    "nlbac_soft_update": [_P, _P, _L, _F, _P],
    "nlbac_gauss_sample_fwd": [_P, _I, _P, _P, _P, _I, _I, _P, _I, _P, _P],
    "nlbac_gauss_sample_bwd": [_P, _I, _P, _P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _F, _P, _I, _P],
    "nlbac_td_targets": [_P] * 7 + [_I] + [_P] * 4 + [_F, _I, _I] + [_P] * 6 + [_P, _F, _P, _P],
    "nlbac_actor_q_terms": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, C.POINTER(ActorScalarArgs), _P, _P],
    "nlbac_actor_scalars": [_P, _I, _I, _I, _I, _F, _P, _I, _P, _P, _P],
    "nlbac_alpha_refresh": [_P, _I, _I, _I, _P, _P],
    "nlbac_unicycle_state": [_P, _I, _I, _F, _P, _I, _P, _P],
    "nlbac_unicycle_lookahead": [_P, _I, _F, _P, _P],
    "nlbac_unicycle_lookahead_bwd": [_P, _P, _P, _I, _F, _P, _P],
    "nlbac_unicycle_constraints_fwd": [_P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _I, _P, _P, _P, C.POINTER(AuglagArgs), _P, _P, _P],
    "nlbac_auglag": [_P, _I, _I, _I, _F, _I, _I, _I, _I, _F, _F, _P, _P],
    "nlbac_unicycle_constraints_bwd": [_P, _P, _P, _P, _I, _F, _F, _I, _P, _P, _P, _P],
    "nlbac_mse_fwd_bwd": [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _P],
    "nlbac_affine_combine_fwd": [_P, _P, _P, _I, _I, _I, _P, _P],
    "nlbac_affine_combine_bwd": [_P, _P, _P, _I, _I, _I, _F, _P, _P, _I, _P],
    "nlbac_rk_combine": [_P, _P, _I, c_float_p, c_float_p, _P, _I, _I, _I, _I, _P, _P],
    "nlbac_rk_stage_bwd": [_P, _P, _P, _I, _I, c_float_p, c_float_p, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P],
    "nlbac_cars_state": [_P, _I, _I, _P, _P],
    "nlbac_cars_rollout_inputs": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P],
    "nlbac_cars_obs": [_P, _I, _P, _P],
    "nlbac_cars_constraints_fwd": [_P, _P, _P, _P, _P, _F, _F, _F, _I, _P, _P, _P, C.POINTER(AuglagArgs), _P, _P, _P],
    "nlbac_cars_constraints_bwd": [_P, _P, _F, _F, _I, _P, _P, _P, _P, _P],
    "nlbac_add_cols": [_P, _I, _I, _P, _I, _I, _I, _P],
    "nlbac_add_cols_plus": [_P, _I, _I, _P, _I, _I, _I, _P, _I, _P],
    "nlbac_pvtol_state": [_P, _I, _I, _P, _P, _P],
    "nlbac_pvtol_obs_fwd": [_P, _P, _I, _F, _F, _F, _I, _P, _I, _P, _P],
    "nlbac_pvtol_obs_bwd": [_P, _P, _I, _F, _F, _F, _I, _P, _I, _P],
    "nlbac_pvtol_constraints_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _F, _F, _I, _I, _P, _P, _P,
                                    C.POINTER(AuglagArgs), _P, _P, _P],
    "nlbac_pvtol_constraints_bwd": [_P, _P, _P, _P, _P, _P, _I, _F, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P],
    "nlbac_copy_blocks": [_P, _L, _P, _L, _L, _L, _P],
    "nlbac_gather_rows": [_P, _L, _I, _P, _L, _P, _P],
    "nlbac_sample_rows": [_P, _L, _I, _L, _P, _P, _L, C.c_uint64, C.c_uint64, _P],
    "nlbac_td_value": [_P, _P, _I, _P, _I, _P, _F, _I, _I, _P, _P, _P, _P, _F, _P, _P],
    "nlbac_unicycle_obs_fwd": [_P, _I, _F, _F, _P, _I, _P],
    "nlbac_unicycle_obs_bwd": [_P, _P, _I, _I, _F, _F, _P, _I, _P],
    "nlbac_barrier_constraints_fwd": [_P, _P, _P, _P, _F, _F, _F, _I, _P, _P, C.POINTER(AuglagArgs), _P, _P, _P],
    "nlbac_barrier_constraints_bwd": [_P, _F, _F, _I, _P, _P, _P, _P],
    "nlbac_node_rk_fwd": [C.POINTER(Mlp), C.POINTER(Mlp), _P, _P, _I, _I, _I, _I, _I, c_float_p, c_float_p, _I,
                          c_float_p, _I, c_float_p, _P, _I, _P, _P, _P, _P, _L, _P, _L, _I, _P, _P, C.POINTER(RkChain),
                          C.POINTER(InMap), _P],
    "nlbac_node_rk_bwd": [C.POINTER(Mlp), C.POINTER(Mlp), _P, _P, _I, _I, _I, _I, _I, _I, c_float_p, c_float_p, _P, _I,
                          _P, _L, _P, _L, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, C.POINTER(RkChain), _I, _P],
    "nlbac_concat_rk_fwd": [C.POINTER(Mlp), _P, _P, _I, _I, _I, _I, _I, c_float_p, c_float_p, _I, c_float_p, _I, c_float_p,
                            _P, _I, _P, _P, _P, _L, _I, _P, _P, _P, _P, C.POINTER(RkChain), _P],
    "nlbac_concat_rk_mask_words": [C.POINTER(Mlp)],
    "nlbac_concat_rk_bwd": [C.POINTER(Mlp), _I, _I, _I, _I, _I, _I, c_float_p, c_float_p, _P, _I, _P, _L, _I, _P, _P, _P, _P,
                            _I, _P, _I, _P, _P, C.POINTER(RkChain), _I, _P],
    "nlbac_node_adj_step": [C.POINTER(Mlp), C.POINTER(Mlp), _P, _I, _I, _I, _I, _I, c_float_p, c_float_p, _I, c_float_p,
                            _I, c_float_p, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P, _P, _P, _D, _P],
    "nlbac_adj_pack": [_P, _P, _I, _I, _I, _P, _P],
    "nlbac_adj_unpack": [_P, _I, _I, _I, _P, _P, _P],
    "nlbac_adj_norm_control": [_P, _P, _P, _P, _P, _I, _F, _F, _I, _I, _I, _I, _D, _P, _P, _P, _P, _P, _D, _P],
    "nlbac_adj_control": [_P, _I, _I, _I, _I, _I, _I, _D, _P, _P, _P],
    "nlbac_adj_param_norm": [_I, _P, _P, _L, _I, c_float_p, c_float_p, c_float_p, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P, _P,
                             _P],
    "nlbac_adj_commit": [_P, _I, _L, _I, _P, _P, _P, _P, _P],
    "nlbac_dopri_norm_partials": [_P, _P, _P, _P, _P, _I, _F, _F, _I, _I, _I, _I, _P, _P, _L, _P],
    "nlbac_dopri_norm_control": [_P, _P, _P, _P, _P, _I, _F, _F, _I, _I, _I, _I, _D, _P, _P, _P, C.POINTER(RkChain), _P],
    "nlbac_dopri_control": [_P, _I, _I, _I, _I, _I, _I, _D, _P, _I, _P, _P, _I, _P],
    "nlbac_dopri_control_tiles": [C.POINTER(RkChain), _I, _I, _I, _I, _P],
    "nlbac_dopri_interp_fwd": [_P, _P, _P, c_float_p, c_float_p, _P, _I, _I, _I, _P, _L, C.POINTER(OutMap), _P],
    "nlbac_dopri_interp_bwd": [_P, c_float_p, c_float_p, _P, _I, _I, _I, _P, _P, _P, _L, C.POINTER(OutMap), _P],
    "nlbac_unicycle_env_step": [_I, c_double_p, _I] + [_P, _I] + [_P] * 13,
    "nlbac_pvtol_env_step": [_I, c_double_p, _I] + [_P, _I] + [_P] * 11,
    "nlbac_cars_env_step": [_I, c_double_p, _I] + [_P] * 12,
    "nlbac_axpby": [_F, _P, _F, _P, _L, _P, _P],
    "nlbac_fill": [_P, _F, _L, _P],
    "nlbac_sum_partials": [_P, _I, _I, _F, _P, _P],
}
_RESTYPE = {"nlbac_last_error": C.c_char_p, "nlbac_mlp_bwd_weights_ws_floats": C.c_long}

EXPORTS = tuple(_PROTOS)

_lib = None


def build(verbose=False):
    """Compile every HIP source for gfx950 into lib/libnlbac_hip.so (in-tree)."""
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise NlbacError("hipcc build failed (see output above)")
    return LIB_PATH


ABI_VERSION = 16      # == NLBAC_ABI_VERSION of include/nlbac_hip.h (bumped with every signature / struct change)


def _stale_sources():
    """Sources / headers newer than the built library (only where the sources are present next to it)."""
    try:
        t_lib = os.path.getmtime(LIB_PATH)
        header = os.path.join(os.path.dirname(_HERE), "include", "nlbac_hip.h")
        srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + [header]
        return [os.path.basename(f) for f in srcs if os.path.exists(f) and os.path.getmtime(f) > t_lib + 1.0]
    except OSError:
        return []


def load():
    """Load the shared library once and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NlbacError("HIP extension missing: %s (run __graft_entry__.build() / make -C %s); "
                         "there is no CPU fallback" % (LIB_PATH, CSRC))
    # PyTorch first: its wheel carries its own libamdhip64.  Loaded after it, this library binds to that same runtime;
    # loaded BEFORE it (nothing else of the package imports torch on the way here), it would pull the system's copy and
    # the process would hold two HIP runtimes — the second one to initialise sees "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    if lib.nlbac_abi_version() != ABI_VERSION:
        raise NlbacError("libnlbac_hip.so reports ABI version %d, this binding is written against %d: rebuild it "
                         "(make -C %s)" % (lib.nlbac_abi_version(), ABI_VERSION, CSRC))
    stale = _stale_sources()
    if stale:      # (a warning, not an error: file times do not survive every way a tree is copied to another machine)
        import warnings
        warnings.warn("libnlbac_hip.so is older than %s: rebuild it (make -C %s)" % (", ".join(stale), CSRC))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise NlbacError("%s failed: %s" % (what, load().nlbac_last_error().decode()))


def call(name, *args):
    """Call an int-returning entry point and raise on error."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise NlbacError("%s: %s" % (name, lib.nlbac_last_error().decode()))


def fptr(*vals):
    """Host float array argument (passed by value into kernel args)."""
    return (C.c_float * len(vals))(*vals)
