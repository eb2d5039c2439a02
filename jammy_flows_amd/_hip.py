"""ctypes binding of libjammy_hip.so (the C ABI declared in include/jammy_hip.h).

This is the ONLY compute path of the package: there is no CPU / eager fallback.  Anything that cannot run on the
HIP kernels raises (missing library, CPU tensors, unsupported option) -- silently routing through PyTorch ops would
void every parity claim made for the kernels.
"""
import ctypes
import threading
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# JF_LIB_PATH: a private build of the same library (instrumented probes, scripts/probe/pp_trace.sh); the product loads the in-tree one
NEWTON_RULE = os.environ.get("JF_NEWTON_RULE", "product")
if NEWTON_RULE not in ("product", "reference"):
    raise ValueError("JF_NEWTON_RULE must be 'reference' or 'product', not %r" % NEWTON_RULE)
LIB_PATH = os.environ.get("JF_LIB_PATH") or os.path.join(_HERE, "libjammy_hip_audit.so" if NEWTON_RULE == "reference" else "libjammy_hip.so")

JF_OK, JF_ERR_BADARG, JF_ERR_UNSUPPORTED, JF_ERR_LAUNCH = 0, -1, -2, -3
JF_ERRORS = {-1: "bad argument", -2: "unsupported configuration (a kernel cap: g / t beyond their dimension cap, spline bins, chain too long, LDS budget ...)", -3: "kernel launch failed"}
JF_STATUS_WORDS = 4
JF_STATUS_NONCONVERGED, JF_STATUS_NONFINITE, JF_STATUS_OUT_OF_RANGE, JF_STATUS_NEWTON_STEPS = 0, 1, 2, 3
JF_MAX_CHAIN = 8

GF_INV_TYPES = {"isigmoid": 0, "inormal_partly_precise": 1, "inormal_partly_crude": 2, "inormal_full_pade": 3}
GF_WIDTH_SMOOTH, GF_WIDTH_EXP, GF_WIDTH_SOFTPLUS = 0, 1, 2
GF_STRETCH_CLASSIC, GF_STRETCH_RQ_SPLINES = 0, 1
GF_ROT_MODES = {"householder": 0, "angles": 1, "cayley": 2, "triangular_combination": 3}
JF_SPLINE_MAX_BINS = 64          # 'g' with the rq_splines stretch (= JF_SPLINE_CAP since round 6: a lane's knot table follows the chain's own bin count)
JF_SPLINE_CAP = 64               # 'r', 'o', splines nested in 'f' (tables at the chain's own bin count; the launch sizes its row tile to the LDS)
JF_CORR_SCRATCH = 81


class HipUnavailable(RuntimeError):
    pass


class jf_gf_layer(ctypes.Structure):
    _fields_ = [("num_kde", ctypes.c_int32), ("hh_iter", ctypes.c_int32), ("model_offset", ctypes.c_int32),
                ("fit_normalization", ctypes.c_int32), ("regulate_normalization", ctypes.c_int32),
                ("inverse_function_type", ctypes.c_int32), ("width_mode", ctypes.c_int32), ("clamp_widths", ctypes.c_int32),
                ("nonlinear_stretch_type", ctypes.c_int32), ("rotation_mode", ctypes.c_int32),
                ("center_mean", ctypes.c_int32), ("add_skewness", ctypes.c_int32),
                ("width_min", ctypes.c_double), ("width_max", ctypes.c_double), ("norm_min", ctypes.c_double),
                ("norm_max", ctypes.c_double)]


class jf_spline_opts(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("num_bins", "smooth", "fix_first", "fix_second", "independent", "fix_bd", "n_w", "n_h", "n_d",
                                               "reserved")] + [(n, ctypes.c_double) for n in ("fix_bd_value", "min_w", "min_h", "min_d", "ratio")]


class jf_r_layer(ctypes.Structure):
    _fields_ = [("sp", jf_spline_opts), ("lo", ctypes.c_double), ("hi", ctypes.c_double), ("first", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class jf_o_layer(ctypes.Structure):
    _fields_ = [("sp", jf_spline_opts), ("natural_direction", ctypes.c_int32), ("hh_iter", ctypes.c_int32), ("first", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class jf_m_layer(ctypes.Structure):
    _fields_ = [("num_components", ctypes.c_int32), ("natural_direction", ctypes.c_int32), ("hh_iter", ctypes.c_int32), ("first", ctypes.c_int32),
                ("omega_pars", ctypes.c_int32)]


JF_MAX_MCHAIN = 4
JF_MAX_NESTED = 4


class jf_f_layer(ctypes.Structure):
    _fields_ = [("hh_iter", ctypes.c_int32), ("first", ctypes.c_int32), ("n_vertical", ctypes.c_int32), ("n_circular", ctypes.c_int32),
                ("correlated", ctypes.c_int32), ("corr_hidden", ctypes.c_int32), ("corr_rank", ctypes.c_int32), ("corr_full2", ctypes.c_int32),
                ("kappa_mode", ctypes.c_int32), ("kappa_clamping", ctypes.c_int32), ("extra_rotation", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("z_sign", ctypes.c_double), ("min_kappa", ctypes.c_double), ("identity_region", ctypes.c_double),
                ("vertical", jf_r_layer * JF_MAX_NESTED), ("circular", jf_o_layer * JF_MAX_NESTED)]


F_KAPPA_MODES = {"direct_log_real_bounded": 0, "softplus_real_bounded": 1, "log_bounded": 2, "mu": 3, "mu_squared": 4, "quatvec": 5,
                 "quatvec_squared": 6}
ROT_CODES = {"angles": -1, "xyz": -2, "quaternion": -3}      # hh_iter encoding of the non-Householder rotation modes
V_KINDS = {"linear": 0, "quadratic": 1, "exponential": 2, "splines": 3}


class jf_v_layer(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("num_components", "exp_map_type", "natural_direction", "hh_iter", "max_newton_iter", "first")]


class jf_c_layer(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("hh_iter", ctypes.c_int32), ("first", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("lo", ctypes.c_double), ("hi", ctypes.c_double)]


class jf_t_layer(ctypes.Structure):
    _fields_ = [("cov_type", ctypes.c_int32), ("model_offset", ctypes.c_int32), ("width_mode", ctypes.c_int32), ("clamp_widths", ctypes.c_int32),
                ("width_min", ctypes.c_double), ("width_max", ctypes.c_double)]


JF_MAX_ROW_LISTS = 16


class jf_row_list(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p * 16), ("n", ctypes.c_int32)]


class jf_adam_tensor(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p), ("n", ctypes.c_int64)]


JF_ADAM_MAX_TENSORS = 64


class jf_cond_segment(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("stride", ctypes.c_int64), ("kind", ctypes.c_int32), ("n_in", ctypes.c_int32)]


JF_MAX_SEGMENTS = 16
MCHAIN_LAYER_TYPES = {"r": jf_r_layer, "o": jf_o_layer, "m": jf_m_layer, "f": jf_f_layer, "v": jf_v_layer, "c": jf_c_layer}

_lib = None

_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_I32 = ctypes.c_int32

# name -> argtypes (without the _f32/_f64 suffix); every entry point of include/jammy_hip.h
_SIGNATURES = {
    "jf_gf_chain_inv": [_P, _I64, _P, _P, _I64, _I32, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _I64, _P, _P, _P, _P, _I64, _P, _P],
    "jf_gf_chain_inv_total": [_P, _I64, _P, _P, _I64, _I32, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _I64, _P, _P, _P, _P, _P, _I64, _P, _P],
    "jf_cond_gf_chain_inv": [_P, _I64, _P, _I64, _P, _P, _I64, _P, _I32, _I32, _P, _I64, _P, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _I64,
                             _P, _P, _P, _P, _P],
    "jf_amlp_gf_chain_inv": [_P, _I64, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _I64, _P, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P,
                             _I64, _P, _P, _P, _P, _P],
    "jf_linear_wgrad": [_P, _I64, _P, _I64, _I64, _I32, _I32, _P, _P, _P],
    "jf_amlp2": [_P, _I64, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _I64, _P],
    "jf_gf_chain_fwd": [_P, _I64, _P, _P, _I64, _I32, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _I64, _P, _P, _I64, _P, _P],
    "jf_gf_chain_fwd_tab": [_P, _I64, _P, _P, _I64, _I32, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _I64, _P, _P, _I64, _P, _P, _P],
    "jf_linear": [_P, _I64, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _I64, _P],
    "jf_normal_logp": [_P, _I64, _I64, _I32, _P, _P, _P],
    "jf_slab_sum": [_P, _I64, _P, _P, _I64, _P, _I32, _I32, _P],
    "jf_slab_sum_map": [_P, _I64, _P, _I64, _P, _P, _I32, _P],
    "jf_tanh_bwd": [_P, _P, _I64, _P, _P],
    "jf_mlp_hidden_bwd": [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I32, _I32, _P, _P],
    "jf_mlp2_small_bwd": [_P, _I64, _P, _I64, _P, _P, _I64, _P, _I64, _I64, _I32, _I32, _I32, _P, _P, _P],
    "jf_activation": [_P, _I64, _I32, _P, _P],
    "jf_device_math": [_P, _I64, _I32, _P, _P],
    "jf_add_rows": [_P, _P, _I64, _P, _P],
    "jf_gf_chain_inv_cot": [_P, _I64, _P, _I64, _I32, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _I64, _P, _I64, _P, _I64, _P, _P],
    "jf_adam_step": [ctypes.POINTER(jf_adam_tensor), _I32, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _I64, _P],
    "jf_adam_step_dev": [ctypes.POINTER(jf_adam_tensor), _I32, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _P, _P],
    "jf_combine_rows": [ctypes.POINTER(jf_row_list), ctypes.POINTER(jf_row_list), _I64, _P, _P, _P, _P],
    "jf_activation_bwd": [_P, _P, _I64, _I32, _P, _P],
    "jf_mlp2": [_P, _I64, _P, _I64, _P, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _I64, _P],
    "jf_sphere_to_embedding": [_P, _I64, _P, _I64, _I32, _P, _I64, _P, _P],
    "jf_sphere_from_embedding": [_P, _I64, _P, _I64, _I32, _P, _I64, _P, _P],
    "jf_conditioning_rows": [ctypes.POINTER(jf_cond_segment), _I32, _I64, _P, _I64, _P],
    "jf_coverage_histogram": [_P, _I64, ctypes.c_double, _P, _I32, _P, _P, _P],
    "jf_segment_reduce": [_P, _I64, _I64, _I32, _P, _P],
    "jf_amlp_stage": [_P, _I64, _P, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _P, _I64, _P, _I64, _P],
    "jf_amlp_stage_bwd": [_P, _I64, _P, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P],
    "jf_t_layer_inv": [_P, _I64, _P, _P, _I64, _I32, _I64, _I32, ctypes.POINTER(jf_t_layer), _P, _I64, _P, _P, _P, _P, _P],
    "jf_t_layer_fwd": [_P, _I64, _P, _P, _I64, _I32, _I64, _I32, ctypes.POINTER(jf_t_layer), _P, _I64, _P, _P, _P, _P, _P],
    "jf_t_layer_inv_bwd": [_P, _I64, _P, _I64, _I32, _I64, _I32, ctypes.POINTER(jf_t_layer), _P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _P],
    "jf_gf_chain_inv_bwd": [_P, _I64, _P, _I64, _I32, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _P],
}
# entry points that exist for one precision only: full symbol name -> (argtypes, restype)
_SIGNATURES_SINGLE = {
    "jf_get_newton_rule": ([], ctypes.c_int),
    "jf_gf_bcast_lane_rows": ([_I64], ctypes.c_int64),
    "jf_merge_begin": ([], ctypes.c_int),
    "jf_merge_abort": ([], ctypes.c_int),
    "jf_merge_captured": ([], ctypes.c_int),
    "jf_merge_end": ([_P], ctypes.c_int),
    "jf_plan_create": ([], ctypes.c_int64),
    "jf_plan_destroy": ([_I64], ctypes.c_int32),
    "jf_plan_add_slot": ([_I64, _P, _I64], ctypes.c_int32),
    "jf_plan_record_begin": ([_I64], ctypes.c_int32),
    "jf_plan_record_end": ([_I64], ctypes.c_int32),
    "jf_plan_add_memset": ([_I64, _P, _I32, _I64], ctypes.c_int32),
    "jf_plan_add_copy_to_host": ([_I64, _P, _P, _I64], ctypes.c_int32),
    "jf_plan_num_ops": ([_I64], ctypes.c_int32),
    "jf_plan_set_lane": ([_I64, _I32], ctypes.c_int32),
    "jf_plan_set_any_order": ([_I64, _I32], ctypes.c_int32),
    "jf_plan_add_fork": ([_I64], ctypes.c_int32),
    "jf_plan_add_join": ([_I64], ctypes.c_int32),
    "jf_plan_num_relocations": ([_I64], ctypes.c_int32),
    "jf_plan_launch": ([_I64, ctypes.POINTER(ctypes.c_void_p), _I32, _P], ctypes.c_int32),
    "jf_plan_set_timing": ([_I64, _I32], ctypes.c_int32),
    "jf_plan_debug_words": ([_I64, _I32, ctypes.POINTER(ctypes.c_uint64), _I32], ctypes.c_int32),
    "jf_plan_read_timing": ([_I64, ctypes.POINTER(ctypes.c_double), _I32, ctypes.POINTER(ctypes.c_int64), _I32], ctypes.c_int32),
    "jf_gf_chain_lds_bytes_f32": ([_I32, _I32, ctypes.POINTER(jf_gf_layer), _I32], ctypes.c_int64),
    "jf_gf_chain_lds_bytes_f64": ([_I32, _I32, ctypes.POINTER(jf_gf_layer), _I32], ctypes.c_int64),
    "jf_gf_chain_inv_bwd_lds_bytes_f32": ([_I32, _I32, ctypes.POINTER(jf_gf_layer), _I32], ctypes.c_int64),
    "jf_gf_chain_inv_bwd_lds_bytes_f64": ([_I32, _I32, ctypes.POINTER(jf_gf_layer), _I32], ctypes.c_int64),
    "jf_linear_wgrad_splits_f32": ([_I64, _I32, _I32], ctypes.c_int64),
    "jf_linear_wgrad_splits_f64": ([_I64, _I32, _I32], ctypes.c_int64),
    "jf_gf_chain_inv_bwd_partials": ([_I64, _I32], ctypes.c_int64),
    "jf_gf_chain_fwd_table_elems": ([_I32, _I32], ctypes.c_int64),
    "jf_linear_split_packed_bytes": ([_I32, _I32], ctypes.c_int64),
    "jf_mlp2_small_bwd_slabs": ([_I64], ctypes.c_int64),
    "jf_mlp2_i8_packed_bytes": ([_I32, _I32], ctypes.c_int64),
    "jf_mlp2_i8_pack_f64": ([_P, _I64, _P, _I32, _I32, _I32, _P, _P], ctypes.c_int),
    "jf_mlp2_i8_f64": ([_P, _I64, _P, _I64, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _I64, _P], ctypes.c_int),
    "jf_mlp2_i8_seg_f64": ([ctypes.POINTER(jf_cond_segment), _I32, _P, _I64, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _I64, _P], ctypes.c_int),
    "jf_cond_gf_chain_split3_f32": ([_I32, _I32, ctypes.POINTER(jf_cond_segment), _I32, _P, _I64, _P, _P, _I32, _I32, _P, _I64, _P, _I64, _I32, _I32,
                                     ctypes.POINTER(jf_gf_layer), _P, _I64, _P, _P, _P, _P, ctypes.POINTER(jf_row_list), ctypes.POINTER(jf_row_list), _P,
                                     _P, _P], ctypes.c_int),
    "jf_linear_wgrad_split_splits": ([_I64, _I32], ctypes.c_int64),
    "jf_linear_wgrad_split_f32": ([_P, _I64, _P, _I64, _I64, _I32, _I32, _P, _P, _P], ctypes.c_int),
    "jf_linear_split_pack_f32": ([_P, _I64, _I64, _I32, _I32, _P, _P], ctypes.c_int),
    "jf_linear_split_f32": ([_P, _I64, _P, _P, _I64, _I32, _I32, _P, _I64, _P], ctypes.c_int),
    "jf_cond_gf_packed_bytes": ([_I32, _I32, ctypes.POINTER(jf_gf_layer)], ctypes.c_int64),
    "jf_cond_gf_split_row_groups": ([_I32], ctypes.c_int),
    "jf_cond_gf_chain_fwd_split_f32": ([_P, _I64, _P, _I64, _P, _P, _I32, _I32, _P, _I64, _P, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _I64,
                                        _P, _P, _P], ctypes.c_int),
    "jf_cond_gf_pack_f32": ([_P, _I64, _P, _I32, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _P], ctypes.c_int),
    "jf_amlp_gf_chain_fwd_f64": ([_P, _I64, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _I64, _P, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer),
                                  _P, _I64, _P, _P, _P], ctypes.c_int),
    "jf_lowrank_gf_chain_inv_f64": ([_P, _I64, _P, _P, _I32, _P, _I64, _P, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _I64, _P, _P, _P, _P, _P, _P],
                                    ctypes.c_int),
    "jf_lowrank_gf_workspace_doubles": ([_I64, _I32], ctypes.c_int64),
    "jf_lowrank_head_f64": ([_P, _I64, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P], ctypes.c_int),
    "jf_lowrank_head_workspace_doubles": ([_I64, _I32, _I32], ctypes.c_int64),
    "jf_lowrank_head_bwd_f64": ([_P, _I64, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P], ctypes.c_int),
    "jf_lowrank_gf_chain_inv_bwd_f64": ([_P, _I64, _P, _P, _I32, _P, _P, _I64, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _I64, _P, _P, _P, _I64,
                                         _P, _P, _P, _P, _P, _P], ctypes.c_int),
    "jf_cond_gf_packed_bytes2": ([_I32, _I32, ctypes.POINTER(jf_gf_layer), _I32], ctypes.c_int64),
    "jf_cond_gf_pack2_f32": ([_P, _I64, _P, _I32, _I32, _I32, ctypes.POINTER(jf_gf_layer), _I32, _P, _P], ctypes.c_int),
    "jf_cond_gf_chain_split2_f32": ([_I32, _I32, _P, _I64, _P, _I64, _P, _P, _I32, _I32, _P, _I64, _P, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer),
                                     _P, _I64, _P, _P, _P, _P, _P, _P], ctypes.c_int),
    "jf_cond_gf_chain_inv_split_save_f32": ([_P, _I64, _P, _I64, _P, _P, _I32, _I32, _P, _I64, _P, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P,
                                             _I64, _P, _P, _P, _P, _P, _P], ctypes.c_int),
    "jf_cond_gf_aux_floats": ([_I64, _I32], ctypes.c_int64),
    "jf_cond_gf_bwd_packed_bytes": ([_I32, _I32, ctypes.POINTER(jf_gf_layer)], ctypes.c_int64),
    "jf_cond_gf_bwd_pack_f32": ([_P, _I64, _I32, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _P], ctypes.c_int),
    "jf_cond_gf_chain_inv_split_bwd_f32": ([_P, _I64, _P, _I64, _P, _P, _P, _I32, _I32, _P, _I64, _P, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer),
                                            _P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P], ctypes.c_int),
    "jf_linear_wgrad_split16_f32": ([_P, _I64, _P, _I64, _I64, _I32, _I32, _P, _I32, _P, _P, _P], ctypes.c_int),
    "jf_cond_gf_chain_inv_split_f32": ([_P, _I64, _P, _I64, _P, _P, _I32, _I32, _P, _I64, _P, _I64, _I32, _I32, ctypes.POINTER(jf_gf_layer), _P, _I64,
                                        _P, _P, _P, _P, _P], ctypes.c_int),
}
for _fam, _cls in MCHAIN_LAYER_TYPES.items():
    for _d in ("inv", "fwd"):
        _SIGNATURES["jf_%s_chain_%s" % (_fam, _d)] = [_P, _I64, _P, _P, _I64, _I32, _I64, _I32, ctypes.POINTER(_cls), _P, _I64, _P, _P, _P, _P,
                                                      _I64, _P, _P]
    _SIGNATURES["jf_%s_chain_inv_sum" % _fam] = [_P, _I64, _P, _P, _I64, _I32, _I64, _I32, ctypes.POINTER(_cls), _P, _I64, _P, _P, _P,
                                                 ctypes.POINTER(jf_row_list), ctypes.POINTER(jf_row_list), _P, _P, _I64, _P, _P]
    if _fam in "romf":
        _SIGNATURES["jf_cond_%s_chain_inv" % _fam] = [_P, _I64, _P, _I64, _P, _P, _I64, _P, _I32, _I32, _P, _I64, _P, _I64, _I32, ctypes.POINTER(_cls),
                                                      _P, _I64, _P, _P, _P, _P, _P]
        _SIGNATURES["jf_cond_%s_chain_fwd" % _fam] = [_P, _I64, _P, _I64, _P, _P, _I64, _P, _I32, _I32, _P, _I64, _P, _I64, _I32, ctypes.POINTER(_cls),
                                                      _P, _I64, _P, _P, _P]
    _SIGNATURES["jf_%s_chain_inv_bwd" % _fam] = [_P, _I64, _P, _I64, _I32, _I64, _I32, ctypes.POINTER(_cls), _P, _I64, _P, _P, _P, _I64, _P, _I64,
                                                 _P, _P]


def exported_symbols():
    """all symbols include/jammy_hip.h declares (used by the CPU-side ABI test)."""
    names = ["jf_abi_version"]
    for base in _SIGNATURES:
        names += [base + "_f32", base + "_f64"]
    return names + list(_SIGNATURES_SINGLE)


def lib():
    """load (once) and return the shared library; raises HipUnavailable with a build hint when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipUnavailable("libjammy_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(or `make -C jammy_flows_amd/csrc`); jammy_flows_amd has no CPU fallback" % LIB_PATH)
    try:
        l = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise HipUnavailable("cannot load %s: %s" % (LIB_PATH, e))
    l.jf_abi_version.restype = ctypes.c_int
    for base, argtypes in _SIGNATURES.items():
        for suf in ("_f32", "_f64"):
            fn = getattr(l, base + suf)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
    for name, (argtypes, restype) in _SIGNATURES_SINGLE.items():
        fn = getattr(l, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = l
    if NEWTON_RULE == "reference" and int(l.jf_get_newton_rule()) != 1:
        raise HipUnavailable("JF_NEWTON_RULE=reference, but %s was not built with the reference's solver rule" % LIB_PATH)
    return _lib


def get_newton_rule():
    """'product' (default) or 'reference': which iteration rule the loaded library's solvers follow.  JF_NEWTON_RULE=reference in the environment
    (read at import) loads libjammy_hip_audit.so -- the same kernels built with the reference's own iteration: 25 bisections on [-1e5, 1e5], Newton
    until the row's update sum is below 1e-14 or 20 steps are done, no float32 floor, 'v' until 1e-12 (bisection_n_newton.py:11-135, 330-465)."""
    return "reference" if int(lib().jf_get_newton_rule()) else "product"


def _suffix(t):
    if t.dtype == torch.float32:
        return "_f32"
    if t.dtype == torch.float64:
        return "_f64"
    raise TypeError("jammy_flows_amd kernels exist for float32 and float64 only, got %s" % t.dtype)


def require_device(*tensors):
    """every tensor must live on a HIP device (no CPU path exists)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HipUnavailable("jammy_flows_amd computes on MI355X (HIP) tensors only; got a %s tensor. "
                                 "Move the pdf and its inputs to 'cuda' -- there is no CPU fallback." % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError("tensors on different devices: %s vs %s" % (dev, t.device))
    return dev


_EMPTY_SENTINEL = 16     # see _ptr


def _ptr(t):
    """device address of a tensor for the C ABI; None = absent (NULL).  torch hands out a NULL data_ptr for zero-element tensors, which the C side
    would read as "absent argument": an empty batch is passed as a non-null sentinel address instead -- never dereferenced, every entry point
    returns JF_OK before touching memory when B == 0."""
    if t is None:
        return None
    p = t.data_ptr()
    return p if (p != 0 or t.numel() != 0) else _EMPTY_SENTINEL


def _stream(dev=None):
    return torch.cuda.current_stream(dev).cuda_stream


# Side streams the library itself issues work on (the training streams of default.pdf).  The caching allocator hands a freed block back to the
# pool of the stream it was ALLOCATED on: a tensor allocated on the caller's stream that a side-stream kernel still reads (or the reverse) can be
# handed out again while that kernel runs, if the host drops its last reference too early -- e.g. a saved tensor, released as soon as
# its backward node returns while the node's kernel is still in flight on the side stream.  record_stream would announce every such use, at a
# price: ~2 us per call and, worse, an event per recorded block when it is freed (the eager C3 training step: 1.33 -> 1.66 ms).  Instead the
# tensors are KEPT ALIVE until the streams have met again: every tensor that crosses between the caller's stream and a side stream is appended to
# KEEPALIVE, which the next gradient-mode forward call empties -- by then the caller's stream has waited for the side streams (at the end of the
# forward pass: default.pdf._forward_with_grad; at the end of backward(): the autograd engine synchronises the streams it used with the
# caller's), so whatever reuses the blocks is ordered behind their last reader.
SIDE_STREAMS = {}                 # cuda_stream handle -> torch.cuda.Stream
KEEPALIVE = []


def register_side_stream(st):
    SIDE_STREAMS[st.cuda_stream] = st


def keep_alive(*tensors):
    """hold these tensors (None entries are skipped) until release_keepalive()"""
    KEEPALIVE.extend(t for t in tensors if isinstance(t, torch.Tensor))


def release_keepalive():
    """the caller's stream and the side streams have met since the kept tensors' last use (see above)"""
    if KEEPALIVE:
        KEEPALIVE.clear()


def keep_if_side_stream(*tensors):
    """inside a custom backward: when the node runs on one of the library's side streams (autograd runs a node on its forward's stream), the
    tensors it reads there must outlive its kernels"""
    if SIDE_STREAMS and torch.cuda.current_stream().cuda_stream in SIDE_STREAMS:
        keep_alive(*tensors)


class KernelTimer:
    """records a pair of HIP events (on torch's current stream = the stream the kernels are launched on) around every kernel launch
    made through this module while active; `summary()` (after a synchronize) gives per-kernel launch counts and mean durations."""

    def __init__(self, plan_every=1):
        self.records = []
        self.plans = []
        self.plan_every = plan_every        # steps replayed from a plan: events on every n-th replay (the instrumentation then costs 1 / n of ~2 %)

    def __enter__(self):
        global _TIMER
        self._prev = _TIMER
        _TIMER = self
        return self

    def __exit__(self, *exc):
        global _TIMER
        _TIMER = self._prev

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, tag, e0, e1 in self.records:
            d = out.setdefault((name, tag), [0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
        for plan in self.plans:                     # steps replayed from a plan while this timer was active (events recorded by jf_plan_launch)
            for (name, tag), (n, ms) in plan.read_timing().items():
                # plan_every > 1: only every n-th replay carried events; launches / total_ms are scaled to ALL replays (mean_ms is the measured mean)
                scale = (plan.replays_since_read / n) if (n > 0 and plan.replays_since_read > n) else 1.0
                d = out.setdefault((name, tag), [0, 0.0])
                d[0] += n * scale
                d[1] += ms * scale
        return {k: {"launches": v[0], "mean_ms": v[1] / v[0], "total_ms": v[1]} for k, v in out.items() if v[0]}


_TIMER = None
# the StepPlan being recorded on THIS thread (entry points then append to it instead of launching).  Per thread, like the C side's launch sink
# (csrc/plan.hip: thread_local): another thread's pdf.forward during a recording launches for real and appends nothing.
# `_hip._RECORDING` (read) resolves through the module __getattr__ below.
_TLS = threading.local()


def _recording():
    return getattr(_TLS, "plan", None)


def __getattr__(name):
    if name == "_RECORDING":
        return _recording()
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


def _launch(name, tag, args, dev, unsupported_ok=False):
    """call entry point `name` with `args` + the stream argument.  The launch goes to the TENSORS' device (`dev`, from require_device) and to
    torch's current stream OF THAT DEVICE -- not to whatever device happens to be current: the C side sizes grids with hipGetDevice and a
    kernel launched on device 0 with device-1 pointers faults or computes on the wrong GPU."""
    fn = getattr(lib(), name)
    if dev is None:
        raise HipUnavailable("%s: no device tensor among the arguments" % name)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        args = tuple(args) + (stream.cuda_stream,)
        rec = _recording()
        if rec is not None:
            n0 = rec.num_ops()
            rc = fn(*args)
            if rc == JF_OK:
                rec.calls.append((name, tag, n0, rec.num_ops()))
        elif _TIMER is None or _MERGING:              # (a captured launch is issued by merge_end: nothing to time here)
            rc = fn(*args)
        else:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rc = fn(*args)
            e1.record(stream)
            if not (unsupported_ok and rc == JF_ERR_UNSUPPORTED):
                _TIMER.records.append((name, tag, e0, e1))
    if unsupported_ok and rc == JF_ERR_UNSUPPORTED:
        return False                                  # the caller has another kernel path for this configuration
    _check(rc, name)
    return True


def _check(rc, what):
    if rc != JF_OK:
        raise RuntimeError("%s failed: %s (code %d)" % (what, JF_ERRORS.get(rc, "unknown"), rc))


def _rowmajor(t):
    """(B, n) tensor usable by the kernels: unit stride in the last dim, any row stride."""
    if t.dim() != 2:
        raise ValueError("expected a 2-d tensor, got shape %s" % (tuple(t.shape),))
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[1] == 1 and t.stride(0) < 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) == 0:      # expanded rows (stride 0): kernels that derive row counts from the stride divide by it
        t = t.contiguous()
    return t


def new_status(device):
    return torch.zeros(JF_STATUS_WORDS, dtype=torch.int32, device=device)


class _MappedHolder:
    def __init__(self, t):
        self.t = t
        self.__cuda_array_interface__ = {"shape": tuple(t.shape), "typestr": "<i4", "data": (t.data_ptr(), False), "version": 2}


def mapped_status(device):
    """status words in PINNED HOST memory -> (host tensor, device-side view of the same words).  The kernels raise a status word with one
    atomic per wave and only when something is wrong (csrc/jf_common.h status_add), so words that live on the host cost a healthy step
    nothing -- and the host can look at them without a copy-back on the stream (a recorded step spent ~5 us per replay on its 16-byte
    __amd_rocclr_copyBuffer: 4 % of a 2^17-row step).  Pinned memory is mapped into the device's address space at the same address."""
    host = torch.zeros(JF_STATUS_WORDS, dtype=torch.int32).pin_memory()
    with torch.cuda.device(device):
        view = torch.as_tensor(_MappedHolder(host), device=device)
    if view.data_ptr() != host.data_ptr():
        raise RuntimeError("pinned host memory is not mapped at its own address on this platform")
    view._jf_host = host                      # the view does not own the memory
    return host, view


# --------------------------------------------------------------------------------------------------------------
class StepPlan:
    """a recorded step (include/jammy_hip.h "step plans", csrc/plan.hip): every launch the entry points make between begin() and end() is
    stored instead of issued; launch(bases) re-issues all of them from C in one call, with the device pointers into the declared slots
    (buffers whose address changes between replays) rebound."""

    def __init__(self):
        self.handle = int(lib().jf_plan_create())
        if self.handle <= 0:
            raise RuntimeError("jf_plan_create failed (%d)" % self.handle)
        self.calls = []            # (entry point, tag, first op, one past last op) per recorded entry-point call
        self.n_slots = 0
        self._bases = None
        self._timing = 0

    def __del__(self):
        try:
            if _lib is not None and self.handle > 0:
                _lib.jf_plan_destroy(self.handle)
        except Exception:           # noqa: BLE001 -- interpreter shutdown
            pass

    def add_slot(self, t):
        """declare tensor `t`'s memory [data_ptr, data_ptr + span) as rebindable; returns the slot index"""
        span = (sum((n - 1) * st for n, st in zip(t.shape, t.stride())) + 1) * t.element_size() if t.numel() else 0
        rc = int(lib().jf_plan_add_slot(self.handle, _ptr(t), span))
        _check(min(rc, 0), "jf_plan_add_slot")
        self.n_slots = rc + 1
        return rc

    def begin(self):
        if _recording() is not None:
            raise RuntimeError("a step plan is already being recorded on this thread")
        _check(int(lib().jf_plan_record_begin(self.handle)), "jf_plan_record_begin")
        _TLS.plan = self

    def abort(self):
        if _recording() is self:
            _TLS.plan = None
            lib().jf_plan_record_end(self.handle)

    def end(self):
        _TLS.plan = None
        rc = int(lib().jf_plan_record_end(self.handle))
        _check(min(rc, 0), "jf_plan_record_end")
        self._bases = (ctypes.c_void_p * max(1, self.n_slots))()
        self._ms = (ctypes.c_double * max(1, rc))()
        self.n_ops = rc
        return rc

    def num_ops(self):
        return int(lib().jf_plan_num_ops(self.handle))

    def set_any_order(self, on):
        _check(int(lib().jf_plan_set_any_order(self.handle, 1 if on else 0)), "jf_plan_set_any_order")

    def set_lane(self, lane):
        _check(int(lib().jf_plan_set_lane(self.handle, lane)), "jf_plan_set_lane")

    def fork(self):
        _check(int(lib().jf_plan_add_fork(self.handle)), "jf_plan_add_fork")

    def join(self):
        _check(int(lib().jf_plan_add_join(self.handle)), "jf_plan_add_join")

    def memset(self, t, value=0):
        _check(int(lib().jf_plan_add_memset(self.handle, _ptr(t), value, t.numel() * t.element_size())), "jf_plan_add_memset")

    def copy_to_host(self, host_t, dev_t):
        _check(int(lib().jf_plan_add_copy_to_host(self.handle, host_t.data_ptr(), _ptr(dev_t), dev_t.numel() * dev_t.element_size())), "jf_plan_add_copy_to_host")

    def launch(self, tensors, dev, stream=None):
        """re-issue the step; tensors[i] takes the place of slot i (same shape / strides as the tensor the slot was declared with); stream: the
        current stream of `dev` when the caller already holds it"""
        b = self._bases
        for i, t in enumerate(tensors):
            b[i] = t.data_ptr()
        timed = _TIMER.plan_every if _TIMER is not None else 0
        if timed != self._timing:
            lib().jf_plan_set_timing(self.handle, timed)
            self._timing = timed
        if timed:
            self._replays_under_timer = getattr(self, "_replays_under_timer", 0) + 1
            if self not in _TIMER.plans:
                _TIMER.plans.append(self)
        if torch.cuda.current_device() == dev.index:
            rc = lib().jf_plan_launch(self.handle, b, self.n_slots, (stream if stream is not None else torch.cuda.current_stream(dev)).cuda_stream)
        else:
            with torch.cuda.device(dev):
                rc = lib().jf_plan_launch(self.handle, b, self.n_slots, (stream if stream is not None else torch.cuda.current_stream(dev)).cuda_stream)
        _check(rc, "jf_plan_launch")

    def read_timing(self):
        """{(entry point, tag): (launches, summed ms)} of the timed replays since the last read (waits for their events)"""
        n = ctypes.c_int64(0)
        _check(int(lib().jf_plan_read_timing(self.handle, self._ms, self.n_ops, ctypes.byref(n), 1)), "jf_plan_read_timing")
        out = {}
        for name, tag, a, b in self.calls:
            if b > a:
                out[(name, tag)] = (int(n.value), sum(self._ms[i] for i in range(a, b)))
        # replays issued while a timer was active since the last read (every n-th of them carried events: KernelTimer.plan_every)
        self.replays_since_read, self._replays_under_timer = getattr(self, "_replays_under_timer", 0), 0
        return out


def combine_rows(ld_list, blp_list, want_total=True):
    """sums of the per-block log-dets and base log-probs in list order (one launch per JF_MAX_ROW_LISTS entries) -> (log_det, base_logp, total); a
    single entry is returned as it is, an empty list gives None"""
    ts = [t for t in list(ld_list) + list(blp_list)]
    dev = require_device(*ts)
    like = ts[0]
    B = like.shape[0]
    keep = []                                          # contiguous copies stay alive until their launch is issued

    def lst(items):
        r = jf_row_list()
        r.n = len(items)
        for i, t in enumerate(items):
            c = t.contiguous()
            keep.append(c)
            r.p[i] = _ptr(c)
        return r

    def fold(items):
        """more than JF_MAX_ROW_LISTS entries: partial sums in list order, each fed back as entry 0 of the next launch (same summation order)"""
        items = list(items)
        while len(items) > JF_MAX_ROW_LISTS:
            head, items = items[:JF_MAX_ROW_LISTS], items[JF_MAX_ROW_LISTS:]
            part = torch.empty_like(like)
            a, b = lst(head), lst([])
            _launch("jf_combine_rows" + _suffix(like), "", (ctypes.byref(a), ctypes.byref(b), B, _ptr(part), None, None), dev)
            items = [part] + items
        return items
    ld_list, blp_list = fold(ld_list), fold(blp_list)
    ld_out = torch.empty_like(like) if len(ld_list) > 1 else None
    blp_out = torch.empty_like(like) if len(blp_list) > 1 else None
    total = torch.empty_like(like) if (want_total and ld_list and blp_list) else None
    if ld_out is not None or blp_out is not None or total is not None:
        a, b = lst(ld_list), lst(blp_list)
        _launch("jf_combine_rows" + _suffix(like), "", (ctypes.byref(a), ctypes.byref(b), B, _ptr(ld_out), _ptr(blp_out), _ptr(total)), dev)
    return (ld_out if ld_out is not None else (ld_list[0] if ld_list else None),
            blp_out if blp_out is not None else (blp_list[0] if blp_list else None), total)


# ---- two side blocks of a log-prob step in one launch (csrc/merged_kernels.hip): merge_begin(); <the blocks' entry points>; merge_end(like)
JF_MERGE_DECLINED = 1
_MERGING = False


def merge_begin():
    """from here to merge_end() the launches of this thread's entry points are captured by the library instead of issued"""
    global _MERGING
    _check(int(lib().jf_merge_begin()), "jf_merge_begin")
    _MERGING = True


def merge_abort():
    global _MERGING
    if _MERGING:
        _MERGING = False
        lib().jf_merge_abort()


def merge_end(like):
    """issue the captured blocks (a broadcast g chain + an `f` block) as one launch on `like`'s device; a combination the library does not
    merge has been issued launch by launch instead: either way the results are in place.  -> True when merged"""
    global _MERGING
    dev = require_device(like)
    _MERGING = False
    fn = lib().jf_merge_end
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        rec = _recording()
        if rec is not None:
            n0 = rec.num_ops()
            rc = int(fn(stream.cuda_stream))
            rec.calls.append(("jf_merge_end", "", n0, rec.num_ops()))
        elif _TIMER is None:
            rc = int(fn(stream.cuda_stream))
        else:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rc = int(fn(stream.cuda_stream))
            e1.record(stream)
            _TIMER.records.append(("jf_merge_end", "", e0, e1))
    if rc != JF_MERGE_DECLINED:
        _check(rc, "jf_merge_end")
    return rc == JF_OK


def add_rows(a, b):
    """a + b of two (n,) tensors as a library launch (the last operation of pdf.forward; part of a recorded step)"""
    dev = require_device(a, b)
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    _launch("jf_add_rows" + _suffix(a), "", (_ptr(a), _ptr(b), a.numel(), _ptr(out)), dev)
    return out


def gf_layer_array(structs):
    arr = (jf_gf_layer * len(structs))()
    for i, s in enumerate(structs):
        arr[i] = s
    return arr


# When set to a list, every spline-carrying chain launch appends its (B, n_searches) int64 tensor of raw searchsorted results
# (execution order, -3 = not written, -2 = row skipped by an identity region): the integer output of the bit-exact parity tests.
BINS_LOG = None


# sampling with broadcast parameters: from this many rows on the solves start from a table of the layer's inverse functions (the table costs
# 1025 solves per layer and coordinate: ~1000 rows' worth of a four-layer chain); JF_FWD_TABLE_MIN_ROWS overrides, 0 = never
FWD_TABLE_MIN_ROWS = int(os.environ.get("JF_FWD_TABLE_MIN_ROWS", "8192")) or (1 << 62)


def gf_chain(direction, x, log_det, params, layer_array, n_layers, D, x_out=None, base_logp_in=None, want_base_logp=False, status=None,
             want_total=False):
    """run a chain of g layers.  direction 'inv' (log-prob) or 'fwd' (sampling).
    x (B, D) view (row stride arbitrary), log_det (B,) or None, params (1|B, P).  Returns (x_out, log_det_out[, base_logp[, total]]);
    want_total (with want_base_logp, 'inv'): the launch also writes total = base_logp + log_det (jf_gf_chain_inv_total)."""
    dev = require_device(x, log_det, params, x_out, base_logp_in, status)
    x = _rowmajor(x)
    params = _rowmajor(params)
    if params.dtype != x.dtype:
        raise TypeError("parameter dtype %s != input dtype %s" % (params.dtype, x.dtype))
    B = x.shape[0]
    if x.shape[1] != D:
        raise ValueError("expected %d target columns, got %d" % (D, x.shape[1]))
    if params.shape[0] not in (1, B):
        raise ValueError("extra_inputs must have 1 or B=%d rows, got %d" % (B, params.shape[0]))
    if log_det is not None:
        log_det = log_det.contiguous()
        if log_det.dtype != x.dtype or log_det.shape[0] != B:
            raise ValueError("log_det must be a (B,) tensor of the input dtype")
    if x_out is None:
        x_out = torch.empty((B, D), dtype=x.dtype, device=x.device)
    ld_out = torch.empty((B,), dtype=x.dtype, device=x.device)
    suf = _suffix(x)
    pb = 1 if (params.shape[0] == 1 and B != 1) else params.shape[0]
    if B == 1:
        pb = 1
    bins = None
    if BINS_LOG is not None:
        n_spl = sum(1 for i in range(n_layers) if layer_array[i].nonlinear_stretch_type == GF_STRETCH_RQ_SPLINES)
        if n_spl:
            bins = torch.full((B, n_spl * D), -3, dtype=torch.int64, device=x.device)
            BINS_LOG.append(bins)
    bs = bins.stride(0) if bins is not None else 0
    if direction == "inv" and want_total and want_base_logp:
        blp_out = torch.empty((B,), dtype=x.dtype, device=x.device)
        total = torch.empty((B,), dtype=x.dtype, device=x.device)
        _launch("jf_gf_chain_inv_total" + suf, "bcast" if pb == 1 else "per-sample",
                (_ptr(x), x.stride(0), _ptr(log_det), _ptr(params), params.stride(0), pb, B, D, n_layers, layer_array, _ptr(x_out),
                 x_out.stride(0), _ptr(ld_out), _ptr(base_logp_in), _ptr(blp_out), _ptr(total), _ptr(bins), bs, _ptr(status)), dev)
        return x_out, ld_out, blp_out, total
    if direction == "inv":
        blp_out = torch.empty((B,), dtype=x.dtype, device=x.device) if want_base_logp else None
        _launch("jf_gf_chain_inv" + suf, "bcast" if pb == 1 else "per-sample",
                (_ptr(x), x.stride(0), _ptr(log_det), _ptr(params), params.stride(0), pb, B, D, n_layers, layer_array, _ptr(x_out),
                 x_out.stride(0), _ptr(ld_out), _ptr(base_logp_in), _ptr(blp_out), _ptr(bins), bs, _ptr(status)), dev)
        return (x_out, ld_out, blp_out) if want_base_logp else (x_out, ld_out)
    if pb == 1 and B >= FWD_TABLE_MIN_ROWS:
        # row-independent parameters: every (layer, coordinate) inverts one fixed function for all rows -- the library tabulates it first and
        # the solves start from the interpolated value (jf_gf_chain_fwd_tab); the table is a temporary of this call
        table = torch.empty((int(lib().jf_gf_chain_fwd_table_elems(D, n_layers)),), dtype=x.dtype, device=x.device)
        _launch("jf_gf_chain_fwd_tab" + suf, "bcast",
                (_ptr(x), x.stride(0), _ptr(log_det), _ptr(params), params.stride(0), pb, B, D, n_layers, layer_array, _ptr(x_out), x_out.stride(0),
                 _ptr(ld_out), _ptr(bins), bs, _ptr(status), _ptr(table)), dev)
        return x_out, ld_out
    _launch("jf_gf_chain_fwd" + suf, "bcast" if pb == 1 else "per-sample",
            (_ptr(x), x.stride(0), _ptr(log_det), _ptr(params), params.stride(0), pb, B, D, n_layers, layer_array, _ptr(x_out), x_out.stride(0),
             _ptr(ld_out), _ptr(bins), bs, _ptr(status)), dev)
    return x_out, ld_out


def gf_chain_inv_cot(x, params, layer_array, n_layers, D, cot):
    """J^{-T} cot for the Jacobian J of gf_chain('inv', x, ...) at x (the co-vector carried through the layers: jf_gf_chain_inv_cot); None for layers
    with the general options (the caller then solves with a dense Jacobian)"""
    dev = require_device(x, params, cot)
    x, params, cot = _rowmajor(x), _rowmajor(params), _rowmajor(cot)
    B = x.shape[0]
    if params.dtype != x.dtype or cot.dtype != x.dtype or cot.shape != (B, D) or x.shape != (B, D) or params.shape[0] not in (1, B):
        raise ValueError("gf_chain_inv_cot: inconsistent shapes / dtypes")
    out = torch.empty((B, D), dtype=x.dtype, device=x.device)
    if B == 0:
        return out
    x_out = torch.empty((B, D), dtype=x.dtype, device=x.device)
    ld_out = torch.empty((B,), dtype=x.dtype, device=x.device)
    ok = _launch("jf_gf_chain_inv_cot" + _suffix(x), "bcast" if params.shape[0] == 1 else "per-sample",
                 (_ptr(x), x.stride(0), _ptr(params), params.stride(0), 1 if params.shape[0] == 1 else B, B, D, n_layers, layer_array, _ptr(cot),
                  cot.stride(0), _ptr(out), out.stride(0), _ptr(x_out), x_out.stride(0), _ptr(ld_out)), dev, unsupported_ok=True)
    return out if ok is not False else None


def gf_chain_inv_bwd(x, params, layer_array, n_layers, D, g_xout, g_ld, g_blp, status=None):
    """vector-Jacobian product of gf_chain('inv', ...): upstream gradients of (x_out, log_det_out, base_logp_out) -> (g_x (B, D), g_params).
    g_params has the shape of `params`: (B, P) per-sample, (1, P) for permanent parameters (partial sums of the launch added up here)."""
    dev = require_device(x, params, g_xout, g_ld, g_blp, status)
    x, params = _rowmajor(x), _rowmajor(params)
    B = x.shape[0]
    pb = 1 if (params.shape[0] == 1) else B
    if params.shape[0] not in (1, B):
        raise ValueError("params must have 1 or B rows")
    if g_xout is not None:
        g_xout = _rowmajor(g_xout)
    g_ld = None if g_ld is None else g_ld.contiguous()
    g_blp = None if g_blp is None else g_blp.contiguous()
    g_x = torch.empty((B, D), dtype=x.dtype, device=x.device)
    P = params.shape[1]
    if pb == 1:
        n_part = int(lib().jf_gf_chain_inv_bwd_partials(B, D))
        g_p = (torch.zeros if B == 0 else torch.empty)((n_part, P), dtype=x.dtype, device=x.device)     # (no rows: no launch writes the partials)
    else:
        g_p = torch.empty((B, P), dtype=x.dtype, device=x.device)
    _launch("jf_gf_chain_inv_bwd" + _suffix(x), "bcast" if pb == 1 else "per-sample",
            (_ptr(x), x.stride(0), _ptr(params), params.stride(0), pb, B, D, n_layers, layer_array, _ptr(g_xout),
             g_xout.stride(0) if g_xout is not None else 0, _ptr(g_ld), _ptr(g_blp), _ptr(g_x), g_x.stride(0), _ptr(g_p), g_p.stride(0),
             _ptr(status)), dev)
    if pb == 1:
        g_p = slab_sum(g_p)[0].unsqueeze(0)
    return g_x, g_p


COND_GF_MAX_IN, COND_GF_MAX_HIDDEN = 28, 128
GF_MAX_DIM = 64              # 'g' layers: coordinates per row (groups of up to 64 lanes = a whole wave per row; fused blocks: 8)
T_MAX_DIM = 32               # 't' layers: the triangular factor of a row in registers
LDS_BYTES_PER_CU = 160 * 1024


def gf_chain_fits(layer_array, n_layers, D, dtype, bcast, backward=False):
    """does ONE launch of this chain fit the LDS of a CU (log-prob / sampling direction, or its backward)?"""
    suf = "_f32" if dtype == torch.float32 else "_f64"
    fn = getattr(lib(), ("jf_gf_chain_inv_bwd_lds_bytes" if backward else "jf_gf_chain_lds_bytes") + suf)
    n = int(fn(D, n_layers, layer_array, 1 if bcast else 0))
    return 0 <= n <= LDS_BYTES_PER_CU


def cond_gf_chain_inv(inp, w1, b1, w2, b2, x, log_det, layer_array, n_layers, D, x_out=None, base_logp_in=None, want_base_logp=False, status=None):
    """amortisation MLP (Linear-tanh-Linear) + the chain of g layers it parametrises in ONE launch; the parameter block stays on chip."""
    dev = require_device(inp, w1, b1, w2, b2, x, log_det, x_out, base_logp_in, status)
    inp, w1, w2, x = _rowmajor(inp), _rowmajor(w1), _rowmajor(w2), _rowmajor(x)
    B, K1 = inp.shape
    H = w1.shape[0]
    if x.shape[0] != B or x.shape[1] != D or w1.shape[1] != K1 or w2.shape[1] != H or b1.shape[0] != H or b2.shape[0] != w2.shape[0]:
        raise ValueError("cond_gf_chain_inv: inconsistent shapes")
    if any(t.dtype != x.dtype for t in (inp, w1, b1, w2, b2)):
        raise TypeError("cond_gf_chain_inv: dtype mismatch")
    if log_det is not None:
        log_det = log_det.contiguous()
    if x_out is None:
        x_out = torch.empty((B, D), dtype=x.dtype, device=x.device)
    ld_out = torch.empty((B,), dtype=x.dtype, device=x.device)
    blp_out = torch.empty((B,), dtype=x.dtype, device=x.device) if want_base_logp else None
    _launch("jf_cond_gf_chain_inv" + _suffix(x), "K%d_H%d_N%d_D%d" % (K1, H, w2.shape[0], D),
            (_ptr(inp), inp.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(w2), w2.stride(0), _ptr(b2.contiguous()), K1, H,
             _ptr(x), x.stride(0), _ptr(log_det), B, D, n_layers, layer_array, _ptr(x_out), x_out.stride(0), _ptr(ld_out), _ptr(base_logp_in),
             _ptr(blp_out), _ptr(status)), dev)
    return (x_out, ld_out, blp_out) if want_base_logp else (x_out, ld_out)


def amlp_gf_chain_inv(inp, v1, u1, b1, v2, u2, b2, x, log_det, layer_array, n_layers, D, x_out=None, base_logp_in=None, want_base_logp=False,
                      status=None):
    """two-stage low-rank AmortizableMLP (v1 None: full first stage u1 (H, K1)) + the chain of g layers it parametrises in ONE launch; the
    parameter block is regenerated from the rank-space vector of each row on chip (see include/jammy_hip.h)."""
    dev = require_device(inp, v1, u1, b1, v2, u2, b2, x, log_det, x_out, base_logp_in, status)
    inp, x = _rowmajor(inp), _rowmajor(x)
    B, K1 = inp.shape
    H, r2 = v2.shape[1], v2.shape[0]
    r1 = 0 if v1 is None else v1.shape[0]
    ws = [t.contiguous() for t in (u1, b1, v2, u2, b2)]
    v1c = None if v1 is None else v1.contiguous()
    if any(t.dtype != x.dtype for t in ws + [inp]) or x.shape != (B, D) or u2.shape[1] != r2 or b2.shape[0] != u2.shape[0]:
        raise ValueError("amlp_gf_chain_inv: inconsistent shapes / dtypes")
    if log_det is not None:
        log_det = log_det.contiguous()
    if x_out is None:
        x_out = torch.empty((B, D), dtype=x.dtype, device=x.device)
    ld_out = torch.empty((B,), dtype=x.dtype, device=x.device)
    blp_out = torch.empty((B,), dtype=x.dtype, device=x.device) if want_base_logp else None
    _launch("jf_amlp_gf_chain_inv" + _suffix(x), "K%d_H%d_N%d_D%d_r%d" % (K1, H, u2.shape[0], D, r2),
            (_ptr(inp), inp.stride(0), _ptr(v1c), _ptr(ws[0]), _ptr(ws[1]), _ptr(ws[2]), _ptr(ws[3]), _ptr(ws[4]), K1, H, r1, r2, _ptr(x), x.stride(0),
             _ptr(log_det), B, D, n_layers, layer_array, _ptr(x_out), x_out.stride(0), _ptr(ld_out), _ptr(base_logp_in), _ptr(blp_out), _ptr(status)),
            dev)
    return (x_out, ld_out, blp_out) if want_base_logp else (x_out, ld_out)


LOWRANK_GF_MAX_RANK = 8
LOWRANK_GF_AUX = 5


def lowrank_head_ok(inp, v1, u1, b1, v2):
    """shapes jf_lowrank_head_f64 takes: float64, low-rank first stage, K1 <= 32, hidden <= 128 and a multiple of 16, ranks <= 8"""
    H = u1.shape[0]
    return (inp.dtype == torch.float64 and v1 is not None and b1 is not None and inp.shape[1] <= 32 and 16 <= H <= 128 and H % 16 == 0
            and v1.shape[0] <= LOWRANK_GF_MAX_RANK and v2.shape[0] <= LOWRANK_GF_MAX_RANK and inp.shape[0] > 0)


def lowrank_head(inp, v1, u1, b1, v2):
    """t1 = v1 c, h = tanh(u1 t1 + b1), t2 = v2 h in one launch -> (t2 (B, r2), t1 (B, 8), h (B, H)); None when the shapes are not supported"""
    dev = require_device(inp, v1, u1, b1, v2)
    inp = _rowmajor(inp)
    v1, u1, b1, v2 = v1.contiguous(), u1.contiguous(), b1.contiguous(), v2.contiguous()
    B, K1 = inp.shape
    H, r1, r2 = u1.shape[0], v1.shape[0], v2.shape[0]
    t1 = torch.empty((B, LOWRANK_GF_MAX_RANK), dtype=inp.dtype, device=inp.device)
    h = torch.empty((B, H), dtype=inp.dtype, device=inp.device)
    t2 = torch.empty((B, LOWRANK_GF_MAX_RANK), dtype=inp.dtype, device=inp.device)
    ok = _launch("jf_lowrank_head_f64", "K%d_H%d_r%d_r%d" % (K1, H, r1, r2),
                 (_ptr(inp), inp.stride(0), _ptr(v1), _ptr(u1), _ptr(b1), _ptr(v2), B, K1, H, r1, r2, _ptr(t1), _ptr(h), _ptr(t2)), dev, unsupported_ok=True)
    return None if ok is False else (t2[:, :r2], t1, h)


def lowrank_head_bwd(inp, v1, u1, v2, t1, h, g_t2, want_input_grad):
    """adjoint of lowrank_head in one launch + one reduction -> (g_inp or None, g_v1, g_u1, g_b1, g_v2)"""
    dev = require_device(inp, v1, u1, v2, t1, h, g_t2)
    inp, g_t2 = _rowmajor(inp), _rowmajor(g_t2)
    v1, u1, v2 = v1.contiguous(), u1.contiguous(), v2.contiguous()
    B, K1 = inp.shape
    H, r1, r2 = u1.shape[0], v1.shape[0], v2.shape[0]
    g_inp = torch.empty((B, K1), dtype=inp.dtype, device=inp.device) if want_input_grad else None
    g_v1, g_u1, g_v2 = torch.empty_like(v1), torch.empty_like(u1), torch.empty_like(v2)
    g_b1 = torch.empty((H,), dtype=inp.dtype, device=inp.device)
    work = torch.empty((int(lib().jf_lowrank_head_workspace_doubles(B, K1, H)),), dtype=inp.dtype, device=inp.device)
    _launch("jf_lowrank_head_bwd_f64", "K%d_H%d_r%d_r%d" % (K1, H, r1, r2),
            (_ptr(inp), inp.stride(0), _ptr(v1), _ptr(u1), _ptr(v2), B, K1, H, r1, r2, _ptr(t1), _ptr(h), _ptr(g_t2), g_t2.stride(0), _ptr(g_inp),
             K1, _ptr(g_v1), _ptr(g_u1), _ptr(g_b1), _ptr(g_v2), _ptr(work)), dev)
    return g_inp, g_v1, g_u1, g_b1, g_v2


def lowrank_gf_chain_inv(t2, u2, b2, x, log_det, layer_array, n_layers, D, base_logp_in=None, want_base_logp=False, want_aux=False, status=None):
    """log-prob direction of a chain of g layers whose parameter rows are u2 t2[row] + b2 (the low-rank last stage of an AmortizableMLP,
    float64, rank <= 8), the block never materialised: jf_lowrank_gf_chain_inv_f64.  want_aux: also the (n_layers, 5, B, 8) block of layer
    inputs and mixture sums its backward launch reads.  -> (x_out, log_det_out[, base_logp_out][, aux]) or None when unsupported."""
    dev = require_device(t2, u2, b2, x, log_det, base_logp_in, status)
    t2, x = _rowmajor(t2), _rowmajor(x)
    u2, b2 = u2.contiguous(), b2.contiguous()
    B, r2 = t2.shape
    if x.dtype != torch.float64 or any(t.dtype != torch.float64 for t in (t2, u2, b2)) or x.shape != (B, D) or u2.shape[1] != r2 or b2.shape[0] != u2.shape[0]:
        raise ValueError("lowrank_gf_chain_inv: inconsistent shapes / dtypes")
    if log_det is not None:
        log_det = log_det.contiguous()
    x_out = torch.empty((B, D), dtype=x.dtype, device=x.device)
    ld_out = torch.empty((B,), dtype=x.dtype, device=x.device)
    blp_out = torch.empty((B,), dtype=x.dtype, device=x.device) if want_base_logp else None
    aux = torch.empty((n_layers, LOWRANK_GF_AUX, 2, B, 4), dtype=x.dtype, device=x.device) if want_aux else None
    ok = _launch("jf_lowrank_gf_chain_inv_f64", "N%d_D%d_r%d" % (u2.shape[0], D, r2),
                 (_ptr(t2), t2.stride(0), _ptr(u2), _ptr(b2), r2, _ptr(x), x.stride(0), _ptr(log_det), B, D, n_layers, layer_array, _ptr(x_out),
                  x_out.stride(0), _ptr(ld_out), _ptr(base_logp_in), _ptr(blp_out), _ptr(aux), _ptr(status)), dev, unsupported_ok=True)
    if ok is False:
        return None
    return (x_out, ld_out) + ((blp_out,) if want_base_logp else ()) + ((aux,) if want_aux else ())


def lowrank_gf_chain_inv_bwd(t2, u2, b2, aux, x_out, layer_array, n_layers, D, g_xout, g_ld, g_blp, status=None):
    """adjoint of lowrank_gf_chain_inv: one launch per layer + one reduction -> (g_x (B, D), g_t2 (B, r2), g_u2 (N, r2), g_b2 (N))"""
    dev = require_device(t2, u2, b2, aux, x_out, g_xout, g_ld, g_blp)
    t2, x_out = _rowmajor(t2), _rowmajor(x_out)
    u2, b2 = u2.contiguous(), b2.contiguous()
    B, r2 = t2.shape
    N = u2.shape[0]
    g_xout = None if g_xout is None else _rowmajor(g_xout)
    g_ld = None if g_ld is None else g_ld.contiguous()
    g_blp = None if g_blp is None else g_blp.contiguous()
    g_x = torch.empty((B, D), dtype=t2.dtype, device=t2.device)
    g_t2 = torch.empty((B, LOWRANK_GF_MAX_RANK), dtype=t2.dtype, device=t2.device)
    g_u2 = torch.empty((N, r2), dtype=t2.dtype, device=t2.device)
    g_b2 = torch.empty((N,), dtype=t2.dtype, device=t2.device)
    if B == 0:
        return g_x, g_t2[:, :r2], g_u2.zero_(), g_b2.zero_()
    work = torch.empty((int(lib().jf_lowrank_gf_workspace_doubles(B, n_layers)),), dtype=t2.dtype, device=t2.device)
    _launch("jf_lowrank_gf_chain_inv_bwd_f64", "N%d_D%d_r%d" % (N, D, r2),
            (_ptr(t2), t2.stride(0), _ptr(u2), _ptr(b2), r2, _ptr(aux), _ptr(x_out), x_out.stride(0), B, D, n_layers, layer_array, _ptr(g_xout),
             0 if g_xout is None else g_xout.stride(0), _ptr(g_ld), _ptr(g_blp), _ptr(g_x), g_x.stride(0), _ptr(g_t2), _ptr(g_u2), _ptr(g_b2),
             _ptr(work), _ptr(status)), dev)
    return g_x, g_t2[:, :r2], g_u2, g_b2


def amlp_gf_chain_fwd(inp, v1, u1, b1, v2, u2, b2, z, log_det, layer_array, n_layers, D, x_out=None, status=None):
    """sampling direction of the low-rank block in one launch (float64, ranks <= 8: jf_amlp_gf_chain_fwd_f64) -> (x, log_det), or None when
    the configuration is outside the matrix-core kernel's set (the caller then runs amlp2 + gf_chain)"""
    dev = require_device(inp, v1, u1, b1, v2, u2, b2, z, log_det, x_out, status)
    if z.dtype != torch.float64:
        return None
    inp, z = _rowmajor(inp), _rowmajor(z)
    B, K1 = inp.shape
    H, r2 = v2.shape[1], v2.shape[0]
    r1 = 0 if v1 is None else v1.shape[0]
    ws = [t.contiguous() for t in (u1, b1, v2, u2, b2)]
    v1c = None if v1 is None else v1.contiguous()
    if any(t.dtype != z.dtype for t in ws + [inp]) or z.shape != (B, D) or u2.shape[1] != r2 or b2.shape[0] != u2.shape[0]:
        raise ValueError("amlp_gf_chain_fwd: inconsistent shapes / dtypes")
    if log_det is not None:
        log_det = log_det.contiguous()
    if x_out is None:
        x_out = torch.empty((B, D), dtype=z.dtype, device=z.device)
    ld_out = torch.empty((B,), dtype=z.dtype, device=z.device)
    ok = _launch("jf_amlp_gf_chain_fwd_f64", "K%d_H%d_N%d_D%d_r%d" % (K1, H, u2.shape[0], D, r2),
                 (_ptr(inp), inp.stride(0), _ptr(v1c), _ptr(ws[0]), _ptr(ws[1]), _ptr(ws[2]), _ptr(ws[3]), _ptr(ws[4]), K1, H, r1, r2, _ptr(z), z.stride(0),
                  _ptr(log_det), B, D, n_layers, layer_array, _ptr(x_out), x_out.stride(0), _ptr(ld_out), _ptr(status)), dev, unsupported_ok=True)
    return (x_out, ld_out) if ok else None


def amlp2(inp, v1, u1, b1, v2, u2, b2):
    """two-stage low-rank AmortizableMLP with permanent weights in one launch: (B, N) = u2 (v2 tanh(W1 inp + b1)) + b2, W1 = u1 v1 or u1 (v1 None)"""
    dev = require_device(inp, v1, u1, b1, v2, u2, b2)
    inp = _rowmajor(inp)
    B, K1 = inp.shape
    H, r2, N = v2.shape[1], v2.shape[0], u2.shape[0]
    r1 = 0 if v1 is None else v1.shape[0]
    ws = [t.contiguous() for t in (u1, b1, v2, u2, b2)]
    v1c = None if v1 is None else v1.contiguous()
    out = torch.empty((B, N), dtype=inp.dtype, device=inp.device)
    _launch("jf_amlp2" + _suffix(inp), "K%d_H%d_N%d_r%d" % (K1, H, N, r2),
            (_ptr(inp), inp.stride(0), _ptr(v1c), _ptr(ws[0]), _ptr(ws[1]), _ptr(ws[2]), _ptr(ws[3]), _ptr(ws[4]), B, K1, H, r1, r2, N, _ptr(out),
             out.stride(0)), dev)
    return out


# the register-resident fused block kernel (cond_split_kernels.hip, 64 / 128 rows per workgroup) in its two matrix arithmetics
_COND_GF_PACK = {"split": ("jf_cond_gf_packed_bytes", "jf_cond_gf_pack_f32", "jf_cond_gf_chain_inv_split_f32", 28),
                 "split16": ("jf_cond_gf_packed_bytes2", "jf_cond_gf_pack2_f32", "jf_cond_gf_chain_split2_f32", 28)}
SPLIT_BF16X3, SPLIT_F16X2 = 0, 1     # include/jammy_hip.h: JF_SPLIT_*; kind "split" = bf16 triple, "split16" = f16 pair (same kernels)
DIR_INV, DIR_FWD = 0, 1


def cond_gf_packed_bytes(layer_array, n_layers, D, kind="split"):
    """size of the packed W2 / b2 image of a register-resident fused block kernel, or a negative JF_ERR_* when the chain is not supported by it."""
    if kind == "split16":
        return int(lib().jf_cond_gf_packed_bytes2(D, n_layers, layer_array, SPLIT_F16X2))
    return int(getattr(lib(), _COND_GF_PACK[kind][0])(D, n_layers, layer_array))


def cond_gf_pack(w2, b2, layer_array, n_layers, D, kind="split"):
    """W2 (N, H) / b2 (N,) of the amortisation MLP -> packed image for cond_gf_chain_inv_split (bf16 pieces in MFMA fragment order,
    rows permuted so that the MFMA result registers are the flow's parameter registers).  Redo whenever the weights change."""
    dev = require_device(w2, b2)
    w2 = _rowmajor(w2)
    if w2.dtype != torch.float32 or b2.dtype != torch.float32:
        raise TypeError("cond_gf_pack: float32 only")
    nbytes = cond_gf_packed_bytes(layer_array, n_layers, D, kind)
    _check(min(nbytes, 0), _COND_GF_PACK[kind][0])
    packed = torch.empty((nbytes,), dtype=torch.uint8, device=w2.device)
    if kind == "split16":
        _launch("jf_cond_gf_pack2_f32", "f16x2", (_ptr(w2), w2.stride(0), _ptr(b2.contiguous()), w2.shape[1], D, n_layers, layer_array, SPLIT_F16X2,
                                                  _ptr(packed)), dev)
        return packed
    _launch(_COND_GF_PACK[kind][1], "", (_ptr(w2), w2.stride(0), _ptr(b2.contiguous()), w2.shape[1], D, n_layers, layer_array, _ptr(packed)), dev)
    return packed


COND_GF_MAX_PRE = 4                  # entries of pre_ld / pre_blp (csrc/cond_split_kernels.hip: CS_MAX_PRE)


def cond_gf_chain_inv_split(inp, w1, b1, packed, x, log_det, layer_array, n_layers, D, x_out=None, base_logp_in=None, want_base_logp=False,
                            status=None, kind="split", aux=None, pre_ld=None, pre_blp=None):
    """as cond_gf_chain_inv with the second product on split-bf16 MFMA and the parameter block in registers (float32, default layer options);
    `kind` selects the kernel the packed image was built for.  aux (cond_gf_aux, "split" only): the launch also leaves what
    cond_gf_chain_inv_split_bwd starts from.
    pre_ld / pre_blp (lists of (B,) tensors, with want_base_logp): this block is the last of its pdf -- the launch also adds the earlier blocks'
    log-dets / base log-probs in front of its own (list order) and returns (x_out, log_det total, base_logp total, total log-prob): the sums of
    combine_rows without its launch.  Returns None when the segment-input kernel that does this is not the one in use."""
    seg = None
    if pre_ld is not None or pre_blp is not None:
        if not (want_base_logp and kind == "split16" and aux is None and isinstance(inp, SegInput) and inp.in_place_ok and inp.dtype == torch.float32
                and len(pre_ld or []) <= COND_GF_MAX_PRE and len(pre_blp or []) <= COND_GF_MAX_PRE):
            return None
    if isinstance(inp, SegInput):
        if kind == "split16" and inp.in_place_ok and inp.dtype == torch.float32:
            seg = inp                                      # read in place by jf_cond_gf_chain_split3_f32
            inp = seg.segments[0][0]
        else:
            inp = inp.materialize()
    dev = require_device(inp, w1, b1, packed, x, log_det, x_out, base_logp_in, status, *(seg.tensors() if seg else []))
    inp, w1, x = _rowmajor(inp), _rowmajor(w1), _rowmajor(x)
    B, K1 = seg.shape if seg else inp.shape
    H = w1.shape[0]
    if x.shape[0] != B or x.shape[1] != D or w1.shape[1] != K1 or b1.shape[0] != H:
        raise ValueError("cond_gf_chain_inv_split: inconsistent shapes")
    if any(t.dtype != torch.float32 for t in (inp, w1, b1, x)) or packed.dtype != torch.uint8:
        raise TypeError("cond_gf_chain_inv_split: float32 inputs and a uint8 packed image expected")
    if log_det is not None:
        log_det = log_det.contiguous()
    if x_out is None:
        x_out = torch.empty((B, D), dtype=x.dtype, device=x.device)
    ld_out = torch.empty((B,), dtype=x.dtype, device=x.device)
    blp_out = torch.empty((B,), dtype=x.dtype, device=x.device) if want_base_logp else None
    if kind == "split16":
        if aux is not None and (aux.dtype != torch.float32 or aux.numel() < n_layers * B * 20 or not aux.is_contiguous()):
            raise ValueError("cond_gf_chain_inv_split: aux = cond_gf_aux(B, n_layers)")
        if seg is not None:
            arr = seg.c_array()
            lists, total = (None, None), None
            if pre_ld is not None or pre_blp is not None:
                def lst(items):
                    r = jf_row_list()
                    r.n = len(items)
                    for i, t in enumerate(items):
                        if t.dtype != torch.float32 or t.shape != (B,) or not t.is_contiguous():
                            raise ValueError("cond_gf_chain_inv_split: pre_ld / pre_blp entries are contiguous float32 (B,) tensors")
                        r.p[i] = _ptr(t)
                    return r
                require_device(x, *(list(pre_ld or []) + list(pre_blp or [])))
                lists = (lst(list(pre_ld or [])), lst(list(pre_blp or [])))
                total = torch.empty((B,), dtype=x.dtype, device=x.device)
            _launch("jf_cond_gf_chain_split3_f32", "K%d_H%d_L%d_D%d" % (K1, H, n_layers, D),
                    (DIR_INV, SPLIT_F16X2, arr, len(seg.segments), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(packed), K1, H, _ptr(x), x.stride(0),
                     _ptr(log_det), B, D, n_layers, layer_array, _ptr(x_out), x_out.stride(0), _ptr(ld_out), _ptr(base_logp_in), _ptr(blp_out), _ptr(aux),
                     None if lists[0] is None else ctypes.byref(lists[0]), None if lists[1] is None else ctypes.byref(lists[1]), _ptr(total),
                     _ptr(status)), dev)
            if total is not None:
                return x_out, ld_out, blp_out, total
            return (x_out, ld_out, blp_out) if want_base_logp else (x_out, ld_out)
        _launch("jf_cond_gf_chain_split2_f32", "K%d_H%d_L%d_D%d" % (K1, H, n_layers, D),
                (DIR_INV, SPLIT_F16X2, _ptr(inp), inp.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(packed), K1, H, _ptr(x), x.stride(0),
                 _ptr(log_det), B, D, n_layers, layer_array, _ptr(x_out), x_out.stride(0), _ptr(ld_out), _ptr(base_logp_in), _ptr(blp_out), _ptr(aux),
                 _ptr(status)), dev)
        return (x_out, ld_out, blp_out) if want_base_logp else (x_out, ld_out)
    if aux is not None:
        if kind != "split" or aux.dtype != torch.float32 or aux.numel() < n_layers * B * 20 or not aux.is_contiguous():
            raise ValueError("cond_gf_chain_inv_split: aux needs the 'split' kernel and cond_gf_aux(B, n_layers)")
        _launch("jf_cond_gf_chain_inv_split_save_f32", "K%d_H%d_L%d_D%d" % (K1, H, n_layers, D),
                (_ptr(inp), inp.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(packed), K1, H, _ptr(x), x.stride(0), _ptr(log_det),
                 B, D, n_layers, layer_array, _ptr(x_out), x_out.stride(0), _ptr(ld_out), _ptr(base_logp_in), _ptr(blp_out), _ptr(aux),
                 _ptr(status)), dev)
        return (x_out, ld_out, blp_out) if want_base_logp else (x_out, ld_out)
    _launch(_COND_GF_PACK[kind][2], "K%d_H%d_L%d_D%d" % (K1, H, n_layers, D),
            (_ptr(inp), inp.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(packed), K1, H, _ptr(x), x.stride(0), _ptr(log_det), B, D,
             n_layers, layer_array, _ptr(x_out), x_out.stride(0), _ptr(ld_out), _ptr(base_logp_in), _ptr(blp_out), _ptr(status)), dev)
    return (x_out, ld_out, blp_out) if want_base_logp else (x_out, ld_out)


COND_GF_SLOTS = 36                   # parameter slots per coordinate lane and layer in the packed gradient rows (csrc/jf_cond_regs.h)


def cond_gf_aux(B, n_layers, device):
    """buffer the gradient-mode forward launch of the fused block fills: per (layer, row, coordinate lane) the layer's input coordinate and
    its mixture sums"""
    return torch.empty((int(lib().jf_cond_gf_aux_floats(B, n_layers)),), dtype=torch.float32, device=device)


def cond_gf_bwd_pack(w2, layer_array, n_layers, D):
    """W2 (N, H) -> W2^T as MFMA fragments for cond_gf_chain_inv_split_bwd's g_h product.  Redo whenever the weights change."""
    dev = require_device(w2)
    w2 = _rowmajor(w2)
    if w2.dtype != torch.float32:
        raise TypeError("cond_gf_bwd_pack: float32 only")
    nbytes = int(lib().jf_cond_gf_bwd_packed_bytes(D, n_layers, layer_array))
    _check(min(nbytes, 0), "jf_cond_gf_bwd_packed_bytes")
    packed = torch.empty((nbytes,), dtype=torch.uint8, device=w2.device)
    _launch("jf_cond_gf_bwd_pack_f32", "", (_ptr(w2), w2.stride(0), w2.shape[1], D, n_layers, layer_array, _ptr(packed)), dev)
    return packed


def cond_gf_packed_rows(layer_array, n_layers, D):
    """index (list of ints, one per column of the natural parameter row) into the packed gradient row [layer][coordinate lane][slot] that
    cond_gf_chain_inv_split_bwd writes: natural order per layer = offset (D, if modelled), Householder vectors (hh_iter x D), means, log-widths,
    log-weights (num_kde x D each) -- gaussianization_flow.py:63-215; slots = csrc/jf_cond_regs.h."""
    idx = []
    for l in range(n_layers):
        o = layer_array[l]
        K = o.num_kde
        base = l * 4 * COND_GF_SLOTS
        if o.model_offset:
            idx += [base + d * COND_GF_SLOTS + 3 * K + 4 for d in range(D)]
        for i in range(o.hh_iter):
            idx += [base + d * COND_GF_SLOTS + 3 * K + i for d in range(D)]
        for sec in range(3):
            for k in range(K):
                idx += [base + d * COND_GF_SLOTS + sec * K + k for d in range(D)]
    return idx


def cond_gf_chain_inv_split_bwd(inp, w1, b1, packed, packed_t, z, aux, layer_array, n_layers, D, g_xout, g_ld, g_blp, want_absmax=False):
    """adjoint of cond_gf_chain_inv_split(..., aux=aux) in one launch -> (g_x (B, D), g_pp (B, n_layers * 144) packed parameter-row gradient,
    h (B, H) hidden activations, g_h (B, H)[, max |g_pp| as a 1-element tensor]); see include/jammy_hip.h."""
    dev = require_device(inp, w1, b1, packed, packed_t, z, aux, g_xout, g_ld, g_blp)
    inp, w1, z = _rowmajor(inp), _rowmajor(w1), _rowmajor(z)
    B, K1 = inp.shape
    H = w1.shape[0]
    if z.shape[0] != B or z.shape[1] != D or w1.shape[1] != K1 or b1.shape[0] != H or aux.numel() < n_layers * B * 20:
        raise ValueError("cond_gf_chain_inv_split_bwd: inconsistent shapes")
    if any(t is not None and t.dtype != torch.float32 for t in (inp, w1, b1, z, aux, g_xout, g_ld, g_blp)):
        raise TypeError("cond_gf_chain_inv_split_bwd: float32 only")
    g_xout = None if g_xout is None else _rowmajor(g_xout)
    g_ld = None if g_ld is None else g_ld.contiguous()
    g_blp = None if g_blp is None else g_blp.contiguous()
    g_x = torch.empty((B, D), dtype=z.dtype, device=z.device)
    g_pp = torch.empty((B, n_layers * 4 * COND_GF_SLOTS), dtype=z.dtype, device=z.device)
    h = torch.empty((B, H), dtype=z.dtype, device=z.device)
    g_h = torch.empty((B, H), dtype=z.dtype, device=z.device)
    absmax = torch.zeros((1,), dtype=z.dtype, device=z.device) if want_absmax else None
    _launch("jf_cond_gf_chain_inv_split_bwd_f32", "K%d_H%d_L%d_D%d" % (K1, H, n_layers, D),
            (_ptr(inp), inp.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(packed), _ptr(packed_t), K1, H, _ptr(z), z.stride(0),
             _ptr(aux), B, D, n_layers, layer_array, _ptr(g_xout), 0 if g_xout is None else g_xout.stride(0), _ptr(g_ld), _ptr(g_blp),
             _ptr(g_x), g_x.stride(0), _ptr(g_pp), g_pp.stride(0), _ptr(h), h.stride(0), _ptr(g_h), g_h.stride(0), _ptr(absmax)), dev)
    return (g_x, g_pp, h, g_h, absmax) if want_absmax else (g_x, g_pp, h, g_h)


def linear_wgrad_split16(g, inp, g_absmax, in_exp=14, want_bias=True, rows=None):
    """linear_wgrad's split path on f16 pairs (jf_linear_wgrad_split16_f32): g scaled by the power of two that brings g_absmax (a device
    scalar >= max |g|) into [2^14, 2^15), inp by 2^in_exp; for the packed gradient rows and tanh activations of the fused block's adjoint.
    rows (tuple of ints): the result's row i is row rows[i] of the product (packed parameter columns back in natural order, padding
    dropped) -- laid out by the slab sum itself"""
    dev = require_device(g, inp, g_absmax)
    g, inp = _rowmajor(g), _rowmajor(inp)
    B, N = g.shape
    K = inp.shape[1]
    S = int(lib().jf_linear_wgrad_split_splits(B, N))
    pw = torch.empty((S, N, K), dtype=g.dtype, device=g.device)
    pb = torch.empty((S, N), dtype=g.dtype, device=g.device) if want_bias else None
    _launch("jf_linear_wgrad_split16_f32", "K%d_N%d" % (K, N), (_ptr(g), g.stride(0), _ptr(inp), inp.stride(0), B, K, N, _ptr(g_absmax), in_exp,
                                                             _ptr(pw), _ptr(pb)), dev)
    if rows is None:
        return slab_sum(pw, pb)
    n = len(rows)

    def build():
        inv = np.full(N, -1, dtype=np.int64)
        inv[np.asarray(rows, dtype=np.int64)] = np.arange(n)
        mw = np.where(inv[:, None] >= 0, inv[:, None] * K + np.arange(K)[None, :], -1).reshape(-1)
        if not want_bias:
            return mw
        return np.concatenate([mw, np.where(inv >= 0, n * K + inv, -1)])
    out = slab_sum(pw, pb, out_map=slab_map(("wgrad_rows", N, K, bool(want_bias), tuple(rows)), g.device, build), out_size=n * K + (n if want_bias else 0))
    return out[:n * K].view(n, K), (out[n * K:] if want_bias else None)


def cond_gf_chain_fwd_split(inp, w1, b1, packed, z, log_det, layer_array, n_layers, D, x_out=None, status=None, kind="split"):
    """sampling direction of a conditional e-block in one launch (amortisation MLP + bisection / Newton solves on register-resident
    parameters); `packed`: the "split" / "split16" image of cond_gf_pack (`kind` names which)."""
    seg = None
    if isinstance(inp, SegInput):
        if kind == "split16" and inp.in_place_ok and inp.dtype == torch.float32:
            seg = inp                                      # read in place by jf_cond_gf_chain_split3_f32 (the earlier blocks' samples where they are)
            inp = seg.segments[0][0]
        else:
            inp = inp.materialize()
    dev = require_device(inp, w1, b1, packed, z, log_det, x_out, status, *(seg.tensors() if seg else []))
    inp, w1, z = _rowmajor(inp), _rowmajor(w1), _rowmajor(z)
    B, K1 = seg.shape if seg else inp.shape
    H = w1.shape[0]
    if z.shape[0] != B or z.shape[1] != D or w1.shape[1] != K1 or b1.shape[0] != H:
        raise ValueError("cond_gf_chain_fwd_split: inconsistent shapes")
    if any(t.dtype != torch.float32 for t in (inp, w1, b1, z)) or packed.dtype != torch.uint8:
        raise TypeError("cond_gf_chain_fwd_split: float32 inputs and a uint8 packed image expected")
    if log_det is not None:
        log_det = log_det.contiguous()
    if x_out is None:
        x_out = torch.empty((B, D), dtype=z.dtype, device=z.device)
    ld_out = torch.empty((B,), dtype=z.dtype, device=z.device)
    if seg is not None:
        _launch("jf_cond_gf_chain_split3_f32", "K%d_H%d_L%d_D%d_fwd" % (K1, H, n_layers, D),
                (DIR_FWD, SPLIT_F16X2, seg.c_array(), len(seg.segments), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(packed), K1, H, _ptr(z),
                 z.stride(0), _ptr(log_det), B, D, n_layers, layer_array, _ptr(x_out), x_out.stride(0), _ptr(ld_out), None, None, None, None, None, None,
                 _ptr(status)), dev)
        return x_out, ld_out
    if kind == "split16":
        _launch("jf_cond_gf_chain_split2_f32", "K%d_H%d_L%d_D%d_fwd" % (K1, H, n_layers, D),
                (DIR_FWD, SPLIT_F16X2, _ptr(inp), inp.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(packed), K1, H, _ptr(z), z.stride(0),
                 _ptr(log_det), B, D, n_layers, layer_array, _ptr(x_out), x_out.stride(0), _ptr(ld_out), None, None, None, _ptr(status)), dev)
        return x_out, ld_out
    _launch("jf_cond_gf_chain_fwd_split_f32", "K%d_H%d_L%d_D%d" % (K1, H, n_layers, D),
            (_ptr(inp), inp.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(packed), K1, H, _ptr(z), z.stride(0), _ptr(log_det), B, D,
             n_layers, layer_array, _ptr(x_out), x_out.stride(0), _ptr(ld_out), _ptr(status)), dev)
    return x_out, ld_out


def linear(inp, weight, bias=None, act=0, out=None):
    """out = act(inp @ weight^T + bias) on the matrix cores; act 0 identity / 1 tanh."""
    dev = require_device(inp, weight, bias, out)
    inp = _rowmajor(inp)
    weight = _rowmajor(weight)
    if weight.dtype != inp.dtype or (bias is not None and bias.dtype != inp.dtype):
        raise TypeError("linear: dtype mismatch")
    B, K = inp.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError("linear: weight %s does not match input width %d" % (tuple(weight.shape), K))
    if bias is not None:
        bias = bias.contiguous()
    if out is None:
        out = torch.empty((B, N), dtype=inp.dtype, device=inp.device)
    suf = _suffix(inp)
    _launch("jf_linear" + suf, "K%d_N%d" % (K, N),
            (_ptr(inp), inp.stride(0), _ptr(weight), weight.stride(0), _ptr(bias), B, K, N, act, _ptr(out), out.stride(0)), dev)
    return out


_SLAB_MAPS = {}


def slab_map(key, device, build):
    """device int32 map of a slab layout -> its consumer's layout (jf_slab_sum_map), built once per (key, device) by `build()` -> 1-d integer
    array-like with one destination index per slab element (a's, then b's; negative = dropped)"""
    k = (key, str(device))
    m = _SLAB_MAPS.get(k)
    if m is None:
        m = _SLAB_MAPS[k] = torch.as_tensor(build(), dtype=torch.int32).contiguous().to(device)
    return m


def slab_sum(a, b=None, out_map=None, out_size=None):
    """(a.sum(0), b.sum(0) or None) of the partial slabs a (S, ...) / b (S, ...) of one backward launch with a fixed summation order
    (jf_slab_sum: one launch for both arrays; two for hundreds of slabs -- chunks of 32 first); S == 1: views, no launch.
    out_map (slab_map) + out_size: ONE flat tensor of out_size elements instead, element i of a / na + i of b summed into position out_map[i]
    (jf_slab_sum_map) -- the caller cuts contiguous views out of it, no copy / gather launch afterwards"""
    S = a.shape[0]
    if S == 1 and out_map is None:
        return a[0], (None if b is None else b[0])
    dev = require_device(a, b, out_map)
    a = a.contiguous()
    b = None if b is None else b.contiguous()
    na, nb = a[0].numel(), 0 if b is None else b[0].numel()
    fn = "jf_slab_sum" + _suffix(a)
    shape_a, shape_b = a.shape[1:], None if b is None else b.shape[1:]
    if S > 128:
        chunk = 32
        n_chunks = (S + chunk - 1) // chunk
        mid_a = torch.empty((n_chunks, na), dtype=a.dtype, device=a.device)
        mid_b = torch.empty((n_chunks, nb), dtype=a.dtype, device=a.device) if b is not None else None
        _launch(fn, "chunks", (_ptr(a), na, _ptr(mid_a), _ptr(b), nb, _ptr(mid_b), S, chunk), dev)
        a, b, S = mid_a, mid_b, n_chunks
    if out_map is not None:
        if out_map.numel() != na + nb or out_map.dtype != torch.int32:
            raise ValueError("slab_sum: out_map must hold one int32 entry per slab element")
        out = torch.empty((out_size,), dtype=a.dtype, device=a.device)
        _launch("jf_slab_sum_map" + _suffix(a), "total", (_ptr(a), na, _ptr(b), nb, _ptr(out_map), _ptr(out), S), dev)
        return out
    out_a = torch.empty(shape_a, dtype=a.dtype, device=a.device)
    out_b = torch.empty(shape_b, dtype=a.dtype, device=a.device) if b is not None else None
    _launch(fn, "total", (_ptr(a), na, _ptr(out_a), _ptr(b), nb, _ptr(out_b), S, S), dev)
    return out_a, out_b


def linear_wgrad(g, inp, want_bias=True):
    """(g^T @ inp (N, K), g.sum(0) (N) or None): the batch-reducing products of a dense layer's backward, split over the grid (jf_linear_wgrad);
    K > 128 goes to the library GEMM."""
    dev = require_device(g, inp)
    g = _rowmajor(g)
    inp = _rowmajor(inp)
    B, N = g.shape
    K = inp.shape[1]
    if (K > 128 and min(K, N) > 16) or B == 0:
        return g.t() @ inp, (g.sum(0) if want_bias else None)
    split = (g.dtype == torch.float32 and K % 4 == 0 and N % 4 == 0 and N >= 64 and 16 < K <= 128 and B >= 4096 and g.stride(0) % 4 == 0 and inp.stride(0) % 4 == 0
             and g.data_ptr() % 16 == 0 and inp.data_ptr() % 16 == 0)
    if split:                                         # large float32 products: split-bf16 MFMA (csrc/split_gemm_kernels.hip)
        S = int(lib().jf_linear_wgrad_split_splits(B, N))
        pw = torch.empty((S, N, K), dtype=g.dtype, device=g.device)
        pb = torch.empty((S, N), dtype=g.dtype, device=g.device) if want_bias else None
        _launch("jf_linear_wgrad_split_f32", "K%d_N%d" % (K, N), (_ptr(g), g.stride(0), _ptr(inp), inp.stride(0), B, K, N, _ptr(pw), _ptr(pb)), dev)
        return slab_sum(pw, pb)
    S = int(getattr(lib(), "jf_linear_wgrad_splits" + _suffix(g))(B, K, N))
    pw = torch.empty((S, N, K), dtype=g.dtype, device=g.device)
    pb = torch.empty((S, N), dtype=g.dtype, device=g.device) if want_bias else None
    _launch("jf_linear_wgrad" + _suffix(g), "K%d_N%d" % (K, N), (_ptr(g), g.stride(0), _ptr(inp), inp.stride(0), B, K, N, _ptr(pw), _ptr(pb)), dev)
    return slab_sum(pw, pb)


def linear_split_ok(x, weight, bias=None):
    """True when jf_linear_split_f32 can take out = x @ weight^T: float32, K and N multiples of 4, 16-byte aligned rows"""
    N, K = weight.shape
    if bias is not None and (bias.dtype != torch.float32 or bias.data_ptr() % 16 or bias.stride(0) != 1):
        return False
    return (x.dtype == torch.float32 and weight.dtype == torch.float32 and K % 4 == 0 and N % 4 == 0 and N <= 8192 and x.dim() == 2 and x.shape[1] == K
            and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and x.shape[0] > 0)


def linear_split(x, weight, bias=None):
    """x (B, K) @ weight (N, K)^T + bias on split-bf16 MFMA (float32-equivalent accuracy); weight may be any strided 2-d view (a transposed
    weight packs without a copy).  The weight image is packed per call (N K elements: microseconds next to the product)."""
    dev = require_device(x, weight, bias)
    if not linear_split_ok(x, weight, bias):
        raise ValueError("linear_split: unsupported shape / dtype / alignment (see linear_split_ok)")
    B, K = x.shape
    N = weight.shape[0]
    nbytes = int(lib().jf_linear_split_packed_bytes(N, K))
    packed = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _launch("jf_linear_split_pack_f32", "", (_ptr(weight), weight.stride(0), weight.stride(1), N, K, _ptr(packed)), dev)
    out = torch.empty((B, N), dtype=x.dtype, device=x.device)
    _launch("jf_linear_split_f32", "K%d_N%d" % (K, N), (_ptr(x), x.stride(0), _ptr(packed), _ptr(bias), B, K, N, _ptr(out), out.stride(0)), dev)
    return out


MLP2_SMALL_MAX_IN, MLP2_SMALL_MAX_OUT = 32, 16
MLP2_SMALL_WIDE_IN, MLP2_SMALL_WIDE_OUT = 8, {4: 64, 8: 48}           # with <= 8 inputs: <= 64 outputs (float64: 48), csrc/wgrad_kernels.hip


def mlp2_small_shape_ok(k1, h, n, itemsize):
    """the shapes jf_mlp2_small_bwd takes (csrc/wgrad_kernels.hip: MS_K1MAX, MS_NMAX, MS_K1WIDE, MS_NWIDE / MS_NWIDE64)"""
    n_max = MLP2_SMALL_WIDE_OUT.get(itemsize, MLP2_SMALL_MAX_OUT) if k1 <= MLP2_SMALL_WIDE_IN else MLP2_SMALL_MAX_OUT
    return 1 <= k1 <= MLP2_SMALL_MAX_IN and 1 <= h <= MLP2_MAX_HIDDEN and 1 <= n <= n_max


def mlp2_small_bwd(x, w1, b1, w2, g):
    """gradients (g_w1, g_b1, g_w2, g_b2) of out = tanh(x w1^T + b1) w2^T + b2 for upstream g (B, N), narrow heads only (K1 <= 32, N <= 16 --
    or K1 <= 8, N <= 64 (float64: 48) -- H <= 128): one launch that recomputes the hidden activations (jf_mlp2_small_bwd); no input gradient."""
    dev = require_device(x, w1, b1, w2, g)
    x, w1, w2, g = _rowmajor(x), _rowmajor(w1), _rowmajor(w2), _rowmajor(g)
    B, K1 = x.shape
    H, N = w1.shape[0], w2.shape[0]
    if B == 0:                                         # no rows: the kernels do not launch, the gradients are exact zeros
        z = torch.zeros((H, K1 + 1 + N), dtype=x.dtype, device=x.device)
        return z[:, :K1], z[:, K1], z[:, K1 + 1:].t(), torch.zeros((N,), dtype=x.dtype, device=x.device)
    S = int(lib().jf_mlp2_small_bwd_slabs(B))
    slab = torch.empty((S, H, K1 + 1 + N), dtype=x.dtype, device=x.device)
    slab_b2 = torch.empty((S, N), dtype=x.dtype, device=x.device)
    _launch("jf_mlp2_small_bwd" + _suffix(x), "K%d_H%d_N%d" % (K1, H, N),
            (_ptr(x), x.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(w2), w2.stride(0), _ptr(g), g.stride(0), B, K1, H, N, _ptr(slab),
             _ptr(slab_b2)), dev)
    # the slab's (H, K1 | 1 | N) columns and the b2 slab -> [g_w1 (H, K1) | g_b1 (H) | g_w2 (N, H) | g_b2 (N)], every piece contiguous
    def build():
        h, c = np.meshgrid(np.arange(H), np.arange(K1 + 1 + N), indexing="ij")
        m = np.where(c < K1, h * K1 + c, np.where(c == K1, H * K1 + h, H * K1 + H + (c - K1 - 1) * H + h)).reshape(-1)
        return np.concatenate([m, H * K1 + H + N * H + np.arange(N)])
    out = slab_sum(slab, slab_b2, out_map=slab_map(("mlp2_small", H, K1, N), x.device, build), out_size=H * K1 + H + N * H + N)
    o1, o2, o3 = H * K1, H * K1 + H, H * K1 + H + N * H
    return out[:o1].view(H, K1), out[o1:o2], out[o2:o3].view(N, H), out[o3:]


def mlp_hidden_bwd(x, w1, b1, g_hidden):
    """(g_w1, g_b1) of h = tanh(x w1^T + b1) for the gradient g_hidden (B, H) with respect to h: tanh derivative + first-layer weight / bias
    gradient in one launch (jf_mlp_hidden_bwd; K1 <= 32, H <= 128), the activations recomputed from x"""
    dev = require_device(x, w1, b1, g_hidden)
    x, w1, g_hidden = _rowmajor(x), _rowmajor(w1), _rowmajor(g_hidden)
    B, K1 = x.shape
    H = w1.shape[0]
    if B == 0:
        z = torch.zeros((H, K1 + 1), dtype=x.dtype, device=x.device)
        return z[:, :K1], z[:, K1]
    S = int(lib().jf_mlp2_small_bwd_slabs(B))
    slab = torch.empty((S, H, K1 + 1), dtype=x.dtype, device=x.device)
    _launch("jf_mlp_hidden_bwd" + _suffix(x), "K%d_H%d" % (K1, H),
            (_ptr(x), x.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(g_hidden), g_hidden.stride(0), B, K1, H, _ptr(slab)), dev)
    def build():
        h, c = np.meshgrid(np.arange(H), np.arange(K1 + 1), indexing="ij")
        return np.where(c < K1, h * K1 + c, H * K1 + h).reshape(-1)
    out = slab_sum(slab, out_map=slab_map(("mlp_hidden", H, K1), x.device, build), out_size=H * K1 + H)
    return out[:H * K1].view(H, K1), out[H * K1:]


def tanh_bwd(g, y, inplace=False):
    """g * (1 - y^2) in one launch (the backward of a tanh whose output y was saved); inplace=True overwrites g (a fresh temporary)."""
    dev = require_device(g, y)
    if g.shape != y.shape or g.dtype != y.dtype:
        raise ValueError("tanh_bwd: shape / dtype mismatch")
    g, y = g.contiguous(), y.contiguous()
    out = g if inplace else torch.empty_like(g)
    _launch("jf_tanh_bwd" + _suffix(g), "", (_ptr(g), _ptr(y), g.numel(), _ptr(out)), dev)
    return out


ACT_CODES = {"relu": 2, "softplus": 3, "elu": 4, "swish": 5, "square": 6, "identity": 7}      # include/jammy_hip.h JF_ACT_*


def activation(z, code):
    """act(z), elementwise, for the AmortizableMLP nonlinearities other than tanh."""
    dev = require_device(z)
    z = z.contiguous()
    out = torch.empty_like(z)
    _launch("jf_activation" + _suffix(z), "", (_ptr(z), z.numel(), code, _ptr(out)), dev)
    return out


def activation_bwd(g, z, code):
    dev = require_device(g, z)
    g, z = g.contiguous(), z.contiguous()
    out = torch.empty_like(z)
    _launch("jf_activation_bwd" + _suffix(z), "", (_ptr(g), _ptr(z), z.numel(), code, _ptr(out)), dev)
    return out


class SegInput:
    """an MLP input row block cat[conditional_input, embed(x_0), ...] (main/default.py:946-962) described by its segments instead of being
    materialised: list of (tensor (B, n), kind) as for conditioning_rows.  Consumers with a *_seg / split3 entry point read the segments in
    place (csrc/jf_cond_in.h); every other consumer gets .materialize() (one jf_conditioning_rows launch)."""
    MAX_IN_PLACE = 4          # csrc/jf_cond_in.h: JF_COND_IN_MAX

    def __init__(self, segments, B, dtype, device):
        self.segments = [(_rowmajor(t), kind) for t, kind in segments]
        self.width = sum(t.shape[1] if kind == 0 else kind + 1 for t, kind in self.segments)
        self.shape = (B, self.width)
        self.dtype, self.device = dtype, device
        self._rows = None
        for i, (t, kind) in enumerate(self.segments):
            if t.dtype != dtype or t.shape[0] != B:
                raise TypeError("SegInput: segment %d has dtype %s / %d rows, expected %s / %d" % (i, t.dtype, t.shape[0], dtype, B))
            if kind != 0 and t.shape[1] != kind:
                raise ValueError("SegInput: an S%d segment needs %d intrinsic columns" % (kind, kind))

    @property
    def in_place_ok(self):
        """worth reading in place?  float32: yes (the consumer embeds angles with the hardware sine / cosine while it stages its tile).
        float64: only plain column ranges -- double-precision sines in a consumer's serial prologue cost more than the launch they replace
        (jf_mlp2_i8 1.39 -> 1.57 ms per 2^20 rows against 0.09 ms for jf_conditioning_rows)."""
        if not 1 <= len(self.segments) <= self.MAX_IN_PLACE:
            return False
        return self.dtype == torch.float32 or all(kind == 0 for _, kind in self.segments)

    def c_array(self):
        arr = (jf_cond_segment * len(self.segments))()
        for i, (t, kind) in enumerate(self.segments):
            arr[i] = jf_cond_segment(_ptr(t), t.stride(0), kind, t.shape[1])
        return arr

    def tensors(self):
        return [t for t, _ in self.segments]

    def materialize(self):
        if self._rows is None:
            self._rows = conditioning_rows(self.segments, self.shape[0], self.dtype, self.device)
        return self._rows


def as_matrix(inp):
    """a consumer without segment support: the (B, K1) input matrix itself (one jf_conditioning_rows launch for a SegInput)"""
    return inp.materialize() if isinstance(inp, SegInput) else inp


def conditioning_rows(segments, B, dtype, device):
    """segments: list of (tensor (B, n), kind) with kind 0 = copy the columns, 1 = S1 angle -> (cos, sin), 2 = S2 (theta, phi) -> (x, y, z).
    Returns the (B, sum of output widths) row block cat[...] in ONE launch (the amortisation MLPs read prefixes of it)."""
    if not 1 <= len(segments) <= JF_MAX_SEGMENTS:
        raise ValueError("conditioning_rows: 1..%d segments" % JF_MAX_SEGMENTS)
    arr = (jf_cond_segment * len(segments))()
    width = 0
    keep = []
    dev = require_device(*[t for t, _ in segments])
    for i, (t, kind) in enumerate(segments):
        t = _rowmajor(t)
        if t.dtype != dtype or t.shape[0] != B:
            raise TypeError("conditioning_rows: segment %d has dtype %s / %d rows, expected %s / %d" % (i, t.dtype, t.shape[0], dtype, B))
        if kind != 0 and t.shape[1] != kind:
            raise ValueError("conditioning_rows: an S%d segment needs %d intrinsic columns" % (kind, kind))
        keep.append(t)
        arr[i] = jf_cond_segment(_ptr(t), t.stride(0), kind, t.shape[1])
        width += t.shape[1] if kind == 0 else kind + 1
    out = torch.empty((B, width), dtype=dtype, device=device)
    _launch("jf_conditioning_rows" + _suffix(out), "", (arr, len(segments), B, _ptr(out), out.stride(0)), dev)
    return out


def coverage_histogram(log_prob_base, log_at_zero, thresholds, want_twice=True):
    """counting step of approximate_coverage on the device -> (counts (n,) int64 on the host: #rows with twice < thresholds[i], twice (B,))"""
    dev = require_device(log_prob_base, thresholds)
    lpb = log_prob_base.contiguous()
    thr = thresholds.to(dtype=lpb.dtype).contiguous()
    n = thr.shape[0]
    hist = torch.zeros((n + 1,), dtype=torch.int64, device=lpb.device)
    twice = torch.empty_like(lpb) if want_twice else None
    _launch("jf_coverage_histogram" + _suffix(lpb), "", (_ptr(lpb), lpb.shape[0], float(log_at_zero), _ptr(thr), n, _ptr(hist), _ptr(twice)), dev)
    return torch.cumsum(hist[:n], 0).cpu().numpy(), twice


def segment_reduce(values, seg_len, mode):
    """values (n_seg * seg_len,) -> (n_seg,): mode 'neg_mean' = -mean over each segment, 'logmeanexp' = logsumexp - log(seg_len)"""
    dev = require_device(values)
    v = values.contiguous().reshape(-1)
    assert v.shape[0] % seg_len == 0
    n_seg = v.shape[0] // seg_len
    out = torch.empty((n_seg,), dtype=v.dtype, device=v.device)
    _launch("jf_segment_reduce" + _suffix(v), mode, (_ptr(v), n_seg, seg_len, {"neg_mean": 0, "logmeanexp": 1}[mode], _ptr(out)), dev)
    return out


def amlp_stage(x, seg, n_in, n_out, rank, has_bias, act, residual=None):
    """one AmortizableMLP stage with per-sample weights: seg (B, n_u + n_v + n_b) = this stage's slice of the per-sample parameter block
    (a strided view is fine).  out = act(W_b x_b + bias_b) (+ residual)."""
    dev = require_device(x, seg, residual)
    x, seg = _rowmajor(x), _rowmajor(seg)
    B = x.shape[0]
    if x.shape[1] != n_in or seg.shape[0] != B or x.dtype != seg.dtype:
        raise ValueError("amlp_stage: inconsistent shapes / dtypes")
    if residual is not None:
        residual = _rowmajor(residual)
    out = torch.empty((B, n_out), dtype=x.dtype, device=x.device)
    _launch("jf_amlp_stage" + _suffix(x), "in%d_out%d_r%d" % (n_in, n_out, rank),
            (_ptr(x), x.stride(0), _ptr(seg), seg.stride(0), B, n_in, n_out, rank, 1 if has_bias else 0, act, _ptr(residual),
             residual.stride(0) if residual is not None else 0, _ptr(out), out.stride(0)), dev)
    return out


def amlp_stage_bwd(x, seg, n_in, n_out, rank, has_bias, act, y, g_out, want_g_in=True):
    dev = require_device(x, seg, y, g_out)
    x, seg, g_out = _rowmajor(x), _rowmajor(seg), _rowmajor(g_out)
    B = x.shape[0]
    g_seg = torch.empty((B, seg.shape[1]), dtype=x.dtype, device=x.device)
    g_in = torch.empty((B, n_in), dtype=x.dtype, device=x.device) if want_g_in else None
    _launch("jf_amlp_stage_bwd" + _suffix(x), "in%d_out%d_r%d" % (n_in, n_out, rank),
            (_ptr(x), x.stride(0), _ptr(seg), seg.stride(0), B, n_in, n_out, rank, 1 if has_bias else 0, act, _ptr(y), y.stride(0) if y is not None else 0,
             _ptr(g_out), g_out.stride(0), _ptr(g_in), g_in.stride(0) if g_in is not None else 0, _ptr(g_seg), g_seg.stride(0)), dev)
    return g_in, g_seg


def normal_logp(z, acc=None):
    """acc + sum_d N(0,1).log_prob(z[:, d]) -> (B,)"""
    dev = require_device(z, acc)
    z = _rowmajor(z)
    B, D = z.shape
    out = torch.empty((B,), dtype=z.dtype, device=z.device)
    suf = _suffix(z)
    _launch("jf_normal_logp" + suf, "", (_ptr(z), z.stride(0), B, D, _ptr(acc), _ptr(out)), dev)
    return out


def mchain(fam, direction, x, log_det, params, layer_structs, dim, x_out=None, base_logp_in=None, want_base_logp=False, bins=None, status=None,
           pre_ld=None, pre_blp=None):
    """run a chain of manifold layers of family `fam` ('r','o','m','f','v','c') on intrinsic coordinates.
    x (B, dim) view; params (1|B, P) or None when the chain has no parameters.  Returns (x_out, log_det_out[, base_logp]).
    pre_ld / pre_blp ('inv' with want_base_logp; lists of (B,) tensors, <= 4 each): the chain is the last block of its pdf and also adds the
    earlier blocks' sums in front of its own -> (x_out, log_det total, base_logp total, total log-prob) (jf_<fam>_chain_inv_sum)."""
    dev = require_device(x, log_det, params, x_out, base_logp_in, status, bins)
    x = _rowmajor(x)
    B = x.shape[0]
    if x.shape[1] != dim:
        raise ValueError("expected %d target columns, got %d" % (dim, x.shape[1]))
    pb = 1
    if params is not None and params.shape[1] > 0:
        params = _rowmajor(params)
        if params.dtype != x.dtype:
            raise TypeError("parameter dtype %s != input dtype %s" % (params.dtype, x.dtype))
        if params.shape[0] not in (1, B):
            raise ValueError("extra_inputs must have 1 or B=%d rows, got %d" % (B, params.shape[0]))
        pb = 1 if params.shape[0] == 1 else B
        pptr, pstride = _ptr(params), params.stride(0)
    else:
        pptr, pstride = None, 0
    if log_det is not None:
        log_det = log_det.contiguous()
        if log_det.dtype != x.dtype or log_det.shape[0] != B:
            raise ValueError("log_det must be a (B,) tensor of the input dtype")
    if x_out is None:
        x_out = torch.empty((B, dim), dtype=x.dtype, device=x.device)
    ld_out = torch.empty((B,), dtype=x.dtype, device=x.device)
    blp_out = torch.empty((B,), dtype=x.dtype, device=x.device) if want_base_logp else None
    n = len(layer_structs)
    arr = (MCHAIN_LAYER_TYPES[fam] * n)(*layer_structs)
    suf = _suffix(x)
    name = "jf_%s_chain_%s%s" % (fam, direction, suf)
    if bins is None and BINS_LOG is not None and fam in ("r", "o", "f"):
        n_search = sum(1 if fam in "ro" else (L.n_vertical + L.n_circular) for L in layer_structs)
        if n_search:
            bins = torch.full((B, n_search), -3, dtype=torch.int64, device=x.device)
            BINS_LOG.append(bins)
    if bins is not None:
        assert bins.dtype == torch.int64 and bins.dim() == 2 and bins.shape[0] == B and bins.stride(1) == 1
    if pre_ld is not None or pre_blp is not None:
        if direction != "inv" or not want_base_logp:
            raise ValueError("mchain: pre_ld / pre_blp need direction 'inv' and want_base_logp")
        def lst(items):
            r = jf_row_list()
            r.n = len(items)
            for i, t in enumerate(items):
                if t.dtype != x.dtype or t.shape != (B,) or not t.is_contiguous():
                    raise ValueError("mchain: pre_ld / pre_blp entries are contiguous (B,) tensors of the input dtype")
                r.p[i] = _ptr(t)
            return r
        require_device(x, *(list(pre_ld or []) + list(pre_blp or [])))
        la, lb = lst(list(pre_ld or [])), lst(list(pre_blp or []))
        total = torch.empty((B,), dtype=x.dtype, device=x.device)
        _launch("jf_%s_chain_inv_sum%s" % (fam, suf), "bcast" if pb == 1 else "per-sample",
                (_ptr(x), x.stride(0), _ptr(log_det), pptr, pstride, pb, B, n, arr, _ptr(x_out), x_out.stride(0), _ptr(ld_out), _ptr(base_logp_in),
                 _ptr(blp_out), ctypes.byref(la), ctypes.byref(lb), _ptr(total), _ptr(bins), bins.stride(0) if bins is not None else 0, _ptr(status)), dev)
        return x_out, ld_out, blp_out, total
    _launch(name, "bcast" if pb == 1 else "per-sample",
            (_ptr(x), x.stride(0), _ptr(log_det), pptr, pstride, pb, B, n, arr, _ptr(x_out), x_out.stride(0), _ptr(ld_out), _ptr(base_logp_in),
             _ptr(blp_out), _ptr(bins), bins.stride(0) if bins is not None else 0, _ptr(status)), dev)
    return (x_out, ld_out, blp_out) if want_base_logp else (x_out, ld_out)


COND_MCHAIN_FAMILIES, COND_MCHAIN_MAX_PARAMS = "romf", 64


def cond_mchain_inv(fam, inp, w1, b1, w2, b2, x, log_det, layer_structs, dim, x_out=None, base_logp_in=None, want_base_logp=False, status=None):
    """default amortisation MLP (Linear-tanh-Linear) + the chain of manifold layers of family `fam` it parametrises in ONE launch"""
    dev = require_device(inp, w1, b1, w2, b2, x, log_det, x_out, base_logp_in, status)
    inp, w1, w2, x = _rowmajor(inp), _rowmajor(w1), _rowmajor(w2), _rowmajor(x)
    B, K1 = inp.shape
    H = w1.shape[0]
    if x.shape != (B, dim) or w1.shape[1] != K1 or w2.shape[1] != H or b1.shape[0] != H or b2.shape[0] != w2.shape[0]:
        raise ValueError("cond_mchain_inv: inconsistent shapes")
    if any(t.dtype != x.dtype for t in (inp, w1, b1, w2, b2)):
        raise TypeError("cond_mchain_inv: dtype mismatch")
    if log_det is not None:
        log_det = log_det.contiguous()
    if x_out is None:
        x_out = torch.empty((B, dim), dtype=x.dtype, device=x.device)
    ld_out = torch.empty((B,), dtype=x.dtype, device=x.device)
    blp_out = torch.empty((B,), dtype=x.dtype, device=x.device) if want_base_logp else None
    n = len(layer_structs)
    arr = (MCHAIN_LAYER_TYPES[fam] * n)(*layer_structs)
    ok = _launch("jf_cond_%s_chain_inv%s" % (fam, _suffix(x)), "K%d_H%d_N%d" % (K1, H, w2.shape[0]),
            (_ptr(inp), inp.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(w2), w2.stride(0), _ptr(b2.contiguous()), K1, H, _ptr(x),
             x.stride(0), _ptr(log_det), B, n, arr, _ptr(x_out), x_out.stride(0), _ptr(ld_out), _ptr(base_logp_in), _ptr(blp_out), _ptr(status)), dev,
            unsupported_ok=True)
    if not ok:
        return None                                   # outside the fused kernel's limits (LDS budget): use the two-launch path
    return (x_out, ld_out, blp_out) if want_base_logp else (x_out, ld_out)


def cond_mchain_fwd(fam, inp, w1, b1, w2, b2, z, log_det, layer_structs, dim, x_out=None, status=None):
    """sampling direction of cond_mchain_inv: default amortisation MLP + the manifold chain forwards in ONE launch -> (x, log_det), or None
    outside the fused kernel's limits"""
    dev = require_device(inp, w1, b1, w2, b2, z, log_det, x_out, status)
    inp, w1, w2, z = _rowmajor(inp), _rowmajor(w1), _rowmajor(w2), _rowmajor(z)
    B, K1 = inp.shape
    H = w1.shape[0]
    if z.shape != (B, dim) or w1.shape[1] != K1 or w2.shape[1] != H or b1.shape[0] != H or b2.shape[0] != w2.shape[0]:
        raise ValueError("cond_mchain_fwd: inconsistent shapes")
    if any(t.dtype != z.dtype for t in (inp, w1, b1, w2, b2)):
        raise TypeError("cond_mchain_fwd: dtype mismatch")
    if log_det is not None:
        log_det = log_det.contiguous()
    if x_out is None:
        x_out = torch.empty((B, dim), dtype=z.dtype, device=z.device)
    ld_out = torch.empty((B,), dtype=z.dtype, device=z.device)
    n = len(layer_structs)
    arr = (MCHAIN_LAYER_TYPES[fam] * n)(*layer_structs)
    ok = _launch("jf_cond_%s_chain_fwd%s" % (fam, _suffix(z)), "K%d_H%d_N%d" % (K1, H, w2.shape[0]),
                 (_ptr(inp), inp.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(w2), w2.stride(0), _ptr(b2.contiguous()), K1, H, _ptr(z),
                  z.stride(0), _ptr(log_det), B, n, arr, _ptr(x_out), x_out.stride(0), _ptr(ld_out), _ptr(status)), dev, unsupported_ok=True)
    return (x_out, ld_out) if ok else None


def t_layer(direction, x, log_det, params, struct, D, x_out=None, base_logp_in=None, want_base_logp=False, status=None):
    """'t' affine layer, direction 'inv' (log-prob) or 'fwd' (sampling).  Returns (x_out, log_det_out[, base_logp])."""
    dev = require_device(x, log_det, params, x_out, base_logp_in, status)
    x = _rowmajor(x)
    B = x.shape[0]
    if x.shape[1] != D:
        raise ValueError("expected %d target columns, got %d" % (D, x.shape[1]))
    pb, pptr, pstride = 1, None, 0
    if params is not None and params.shape[1] > 0:
        params = _rowmajor(params)
        if params.dtype != x.dtype or params.shape[0] not in (1, B):
            raise ValueError("extra_inputs must be (1 | B, P) of the input dtype")
        pb = 1 if params.shape[0] == 1 else B
        pptr, pstride = _ptr(params), params.stride(0)
    if log_det is not None:
        log_det = log_det.contiguous()
    if x_out is None:
        x_out = torch.empty((B, D), dtype=x.dtype, device=x.device)
    ld_out = torch.empty((B,), dtype=x.dtype, device=x.device)
    blp_out = torch.empty((B,), dtype=x.dtype, device=x.device) if want_base_logp else None
    _launch("jf_t_layer_%s%s" % (direction, _suffix(x)), "bcast" if pb == 1 else "per-sample",
            (_ptr(x), x.stride(0), _ptr(log_det), pptr, pstride, pb, B, D, ctypes.byref(struct), _ptr(x_out), x_out.stride(0), _ptr(ld_out),
             _ptr(base_logp_in), _ptr(blp_out), _ptr(status)), dev)
    return (x_out, ld_out, blp_out) if want_base_logp else (x_out, ld_out)


def t_layer_inv_bwd(x, params, struct, D, g_xout, g_ld, g_blp, status=None):
    dev = require_device(x, params, g_xout, g_ld, g_blp, status)
    x = _rowmajor(x)
    B = x.shape[0]
    P = 0 if params is None else params.shape[1]
    pb, pptr, pstride, g_p = 1, None, 0, None
    if P > 0:
        params = _rowmajor(params)
        pb = 1 if params.shape[0] == 1 else B
        pptr, pstride = _ptr(params), params.stride(0)
        g_p = torch.zeros((1, P), dtype=x.dtype, device=x.device) if pb == 1 else torch.empty((B, P), dtype=x.dtype, device=x.device)
    if g_xout is not None:
        g_xout = _rowmajor(g_xout)
    g_ld = None if g_ld is None else g_ld.contiguous()
    g_blp = None if g_blp is None else g_blp.contiguous()
    g_x = torch.empty((B, D), dtype=x.dtype, device=x.device)
    _launch("jf_t_layer_inv_bwd" + _suffix(x), "bcast" if pb == 1 else "per-sample",
            (_ptr(x), x.stride(0), pptr, pstride, pb, B, D, ctypes.byref(struct), _ptr(g_xout), g_xout.stride(0) if g_xout is not None else 0,
             _ptr(g_ld), _ptr(g_blp), _ptr(g_x), g_x.stride(0), _ptr(g_p), g_p.stride(0) if g_p is not None else 0, _ptr(status)), dev)
    if g_p is None:
        g_p = torch.zeros((1, 0), dtype=x.dtype, device=x.device)
    return g_x, g_p


def mchain_inv_bwd(fam, x, params, layer_structs, dim, g_xout, g_ld, g_blp, status=None):
    """vector-Jacobian product of mchain(fam, 'inv', ...): -> (g_x (B, dim), g_params in the shape of `params`)."""
    dev = require_device(x, params, g_xout, g_ld, g_blp, status)
    x = _rowmajor(x)
    B = x.shape[0]
    P = 0 if params is None else params.shape[1]
    pb = 1
    pptr, pstride = None, 0
    g_p = None
    if P > 0:
        params = _rowmajor(params)
        pb = 1 if params.shape[0] == 1 else B
        pptr, pstride = _ptr(params), params.stride(0)
        g_p = torch.zeros((1, P), dtype=x.dtype, device=x.device) if pb == 1 else torch.empty((B, P), dtype=x.dtype, device=x.device)
    if g_xout is not None:
        g_xout = _rowmajor(g_xout)
    g_ld = None if g_ld is None else g_ld.contiguous()
    g_blp = None if g_blp is None else g_blp.contiguous()
    g_x = torch.empty((B, dim), dtype=x.dtype, device=x.device)
    n = len(layer_structs)
    arr = (MCHAIN_LAYER_TYPES[fam] * n)(*layer_structs)
    _launch("jf_%s_chain_inv_bwd%s" % (fam, _suffix(x)), "bcast" if pb == 1 else "per-sample",
            (_ptr(x), x.stride(0), pptr, pstride, pb, B, n, arr, _ptr(g_xout), g_xout.stride(0) if g_xout is not None else 0, _ptr(g_ld), _ptr(g_blp),
             _ptr(g_x), g_x.stride(0), _ptr(g_p), g_p.stride(0) if g_p is not None else 0, _ptr(status)), dev)
    if g_p is None:
        g_p = torch.zeros((params.shape[0] if params is not None else 1, 0), dtype=x.dtype, device=x.device)
    return g_x, g_p


def sphere_embedding(x, log_det, dim, to_embedding, want_log_det=True):
    """S1: angle <-> (cos, sin); S2: (theta, phi) <-> (x, y, z), with the log-det bookkeeping of sphere_base.py:242-335:
    S2 adds +log sin(theta) towards the embedding and -log sin(theta) back, S1 adds nothing.  `log_det` may be a (B,) tensor, None (= 0) or a
    python number (the reference's public default ``log_det=0``): in every case the S2 Jacobian is included in what comes back, as a (B,)
    tensor.  want_log_det=False skips it (callers that only need the coordinates)."""
    is_tensor = isinstance(log_det, torch.Tensor)
    dev = require_device(x, log_det if is_tensor else None)
    x = _rowmajor(x)
    B = x.shape[0]
    ld_in = log_det.contiguous() if is_tensor else None
    out = torch.empty((B, dim + 1 if to_embedding else dim), dtype=x.dtype, device=x.device)
    want_ld = want_log_det and dim == 2
    ld_out = torch.empty((B,), dtype=x.dtype, device=x.device) if want_ld else None
    suf = _suffix(x)
    name = ("jf_sphere_to_embedding" if to_embedding else "jf_sphere_from_embedding") + suf
    _launch(name, "", (_ptr(x), x.stride(0), _ptr(ld_in), B, dim, _ptr(out), out.stride(0), _ptr(ld_out)), dev)
    if not want_ld:
        return out, log_det                      # S1 (no Jacobian) or not asked for: handed back unchanged
    if not is_tensor and log_det is not None and log_det != 0:
        ld_out = ld_out + log_det                # numeric offset of the caller
    return out, ld_out


MLP2_MAX_IN, MLP2_MAX_HIDDEN = 32, 128


def mlp2(inp, w1, b1, w2, b2, out=None):
    """tanh(inp @ w1^T + b1) @ w2^T + b2 in one launch (the hidden activations stay in registers, nothing but the result reaches HBM)."""
    dev = require_device(inp, w1, b1, w2, b2, out)
    inp, w1, w2 = _rowmajor(inp), _rowmajor(w1), _rowmajor(w2)
    B, K1 = inp.shape
    H, N = w1.shape[0], w2.shape[0]
    if w1.shape[1] != K1 or w2.shape[1] != H or b1.shape[0] != H or b2.shape[0] != N:
        raise ValueError("mlp2: inconsistent shapes")
    if any(t.dtype != inp.dtype for t in (w1, b1, w2, b2)):
        raise TypeError("mlp2: dtype mismatch")
    if out is None:
        # rows padded to whole 128-byte lines: every 64-lane result store of the kernel then covers full, aligned HBM lines
        # (an unpadded 548-float row makes every line at a tile boundary a partial write); the pad columns are never touched
        line = 128 // inp.element_size()
        out = torch.empty((B, (N + line - 1) // line * line), dtype=inp.dtype, device=inp.device)[:, :N]
    _launch("jf_mlp2" + _suffix(inp), "K%d_H%d_N%d" % (K1, H, N),
            (_ptr(inp), inp.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(w2), w2.stride(0), _ptr(b2.contiguous()), B, K1, H, N,
             _ptr(out), out.stride(0)), dev)
    return out


MATH_EXP_FAST, MATH_LOG_FAST, MATH_TANH_FAST, MATH_RCP = 0, 1, 2, 3


def device_math(x, fn):
    """elementwise M<T>::{exp_fast, log_fast, tanh_fast, rcp} of the flow kernels (csrc/jf_math.h) -- for accuracy tests"""
    dev = require_device(x)
    x = x.contiguous()
    out = torch.empty_like(x)
    _launch("jf_device_math" + _suffix(x), "", (_ptr(x), x.numel(), fn, _ptr(out)), dev)
    return out


MLP2_I8_MAX_IN, MLP2_I8_MAX_HIDDEN = 28, 128


def mlp2_i8_pack(w2, b2, slices=6):
    """W2 (N, H) / b2 (N,) float64 -> the int8 digit image mlp2_i8 streams (csrc/mlp_i8_kernels.hip).  Redo whenever the weights change."""
    dev = require_device(w2, b2)
    w2 = _rowmajor(w2)
    if w2.dtype != torch.float64 or b2.dtype != torch.float64:
        raise TypeError("mlp2_i8_pack: float64 only")
    N, H = w2.shape
    nbytes = int(lib().jf_mlp2_i8_packed_bytes(N, slices))
    _check(min(nbytes, 0), "jf_mlp2_i8_packed_bytes")
    packed = torch.empty((nbytes,), dtype=torch.uint8, device=w2.device)
    _launch("jf_mlp2_i8_pack_f64", "x%d" % slices, (_ptr(w2), w2.stride(0), _ptr(b2.contiguous()), H, N, slices, _ptr(packed)), dev)
    return packed


def mlp2_i8(inp, w1, b1, packed, N, slices=6, out=None):
    """mlp2 in float64 with the second product as int8 digit-slice products on the matrix cores; `packed` = mlp2_i8_pack(w2, b2, slices)."""
    seg = None
    if isinstance(inp, SegInput):
        if inp.in_place_ok and inp.dtype == torch.float64:
            seg = inp                                      # read in place by jf_mlp2_i8_seg_f64
            inp = seg.segments[0][0]
        else:
            inp = inp.materialize()
    dev = require_device(inp, w1, b1, packed, out, *(seg.tensors() if seg else []))
    inp, w1 = _rowmajor(inp), _rowmajor(w1)
    B, K1 = seg.shape if seg else inp.shape
    H = w1.shape[0]
    if w1.shape[1] != K1 or b1.shape[0] != H:
        raise ValueError("mlp2_i8: inconsistent shapes")
    if any(t.dtype != torch.float64 for t in (inp, w1, b1)) or packed.dtype != torch.uint8:
        raise TypeError("mlp2_i8: float64 inputs and a uint8 packed image expected")
    if packed.numel() != int(lib().jf_mlp2_i8_packed_bytes(N, slices)):
        raise ValueError("mlp2_i8: the packed image does not belong to N = %d, slices = %d" % (N, slices))
    if out is None:
        out = torch.empty((B, (N + 15) // 16 * 16), dtype=inp.dtype, device=inp.device)[:, :N]      # rows of whole 128-byte lines (see mlp2)
    if seg is not None:
        arr = seg.c_array()
        _launch("jf_mlp2_i8_seg_f64", "K%d_H%d_N%d_x%d" % (K1, H, N, slices),
                (arr, len(seg.segments), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(packed), B, K1, H, N, slices, _ptr(out), out.stride(0)), dev)
        return out
    _launch("jf_mlp2_i8_f64", "K%d_H%d_N%d_x%d" % (K1, H, N, slices),
            (_ptr(inp), inp.stride(0), _ptr(w1), w1.stride(0), _ptr(b1.contiguous()), _ptr(packed), B, K1, H, N, slices, _ptr(out), out.stride(0)), dev)
    return out
