"""Layer registry and default options of the in-scope flow layers.

Option names, defaults and validators are the user-visible API of the reference (jammy_flows/flow_options.py:25-240) and are
kept verbatim; the registry below only lists the layers of the MI355X hot path:

    g  Gaussianization flow (Euclidean)          t  affine flow / multivariate normal (Euclidean)
    r  rational-quadratic spline (interval)
    o  circular spline (S1)                      m  Moebius (S1)
    f  von-Mises-Fisher + splines (S2)           v  exponential map (S2, float64 only)
    x / y / z  identity layers
    n  legacy name of the S2 autoregressive spline layer; accepted as an alias of "f" 

Out of scope (no kernel, constructing them raises): h (deprecated), c, u, w.
"""
import importlib

ALIASES = {"n": "f"}

opts_dict = dict()


def _register(letter, module, cls, kind, kwargs):
    opts_dict[letter] = {"module_path": module, "class_name": cls, "type": kind, "kwargs": kwargs}


_pos = lambda x: x > 0
_int_or_m1 = lambda x: (x == -1) or (x > 0)

_register("g", "jammy_flows_amd.layers.euclidean.gaussianization_flow", "gf_block", "e", {
    "fit_normalization": (1, [0, 1]),
    "num_householder_iter": (-1, _int_or_m1),
    "num_kde": (10, _pos),
    "inverse_function_type": ("isigmoid", ["isigmoid", "inormal_partly_precise", "inormal_full_pade", "inormal_partly_crude"]),
    "replace_first_sigmoid_with_icdf": (1, [0, 1]),
    "skip_model_offset": (0, [0, 1]),
    "softplus_for_width": (0, [0, 1]),
    "upper_bound_for_widths": (100, _int_or_m1),
    "lower_bound_for_widths": (0.01, _pos),
    "upper_bound_for_norms": (10, _int_or_m1),
    "lower_bound_for_norms": (1, _pos),
    "center_mean": (0, [0, 1]),
    "clamp_widths": (0, [0, 1]),
    "width_smooth_saturation": (1, [0, 1]),
    "regulate_normalization": (1, [0, 1]),
    "add_skewness": (0, [0, 1]),
    "rotation_mode": ("householder", ["householder", "triangular_combination", "angles", "cayley", "none"]),
    "nonlinear_stretch_type": ("classic", ["classic", "rq_splines"]),
})

_register("t", "jammy_flows_amd.layers.euclidean.multivariate_normal", "mvn_block", "e", {
    "skip_model_offset": (0, [0, 1]),
    "softplus_for_width": (0, [0, 1]),
    "upper_bound_for_widths": (100, _int_or_m1),
    "lower_bound_for_widths": (0.01, _pos),
    "clamp_widths": (0, [0, 1]),
    "width_smooth_saturation": (1, [0, 1]),
    "cov_type": ("diagonal", ["identity", "diagonal_symmetric", "diagonal", "full"]),
})

_register("m", "jammy_flows_amd.layers.spheres.moebius_1d", "moebius", "s", {
    "add_rotation": (0, [0, 1]),
    "num_basis_functions": (5, _pos),
    "natural_direction": (0, [0, 1]),
})

_spline_common = {
    "fix_boundary_derivatives": (-1.0, lambda x: (x == -1.0) or (x > 0.0)),
    "fix_first_width_n_height_to_zero": (0, [0, 1]),
    "also_fix_second_width_to_zero": (0, [0, 1]),
    "independent_width_height_parametrization": (0, [0, 1]),
    "min_width": (1e-4, _pos),
    "min_height": (1e-4, _pos),
    "min_derivative": (1e-4, _pos),
}

_register("o", "jammy_flows_amd.layers.spheres.splines_1d", "spline_1d", "s", dict({
    "add_rotation": (1, [0, 1]),
    "num_basis_functions": (2, _pos),
    "natural_direction": (1, [0, 1]),
    "smooth_second_derivative": (1, [0, 1]),
}, **_spline_common))

_register("v", "jammy_flows_amd.layers.spheres.exponential_map_s2", "exponential_map_s2", "s", {
    "exp_map_type": ("exponential", ["linear", "quadratic", "splines", "exponential"]),
    "num_components": (10, _pos),
    "natural_direction": (0, [0, 1]),
    "add_rotation": (0, [0, 1]),
    "max_num_newton_iter": (1000, _pos),
    "mean_parametrization": ("old", ["old", "householder"]),
})

_register("f", "jammy_flows_amd.layers.spheres.fvm_2d", "fisher_von_mises_2d", "s", {
    "add_vertical_rq_spline_flow": (0, [0, 1]),
    "add_circular_rq_spline_flow": (0, [0, 1]),
    "add_correlated_rq_spline_flow": (0, [0, 1]),
    "circular_flow_defs": ("oo", lambda x: type(x) == str),
    "vertical_flow_defs": ("rr", lambda x: type(x) == str),
    "correlated_max_rank": (3, lambda x: x >= 0),
    "inverse_z_scaling": (1, [0, 1]),
    "boundary_cos_theta_identity_region": (0.0, lambda x: (x >= 0) and (x < 1)),
    "spline_num_basis_functions": (5, lambda x: (x > 0) or (x == -1)),
    "vertical_smooth": (0, [0, 1]),
    "vertical_restrict_max_min_width_height_ratio": (-1.0, lambda x: (x == -1.0) or (x > 0.0)),
    "vertical_fix_boundary_derivative": (1, [0, 1]),
    "vertical_fix_first_width_n_height_to_zero": (0, [0, 1]),
    "vertical_also_fix_second_width_to_zero": (0, [0, 1]),
    "vertical_independent_width_height_parametrization": (0, [0, 1]),
    "circular_add_rotation": (0, [0, 1]),
    "min_kappa": (1e-10, _pos),
    "kappa_prediction": ("direct_log_real_bounded", ["direct_log_real_bounded", "softplus_real_bounded", "log_bounded", "mu", "mu_squared",
                                                     "quatvec", "quatvec_squared"]),
    "add_extra_rotation_inbetween": (0, [0, 1]),
    "add_rotation": (1, [0, 1]),
    "rotation_mode": ("householder", ["householder", "angles", "xyz", "quaternion"]),
    "kappa_clamping": (0, [0, 1]),
    "num_householder_iter": (-1, _int_or_m1),
})

_register("r", "jammy_flows_amd.layers.intervals.rational_quadratic_spline", "rational_quadratic_spline", "i", dict({
    "num_basis_functions": (5, _pos),
    "smooth_second_derivative": (0, lambda x: (type(x) == int) and (x >= 0)),
    "restrict_max_min_width_height_ratio": (-1.0, lambda x: (x == -1.0) or (x > 0.0)),
}, **_spline_common))

_register("x", "jammy_flows_amd.layers.euclidean.euclidean_do_nothing", "euclidean_do_nothing", "e", {"add_offset": (0, [0, 1])})
_register("y", "jammy_flows_amd.layers.spheres.spherical_do_nothing", "spherical_do_nothing", "s", {"add_rotation": (0, [0, 1])})
_register("z", "jammy_flows_amd.layers.intervals.interval_do_nothing", "interval_do_nothing", "i", {})

OUT_OF_SCOPE = {"h": "deprecated Gaussianization flow", "c": "manifold continuous NF (needs torchdiffeq)",
                "u": "simplex flow", "w": "simplex flow"}


def canonical(letter):
    return ALIASES.get(letter, letter)


def obtain_default_options(flow_abbrevation):
    """dict of default options of a layer letter (flow_options.py:242-257)."""
    letter = canonical(flow_abbrevation)
    if letter in OUT_OF_SCOPE:
        raise NotImplementedError("flow layer '%s' (%s) is outside the MI355X hot path and has no kernel" % (letter, OUT_OF_SCOPE[letter]))
    assert letter in opts_dict, "Unknown flow abbreviation for default options: %s" % flow_abbrevation
    return {k: v[0] for k, v in opts_dict[letter]["kwargs"].items()}


def check_flow_option(flow_abbrevation, opt_name, opt_val):
    """validate one option value (flow_options.py:259-274)."""
    letter = canonical(flow_abbrevation)
    assert letter in opts_dict, "flow abbreviation %s not found in options dict" % flow_abbrevation
    kw = opts_dict[letter]["kwargs"]
    assert opt_name in kw, "option name %s not found in defined options for flow %s" % (opt_name, letter)
    rule = kw[opt_name][1]
    if callable(rule):
        assert rule(opt_val), ("Lambda function check of configured option", opt_name, " failed with value ", opt_val)
    elif type(rule) == list:
        assert opt_val in rule, ("Configured option ", opt_name, " with value ", opt_val, " not part of allowed options: ", rule)
    else:
        raise Exception("Unknown value check type!", type(rule))


def layer_class(letter):
    e = opts_dict[canonical(letter)]
    try:
        mod = importlib.import_module(e["module_path"])
    except ModuleNotFoundError as err:
        raise NotImplementedError("flow layer '%s' has no HIP implementation in this build (%s)" % (letter, err))
    return getattr(mod, e["class_name"])


def obtain_overall_flow_info():
    """letter -> {"type", "module"} (flow_options.py:276-286); modules are imported lazily."""
    return {k: {"type": v["type"], "module": layer_class(k)} for k, v in opts_dict.items()}
