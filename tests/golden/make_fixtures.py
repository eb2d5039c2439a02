#!/usr/bin/env python3
"""Generate the golden fixtures by importing the REAL reference (thoglu/jammy_flows at /root/reference).

Runs only in the build container (the reference tree does not exist on the GPU box).  Nothing of the
reference's source is written anywhere: the output files hold construction arguments, the state_dict
tensors, the inputs, and what the reference computed for them (final outputs, the (x, log_det) pair after
every top-level layer call, every spline bin index).

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/make_fixtures.py [case-name-substring ...]

Conventions:
  * reference constructed after seed_everything(1), then .double()  (tests/test_general.py:409,496 style)
  * for pdfs that have amortisation MLPs the damped init (main/default.py:1924, everything / 1000) is undone
    by multiplying every MLP tensor except the final bias by ``mlp_scale`` so that per-sample parameter
    blocks really vary from row to row (recorded in the header)
  * inputs = some of the pdf's own samples + generator rows + adversarial rows (tails, poles, seams)
"""
import contextlib
import io
import json
import os
import random
import sys

import numpy
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

with contextlib.redirect_stdout(io.StringIO()):
    import jammy_flows  # noqa: E402
    from jammy_flows.layers import spline_fns  # noqa: E402
    from jammy_flows.amortizable_mlp import AmortizableMLP  # noqa: E402

from fixture_io import encode_opts  # noqa: E402


def seed_everything(s):
    random.seed(s)
    numpy.random.seed(s)
    torch.manual_seed(s)


# ----------------------------------------------------------------------------------------------
# case list
# ----------------------------------------------------------------------------------------------
def _f_splines():
    return {"f": {"add_vertical_rq_spline_flow": 1, "add_circular_rq_spline_flow": 1, "circular_add_rotation": 0}}


CASES = [
    # BASELINE.json configs (small batches of the same construction)
    dict(name="c1_e2_gg", pdf="e2", flow="gg"),
    dict(name="c2_e4_gggg", pdf="e4", flow="gggg"),
    dict(name="c3_e4s2e4", pdf="e4+s2+e4", flow="gggg+f+gggg", mlp_scale=1000.0),
    dict(name="c3b_e4s2e4_fsplines", pdf="e4+s2+e4", flow="gggg+f+gggg", mlp_scale=300.0,
         kwargs=dict(options_overwrite=_f_splines())),
    dict(name="c4_i1s1_ro", pdf="i1+s1", flow="r+o", mlp_scale=1000.0, perturb=0.5),
    dict(name="c5_e8s2_ggggv", pdf="e8+s2", flow="gggg+v", mlp_scale=30.0,
         kwargs=dict(conditional_input_dim=16, amortization_mlp_use_custom_mode=True,
                     amortization_mlp_dims="128", amortization_mlp_ranks=8)),
    # g variants
    dict(name="g_e1_g", pdf="e1", flow="g"),
    dict(name="g_e3_ggg_cond", pdf="e3", flow="ggg", mlp_scale=1000.0, kwargs=dict(conditional_input_dim=2)),
    dict(name="g_e2_precise", pdf="e2", flow="gg",
         kwargs=dict(options_overwrite={"g": {"inverse_function_type": "inormal_partly_precise"}})),
    dict(name="g_e2_crude", pdf="e2", flow="gg",
         kwargs=dict(options_overwrite={"g": {"inverse_function_type": "inormal_partly_crude"}})),
    dict(name="g_e2_fullpade", pdf="e2", flow="gg",
         kwargs=dict(options_overwrite={"g": {"inverse_function_type": "inormal_full_pade"}})),
    dict(name="g_e2_softplusw", pdf="e2", flow="gg", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"g": {"softplus_for_width": 1}})),
    dict(name="g_e2_clampw", pdf="e2", flow="gg", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"g": {"clamp_widths": 1}})),
    dict(name="g_e2_nosat", pdf="e2", flow="gg", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"g": {"upper_bound_for_widths": -1, "clamp_widths": 1,
                                                                     "width_smooth_saturation": 0}})),
    dict(name="g_e3_nonorm_hh2", pdf="e3", flow="gg", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"g": {"fit_normalization": 0, "num_householder_iter": 2}})),
    dict(name="g_e3_norot_noreg", pdf="e3", flow="gg", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"g": {"rotation_mode": "none", "regulate_normalization": 0}})),
    dict(name="g_e3_rqs", pdf="e3", flow="gg", kwargs=dict(options_overwrite={"g": {"nonlinear_stretch_type": "rq_splines"}})),
    dict(name="g_e3_rqs_cond", pdf="e3", flow="gg", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"g": {"nonlinear_stretch_type": "rq_splines"}})),
    # more than 16 bins per dimension (round 6: the kernels' knot tables follow the chain's own bin count, cap 64)
    dict(name="g_e3_rqs_bins32", pdf="e3", flow="gg", kwargs=dict(options_overwrite={"g": {"nonlinear_stretch_type": "rq_splines", "num_kde": 32}})),
    dict(name="g_e2_rqs_bins24_cond", pdf="e2", flow="g", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"g": {"nonlinear_stretch_type": "rq_splines", "num_kde": 24}})),
    dict(name="g_e1e2e1_cond", pdf="e1+e2+e1", flow="gg+g+ggg", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, amortization_mlp_dims="64-30")),
    dict(name="g_e1e2e1_cond_lowrank", pdf="e1+e2+e1", flow="gg+g+ggg", mlp_scale=30.0,
         kwargs=dict(conditional_input_dim=2, amortization_mlp_dims="64-30", amortization_mlp_ranks="2-10-1000",
                     amortization_mlp_use_custom_mode=True)),
    # the non-default mixture / rotation options of 'g' (general-option kernel, csrc/jf_gf_ext.h)
    dict(name="g_e2_skew_cond", pdf="e2", flow="gg", mlp_scale=300.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"g": {"add_skewness": 1}})),
    dict(name="g_e3_center_mean", pdf="e3", flow="gg", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"g": {"center_mean": 1}})),
    dict(name="g_e3_rot_angles", pdf="e3", flow="gg", perturb=0.5, kwargs=dict(options_overwrite={"g": {"rotation_mode": "angles"}})),
    dict(name="g_e2_rot_cayley_cond", pdf="e2", flow="gg", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"g": {"rotation_mode": "cayley"}})),
    dict(name="g_e3_rot_triangular", pdf="e3", flow="gg", mlp_scale=300.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"g": {"rotation_mode": "triangular_combination"}})),
    dict(name="g_e4_all_options", pdf="e4", flow="ggg", perturb=0.3,
         kwargs=dict(options_overwrite={"g": {"rotation_mode": "angles", "center_mean": 1, "add_skewness": 1, "num_kde": 7}})),
    # affine / multivariate-normal layer 't' (docs/source/usage/suggested_settings.rst:12-42 recommends "gggt")
    dict(name="t_e3_gggt", pdf="e3", flow="gggt", perturb=0.4),
    dict(name="t_e3_gt_full_cond", pdf="e3", flow="gt", mlp_scale=300.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"t": {"cov_type": "full"}})),
    dict(name="t_e2_tt_variants", pdf="e2+e2", flow="t+tg", mlp_scale=300.0,
         kwargs=dict(options_overwrite={0: {"t": {"cov_type": "diagonal_symmetric", "softplus_for_width": 1}},
                                        1: {"t": {"cov_type": "full", "clamp_widths": 1, "skip_model_offset": 1}}})),
    # more than 8 Euclidean dimensions (examples/jammy_flows.py:308 defaults to e10 / ggggg; tests/test_general.py:364 lists e10 / t entries):
    # groups of 16 / 32 lanes per row in the 'g' kernels, 16- / 32-coordinate instantiations of the 't' kernels
    dict(name="g_e10_gg", pdf="e10", flow="gg", perturb=0.3),
    dict(name="g_e10_ggggg", pdf="e10", flow="ggggg", perturb=0.2),
    dict(name="g_e12_cond", pdf="e12", flow="gg", mlp_scale=300.0, kwargs=dict(conditional_input_dim=3)),
    dict(name="g_e20_g", pdf="e20", flow="g", perturb=0.3),
    # more than 32 Euclidean dimensions: a whole wave per row in the 'g' kernels (round 6)
    dict(name="g_e40_gg", pdf="e40", flow="gg", perturb=0.2),
    dict(name="g_e64_g_cond", pdf="e64", flow="g", mlp_scale=100.0, kwargs=dict(conditional_input_dim=2, amortization_mlp_dims="16")),
    dict(name="t_e10_full", pdf="e10", flow="t", perturb=0.4, kwargs=dict(options_overwrite={"t": {"cov_type": "full"}})),
    dict(name="t_e10_diagonal", pdf="e10", flow="gt", perturb=0.4, kwargs=dict(options_overwrite={"t": {"cov_type": "diagonal"}})),
    dict(name="t_e10_diagonal_symmetric", pdf="e10", flow="t", perturb=0.4, kwargs=dict(options_overwrite={"t": {"cov_type": "diagonal_symmetric"}})),
    dict(name="t_e10_identity", pdf="e10", flow="gt", perturb=0.4, kwargs=dict(options_overwrite={"t": {"cov_type": "identity"}})),
    dict(name="t_e12_full_cond", pdf="e12", flow="gt", mlp_scale=300.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"t": {"cov_type": "full"}})),
    # interval splines
    dict(name="r_i1", pdf="i1", flow="r", perturb=0.7),
    dict(name="r_i1_m1p1_rr_cond", pdf="i1_-1.0_1.0", flow="rr", mlp_scale=1000.0, kwargs=dict(conditional_input_dim=2)),
    dict(name="r_i1_smooth2", pdf="i1_-1.0_1.0", flow="rr", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"r": {"num_basis_functions": 2, "smooth_second_derivative": 1}})),
    dict(name="r_i1_smooth3", pdf="i1_-1.0_1.0", flow="r", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"r": {"num_basis_functions": 3, "smooth_second_derivative": 1,
                                                                     "fix_boundary_derivatives": 1.0}})),
    dict(name="r_i1_fixopts", pdf="i1", flow="rr", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"r": {"fix_boundary_derivatives": 1.0,
                                                                     "fix_first_width_n_height_to_zero": 1,
                                                                     "independent_width_height_parametrization": 1,
                                                                     "restrict_max_min_width_height_ratio": 20.0}})),
    # circle
    dict(name="o_s1", pdf="s1", flow="o", perturb=0.7),
    dict(name="o_s1_cond_oo", pdf="s1", flow="oo", mlp_scale=1000.0, kwargs=dict(conditional_input_dim=2)),
    dict(name="o_s1_nosmooth", pdf="s1", flow="o", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"o": {"smooth_second_derivative": 0, "num_basis_functions": 4}})),
    dict(name="o_s1_nat0_norot", pdf="s1", flow="o", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"o": {"natural_direction": 0, "add_rotation": 0}})),
    dict(name="m_s1", pdf="s1", flow="m"),
    dict(name="m_s1_cond", pdf="s1", flow="mm", mlp_scale=1000.0, kwargs=dict(conditional_input_dim=2)),
    dict(name="m_s1_nat1_rot", pdf="s1", flow="m", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"m": {"natural_direction": 1, "add_rotation": 1}})),
    # 2-sphere
    dict(name="f_s2", pdf="s2", flow="f"),
    dict(name="f_s2_cond_ff", pdf="s2", flow="ff", mlp_scale=1000.0, kwargs=dict(conditional_input_dim=2)),
    dict(name="f_s2_splines", pdf="s2", flow="f", perturb=0.5, kwargs=dict(options_overwrite=_f_splines())),
    dict(name="f_s2_splines_cond", pdf="s2", flow="ff", mlp_scale=300.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"f": dict(_f_splines()["f"], circular_flow_defs="ooo",
                                                                          vertical_fix_first_width_n_height_to_zero=1)})),
    dict(name="f_s2_identity_region", pdf="s2", flow="f", mlp_scale=300.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"f": dict(_f_splines()["f"], boundary_cos_theta_identity_region=0.4)})),
    dict(name="f_s2_correlated", pdf="s2", flow="f", mlp_scale=100.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"f": {"add_correlated_rq_spline_flow": 1}})),
    # rotation modes of the sphere base class and kappa parametrisations of 'f' (tests/test_general.py:124-177 of the reference)
    dict(name="f_s2_rot_angles", pdf="s2", flow="f", mlp_scale=300.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"f": {"rotation_mode": "angles", "kappa_prediction": "softplus_real_bounded"}})),
    dict(name="f_s2_rot_xyz_mu", pdf="s2", flow="ff", mlp_scale=300.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"f": {"rotation_mode": "xyz", "kappa_prediction": "mu"}})),
    dict(name="f_s2_rot_quat_sq", pdf="s2", flow="f", mlp_scale=300.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"f": dict(_f_splines()["f"], rotation_mode="quaternion",
                                                                          kappa_prediction="quatvec_squared")})),
    dict(name="f_s2_kappa_logb_clamp", pdf="s2", flow="f", perturb=0.5,
         kwargs=dict(options_overwrite={"f": {"kappa_prediction": "log_bounded", "kappa_clamping": 1, "rotation_mode": "quaternion"}})),
    dict(name="f_s2_extra_rot", pdf="s2", flow="ff", mlp_scale=300.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"f": dict(_f_splines()["f"], add_extra_rotation_inbetween=1)})),
    dict(name="f_s2_emb", pdf="s2", flow="f", mlp_scale=1000.0, embedding=True, kwargs=dict(conditional_input_dim=2)),
    dict(name="v_s2", pdf="s2", flow="v", B=96),
    dict(name="v_s2_cond_vv", pdf="s2", flow="vv", mlp_scale=300.0, B=96, kwargs=dict(conditional_input_dim=2)),
    dict(name="v_s2_nat1_rot", pdf="s2", flow="v", mlp_scale=300.0, B=96,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"v": {"natural_direction": 1, "add_rotation": 1}})),
    # (no_bins: the potential's spline searches once per component and Newton iteration -- an internal of the exponential map, not a layer output)
    dict(name="v_s2_splines_cond", pdf="s2", flow="v", mlp_scale=100.0, B=96, no_bins=True,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"v": {"exp_map_type": "splines"}})),
    # more than 16 bins (round 5: the 'r' / 'o' / nested-'f' kernels size a lane's knot table by the chain's own bin count, up to 64)
    dict(name="r_i1_bins24_cond", pdf="i1_-1.0_1.0", flow="rr", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"r": {"num_basis_functions": 24}})),
    dict(name="r_i1_bins40", pdf="i1", flow="r", perturb=0.7, kwargs=dict(options_overwrite={"r": {"num_basis_functions": 40}})),
    dict(name="o_s1_bins20_cond", pdf="s1", flow="oo", mlp_scale=1000.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"o": {"smooth_second_derivative": 0, "num_basis_functions": 20}})),
    dict(name="f_s2_splines_bins24", pdf="s2", flow="f", mlp_scale=300.0,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"f": dict(_f_splines()["f"], spline_num_basis_functions=24)})),
    # spline potentials with the log-prob in the SOLVING direction (natural_direction = 1: sphere Newton on a C1 potential), 10 components
    dict(name="v_s2_splines_nat1", pdf="s2", flow="v", mlp_scale=100.0, B=96, no_bins=True,
         kwargs=dict(conditional_input_dim=2, options_overwrite={"v": {"exp_map_type": "splines", "natural_direction": 1, "num_components": 10}})),
    # mixed extra
    dict(name="mix_e2s1i1", pdf="e2+s1+i1_-2.0_3.0", flow="gg+m+rr", mlp_scale=1000.0, kwargs=dict(conditional_input_dim=3)),
    dict(name="mix_s2e2_emb", pdf="s2+e2", flow="f+gg", mlp_scale=1000.0, embedding=True),
]


# ----------------------------------------------------------------------------------------------
def generator_rows(pdf_defs, n, rng):
    """synthetic target rows in intrinsic coordinates, one column block per sub-manifold."""
    cols = []
    for sub in pdf_defs.split("+"):
        kind = sub[0]
        dim = int(sub.split("_")[0][1:])
        if kind == "e":
            cols.append(rng.normal(size=(n, dim)) * 1.5)
        elif kind == "s" and dim == 2:
            th = numpy.arccos(rng.uniform(-1, 1, size=(n, 1))).clip(1e-3, numpy.pi - 1e-3)
            ph = rng.uniform(0, 2 * numpy.pi, size=(n, 1))
            cols.append(numpy.concatenate([th, ph], axis=1))
        elif kind == "s" and dim == 1:
            cols.append(rng.uniform(0, 2 * numpy.pi, size=(n, 1)))
        elif kind == "i":
            parts = sub.split("_")
            lo, hi = (0.0, 1.0) if len(parts) == 1 else (float(parts[1]), float(parts[2]))
            w = hi - lo
            cols.append(rng.uniform(lo + 1e-6 * w, hi - 1e-6 * w, size=(n, 1)))
        else:
            raise ValueError(sub)
    return numpy.concatenate(cols, axis=1)


def adversarial_rows(pdf_defs):
    """tails, poles, seams, interval ends (a handful of rows)."""
    per_sub = []
    for sub in pdf_defs.split("+"):
        kind = sub[0]
        dim = int(sub.split("_")[0][1:])
        if kind == "e":
            vals = [0.0, 50.0, -50.0, 12.0, -12.0, 6.5, -6.5, 3.0]
            per_sub.append(numpy.array([[v * (1.0 if d % 2 == 0 else -0.7) for d in range(dim)] for v in vals]))
        elif kind == "s" and dim == 2:
            pi = numpy.pi
            per_sub.append(numpy.array([[0.0, 0.0], [pi, 2 * pi], [pi / 2, pi], [1e-9, 3.0], [pi - 1e-9, 1e-9], [0.3, 2 * pi],
                                        [2.8, 0.0], [1.0, pi + 1e-7]]))
        elif kind == "s" and dim == 1:
            pi = numpy.pi
            # (exactly pi is left out: the Moebius layer's atan2 branch is decided by last-bit rounding there, also in the reference)
            per_sub.append(numpy.array([[0.0], [2 * pi], [3.0], [1e-9], [2 * pi - 1e-9], [pi + 1e-7], [pi - 1e-7], [0.5]]))
        elif kind == "i":
            parts = sub.split("_")
            lo, hi = (0.0, 1.0) if len(parts) == 1 else (float(parts[1]), float(parts[2]))
            w = hi - lo
            per_sub.append(numpy.array([[lo + f * w] for f in [1e-9, 1 - 1e-9, 0.5, 1e-4, 1 - 1e-4, 0.25, 0.75, 0.999]]))
    return numpy.concatenate(per_sub, axis=1)


class Recorder:
    def __init__(self):
        self.layers = []
        self.bins = []
        self.active = None

    def start(self, which):
        self.active = which
        self.layers = []
        self.bins = []


def install_hooks(pdf, rec):
    for bi, block in enumerate(pdf.layer_list):
        for li, layer in enumerate(block):
            tag = "%d.%d" % (bi, li)

            def wrap(fn, tag=tag):
                def inner(inputs, *a, **kw):
                    out = fn(inputs, *a, **kw)
                    rec.layers.append((tag, out[0].detach().clone().numpy(), out[1].detach().clone().numpy()))
                    return out
                return inner
            layer.inv_flow_mapping = wrap(layer.inv_flow_mapping)
            layer.flow_mapping = wrap(layer.flow_mapping)

    orig = spline_fns.searchsorted

    def ss(bin_locations, inputs, eps=1e-6):
        out = orig(bin_locations, inputs, eps=eps)
        rec.bins.append(out.detach().clone().numpy().astype(numpy.int64))
        return out
    spline_fns.searchsorted = ss
    return orig


def scale_mlps(pdf, scale):
    """undo the /damping_factor of main/default.py:1924 / amortizable_mlp.py:452 on everything but the final bias."""
    for mlp in pdf.mlp_predictors:
        if mlp is None:
            continue
        if isinstance(mlp, AmortizableMLP):
            last = mlp.sub_mlp_structures["mlp_list"][-1]
            if "linear_highway" in mlp.sub_mlp_structures:
                last = mlp.sub_mlp_structures["linear_highway"]
            nb = last["num_b_s"][-1]
            with torch.no_grad():
                mlp.u_v_b_pars.data[0, :-nb] *= scale
        else:
            lin = [m for m in mlp if hasattr(m, "weight")]
            with torch.no_grad():
                for i, m in enumerate(lin):
                    m.weight.data *= scale
                    if i < len(lin) - 1:
                        m.bias.data *= scale


def make_case(case):
    name = case["name"]
    kwargs = dict(case.get("kwargs", {}))
    B = case.get("B", 192)
    seed_everything(1)
    with contextlib.redirect_stdout(io.StringIO()):
        pdf = jammy_flows.pdf(case["pdf"], case["flow"], **kwargs)
    pdf.double()
    if case.get("mlp_scale"):
        scale_mlps(pdf, case["mlp_scale"])
    if case.get("perturb"):
        # flat default inits (r: 0.54, o: 0) would make the unconditional splines trivial: jitter the permanent layer params
        g = torch.Generator().manual_seed(99)
        with torch.no_grad():
            for prm in pdf.layer_list.parameters():
                prm.data += case["perturb"] * torch.randn(prm.shape, generator=g, dtype=prm.dtype)
    embedding = bool(case.get("embedding", False))

    rng = numpy.random.default_rng(1234)
    cdim = kwargs.get("conditional_input_dim", None)
    cond = None
    if cdim is not None:
        cond = torch.from_numpy(rng.normal(size=(B, cdim)))

    # ---- inputs: own samples, generator rows, adversarial rows (intrinsic coordinates)
    n_adv = 8
    n_own = (B - n_adv) // 2
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        if cond is None:
            own = pdf.sample(samplesize=n_own, seed=7)[0]
        else:
            own = pdf.sample(conditional_input=cond[:n_own], seed=7)[0]
    x = numpy.concatenate([own.numpy(), generator_rows(case["pdf"], B - n_own - n_adv, rng), adversarial_rows(case["pdf"])], axis=0)
    x = torch.from_numpy(x)
    if embedding:
        x, _ = pdf.transform_target_space(x, transform_from="default", transform_to="embedding")

    rec = Recorder()
    orig_ss = install_hooks(pdf, rec)
    out = {}
    try:
        rec.start("inv")
        xc = x.clone()
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            logp, logp_base, base = pdf(xc, conditional_input=cond, force_embedding_coordinates=embedding)
        assert torch.equal(xc, x)
        inv_layers, inv_bins = rec.layers, rec.bins

        z = rng.normal(size=(B, pdf.total_base_dim))
        z[-4:] = numpy.sign(z[-4:]) * numpy.array([[3.0], [4.0], [0.01], [5.0]])
        z = torch.from_numpy(z)
        rec.start("fwd")
        buf = io.StringIO()
        with torch.no_grad(), contextlib.redirect_stdout(buf):
            sx, _, slogp, slogp_base = pdf._obtain_sample(conditional_input=cond, predefined_target_input=z.clone(),
                                                         force_embedding_coordinates=embedding)
        fwd_layers, fwd_bins = rec.layers, rec.bins
        newton_msgs = buf.getvalue()
    finally:
        spline_fns.searchsorted = orig_ss

    meta = dict(
        name=name, pdf_defs=case["pdf"], flow_defs=case["flow"],
        kwargs={k: (encode_opts(v) if k == "options_overwrite" else v) for k, v in kwargs.items()},
        embedding=embedding, mlp_scale=case.get("mlp_scale"), perturb=case.get("perturb"), dtype="float64",
        torch_version=torch.__version__, numpy_version=numpy.__version__, reference="thoglu/jammy_flows v1.1.0",
        B=B, total_base_dim=int(pdf.total_base_dim), total_target_dim=int(pdf.total_target_dim),
        total_target_dim_intrinsic=int(pdf.total_target_dim_intrinsic), total_target_dim_embedded=int(pdf.total_target_dim_embedded),
        layer_param_nums=[[int(l.total_param_num) for l in blk] for blk in pdf.layer_list],
        layer_types=[[type(l).__name__ for l in blk] for blk in pdf.layer_list],
        count_parameters=int(pdf.count_parameters()),
        n_trace_inv=len(inv_layers), n_trace_fwd=len(fwd_layers), n_bins_inv=0 if case.get("no_bins") else len(inv_bins), n_bins_fwd=0 if case.get("no_bins") else len(fwd_bins),
        trace_tags_inv=[t for t, _, _ in inv_layers], trace_tags_fwd=[t for t, _, _ in fwd_layers],
        sample_warnings=newton_msgs[:300],
    )
    out["meta"] = numpy.array(json.dumps(meta))
    for k, v in pdf.state_dict().items():
        out["sd/" + k] = v.detach().numpy()
    out["x"] = x.numpy()
    if cond is not None:
        out["cond"] = cond.numpy()
    out["logp"] = logp.numpy()
    out["logp_base"] = logp_base.numpy()
    out["base"] = base.numpy()
    out["z"] = z.numpy()
    out["sample_x"] = sx.numpy()
    out["sample_logp"] = slogp.numpy()
    out["sample_logp_base"] = slogp_base.numpy()
    for d, lay, bins in (("inv", inv_layers, inv_bins), ("fwd", fwd_layers, fwd_bins)):
        for i, (_, lx, ld) in enumerate(lay):
            out["trace_%s/%d/x" % (d, i)] = lx
            out["trace_%s/%d/ld" % (d, i)] = ld
        for i, b in enumerate(bins if not case.get("no_bins") else []):
            out["bins_%s/%d" % (d, i)] = b
    path = os.path.join(HERE, name + ".npz")
    numpy.savez_compressed(path, **out)
    nonfin = int((~numpy.isfinite(logp.numpy())).sum())
    print("%-28s B=%d P=%s logp[min,max]=(%.3f, %.3f) nonfinite=%d bytes=%d" % (
        name, B, meta["layer_param_nums"], numpy.nanmin(logp.numpy()), numpy.nanmax(logp.numpy()), nonfin, os.path.getsize(path)))
    if newton_msgs.strip():
        print("    sample() printed:", newton_msgs.strip().splitlines()[0][:120])


if __name__ == "__main__":
    sel = sys.argv[1:]
    for c in CASES:
        if sel and not any(s in c["name"] for s in sel):
            continue
        make_case(c)
