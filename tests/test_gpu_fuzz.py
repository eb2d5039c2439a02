"""GPU fuzz test (run with -m gpu): random option combinations of the 'g' layer -- widths (three regulators, clamping, bounds), weights (fitted,
regulated or not), every inverse_function_type, every rotation_mode, center_mean, add_skewness, num_kde, D = 1..8, with and without offset --
through the per-layer C-ABI entry points in both directions, per-sample and broadcast parameters, against the float64 oracle
(oracle/gf.py, itself pinned on the reference's golden fixtures).  The golden fixtures cover the options one at a time; this covers their
products, where a kernel-side option test in the wrong place would hide."""
import numpy as np
import pytest
import torch

from oracle import gf as ogf
from jammy_flows_amd import flow_options

pytestmark = pytest.mark.gpu
N_CASES = 48


def random_options(rng, D):
    o = flow_options.obtain_default_options("g")
    o["num_kde"] = int(rng.integers(1, 13))
    o["fit_normalization"] = int(rng.integers(0, 2))
    o["regulate_normalization"] = int(rng.integers(0, 2))
    o["inverse_function_type"] = str(rng.choice(["isigmoid", "inormal_partly_precise", "inormal_full_pade", "inormal_partly_crude"]))
    mode = int(rng.integers(0, 3))
    o["softplus_for_width"] = 1 if mode == 0 else 0
    o["width_smooth_saturation"] = 1 if mode == 1 else 0
    o["clamp_widths"] = int(rng.integers(0, 2))
    o["lower_bound_for_widths"] = float(rng.choice([0.01, 0.05, 0.3]))
    o["upper_bound_for_widths"] = float(rng.choice([100, 20])) if (mode == 1 or rng.integers(0, 2)) else -1
    o["lower_bound_for_norms"], o["upper_bound_for_norms"] = (1, 10) if rng.integers(0, 2) else (0.5, 4)
    rots = ["householder", "none", "angles", "triangular_combination"] + (["cayley"] if D == 2 else [])
    o["rotation_mode"] = str(rng.choice(rots))
    o["num_householder_iter"] = int(rng.choice([-1, 1, 2])) if D > 1 else -1
    o["center_mean"] = int(rng.integers(0, 2)) if o["num_kde"] > 1 else 0
    o["add_skewness"] = int(rng.integers(0, 2))
    return o


@pytest.mark.parametrize("seed", range(N_CASES))
def test_random_g_layer_options_vs_oracle(seed):
    import jammy_flows_amd
    from jammy_flows_amd.layers.euclidean import gaussianization_flow as gfl
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.integers(1, 9))
    o = random_options(rng, D)
    model_offset = int(rng.integers(0, 2))
    spec = ogf.GfSpec(D, o, model_offset)
    kw = {k: v for k, v in o.items() if k not in ("replace_first_sigmoid_with_icdf", "skip_model_offset")}
    layer = gfl.gf_block(D, use_permanent_parameters=False, model_offset=model_offset, **kw)
    assert layer.total_param_num == spec.total_param_num, (o, layer.total_param_num, spec.total_param_num)
    B = 96
    for pb in (B, 1):
        params = rng.normal(size=(pb, spec.total_param_num)) * 0.8
        x = rng.normal(size=(B, D)) * 2.0
        x[:4] *= 8.0                                             # a few rows far out in the tails
        y, ld, _ = ogf.inverse(spec, x, np.zeros(B), params)
        tp, tx = torch.from_numpy(params).cuda(), torch.from_numpy(x).cuda()
        gy, gld = layer.inv_flow_mapping([tx, torch.zeros(B, dtype=torch.float64, device="cuda")], extra_inputs=tp)
        what = "seed %d D %d pb %d %s" % (seed, D, pb, {k: o[k] for k in ("num_kde", "inverse_function_type", "rotation_mode", "center_mean",
                                                                         "add_skewness", "fit_normalization", "regulate_normalization")})
        ok = np.isfinite(y).all(axis=1) & np.isfinite(ld)
        assert ok.sum() >= B - 4, what
        err_y = np.abs(gy.cpu().numpy() - y)[ok] / (1.0 + np.abs(y[ok]))
        err_ld = np.abs(gld.cpu().numpy() - ld)[ok] / (1.0 + np.abs(ld[ok]))
        tol = 2e-5 if o["inverse_function_type"] == "inormal_full_pade" else 1e-7      # the full-Pade centre is ill-conditioned in the reference's own form
        assert err_y.max() < tol and err_ld.max() < tol, (what, float(err_y.max()), float(err_ld.max()))
        # sampling direction on moderate base points (bisection + Newton), then back
        z = rng.normal(size=(B, D))
        xs, lds, _ = ogf.forward(spec, z, np.zeros(B), params)
        gxs, glds = layer.flow_mapping([torch.from_numpy(z).cuda(), torch.zeros(B, dtype=torch.float64, device="cuda")], extra_inputs=tp)
        oks = np.isfinite(xs).all(axis=1) & np.isfinite(lds)
        assert oks.sum() >= B - 4, what
        assert (np.abs(gxs.cpu().numpy() - xs)[oks] / (1.0 + np.abs(xs[oks]))).max() < 1e-6, what
        assert (np.abs(glds.cpu().numpy() - lds)[oks] / (1.0 + np.abs(lds[oks]))).max() < 1e-6, what


# ------------------------------------------------------------------------------------------------------------------------------------------
# pdf-level fuzz: random sub-manifold structures and random option products of every layer family, product pdf vs OraclePdf on the product's
# own (randomly initialised, MLP-rescaled) state_dict; log-prob and sampling with injected base points
def _rand_layer_options(rng, letter):
    ch = lambda *v: v[int(rng.integers(0, len(v)))]
    if letter == "g":
        stretch = ch("classic", "classic", "rq_splines")
        return {"num_kde": ch(3, 6, 10), "inverse_function_type": ch("isigmoid", "inormal_partly_precise", "inormal_partly_crude"),
                "rotation_mode": ch("householder", "none", "angles", "triangular_combination"), "center_mean": ch(0, 1) if stretch == "classic" else 0,
                "regulate_normalization": ch(0, 1), "clamp_widths": ch(0, 1), "nonlinear_stretch_type": stretch}
    if letter == "t":
        return {"cov_type": ch("identity", "diagonal_symmetric", "diagonal", "full"), "softplus_for_width": ch(0, 1), "clamp_widths": ch(0, 1),
                "skip_model_offset": ch(0, 1)}
    if letter == "r":
        smooth = ch(0, 0, 1)
        o = {"smooth_second_derivative": smooth, "num_basis_functions": ch(2, 3) if smooth else ch(1, 3, 5, 8),
             "fix_boundary_derivatives": ch(-1.0, 1.0, 2.0), "independent_width_height_parametrization": ch(0, 1)}
        o["fix_first_width_n_height_to_zero"] = ch(0, 1) if o["num_basis_functions"] >= 2 else 0
        if o["num_basis_functions"] == 1:
            o["fix_boundary_derivatives"] = -1.0       # one bin with fixed boundary derivatives: the reference itself raises (no free derivative left)
        if not smooth and o["num_basis_functions"] >= 3:
            o["restrict_max_min_width_height_ratio"] = ch(-1.0, 20.0)
        return o
    if letter == "o":
        smooth = ch(0, 1)
        nb = 2 if smooth else ch(1, 2, 4, 6)
        return {"smooth_second_derivative": smooth, "num_basis_functions": nb, "add_rotation": ch(0, 1),
                "natural_direction": ch(0, 1), "fix_boundary_derivatives": ch(-1.0, 1.0) if (not smooth and nb > 1) else -1.0,
                "independent_width_height_parametrization": ch(0, 1)}
    if letter == "m":
        return {"add_rotation": ch(0, 1), "num_basis_functions": ch(1, 3, 5), "natural_direction": ch(0, 1)}
    if letter == "f":
        kp, rm = ch(("direct_log_real_bounded", None), ("softplus_real_bounded", None), ("log_bounded", None), ("mu", "xyz"), ("mu_squared", "xyz"),
                    ("quatvec", "quaternion"), ("quatvec_squared", "quaternion"))
        return {"kappa_prediction": kp, "rotation_mode": rm or ch("householder", "angles", "xyz", "quaternion"), "kappa_clamping": ch(0, 1),
                "add_vertical_rq_spline_flow": ch(0, 1), "add_circular_rq_spline_flow": ch(0, 1),
                "inverse_z_scaling": ch(0, 1), "boundary_cos_theta_identity_region": ch(0.0, 0.0, 0.3), "add_extra_rotation_inbetween": ch(0, 1),
                "spline_num_basis_functions": ch(3, 5), "vertical_fix_first_width_n_height_to_zero": ch(0, 1)}
    if letter == "v":
        return {"exp_map_type": ch("linear", "quadratic", "exponential", "splines"), "natural_direction": ch(0, 1), "add_rotation": ch(0, 1),
                "num_components": ch(1, 4, 10)}
    raise KeyError(letter)


_SUBS = [("e2", ("g", "gg", "gt", "tg")), ("e3", ("gg", "ggt")), ("i1", ("r", "rr")), ("i1_-1.0_1.0", ("r", "rr")), ("s1", ("o", "m", "om", "mo")),
         ("s2", ("f", "ff", "v"))]


def random_pdf_case(rng):
    n_sub = int(rng.integers(1, 3))
    subs, flows, ow = [], [], {}
    for si in range(n_sub):
        sub, choices = _SUBS[int(rng.integers(0, len(_SUBS)))]
        flow = choices[int(rng.integers(0, len(choices)))]
        subs.append(sub); flows.append(flow)
        ow[si] = {letter: _rand_layer_options(rng, letter) for letter in sorted(set(flow))}
    kwargs = {"options_overwrite": ow}
    if rng.integers(0, 2):
        kwargs["conditional_input_dim"] = 2
    return "+".join(subs), "+".join(flows), kwargs


def domain_rows(pdf_defs, n, rng):
    cols = []
    for sub in pdf_defs.split("+"):
        kind, parts = sub[0], sub.split("_")
        dim = int(parts[0][1:])
        if kind == "e":
            cols.append(rng.normal(size=(n, dim)) * 1.5)
        elif kind == "i":
            lo, hi = (0.0, 1.0) if len(parts) == 1 else (float(parts[1]), float(parts[2]))
            cols.append(lo + (hi - lo) * rng.uniform(0.02, 0.98, size=(n, 1)))
        elif dim == 1:
            cols.append(rng.uniform(0.1, 2 * np.pi - 0.1, size=(n, 1)))
        else:
            cols.append(np.concatenate([rng.uniform(0.1, np.pi - 0.1, size=(n, 1)), rng.uniform(0.1, 2 * np.pi - 0.1, size=(n, 1))], axis=1))
    return np.concatenate(cols, axis=1)


def build_fuzz_pdf(seed):
    import jammy_flows_amd
    rng = np.random.default_rng(5000 + seed)
    pdf_defs, flow_defs, kwargs = random_pdf_case(rng)
    torch.manual_seed(seed)
    pdf = jammy_flows_amd.pdf(pdf_defs, flow_defs, **kwargs).double()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for prm in pdf.layer_list.parameters():                          # jitter the flat default inits of the permanent layer parameters
            prm.data += 0.4 * torch.randn(prm.shape, generator=g, dtype=prm.dtype)
        for mlp in pdf.mlp_predictors:                                   # undo the /1000 damping on everything but the final bias (main/default.py:1924)
            if mlp is None:
                continue
            mods = [m for m in mlp if hasattr(m, "weight")]
            for m in mods:
                m.weight.data *= 200.0
            for m in mods[:-1]:
                m.bias.data *= 200.0
    return rng, pdf_defs, flow_defs, kwargs, pdf


@pytest.mark.parametrize("seed", range(40))
def test_random_pdf_structures_and_options_vs_oracle(seed):
    from oracle import OraclePdf
    rng, pdf_defs, flow_defs, kwargs, pdf = build_fuzz_pdf(seed)
    sd = {k: v.detach().cpu().numpy() for k, v in pdf.state_dict().items()}
    oracle = OraclePdf(pdf_defs, flow_defs, state_dict=sd, **kwargs)
    what = "seed %d %s / %s %s" % (seed, pdf_defs, flow_defs, kwargs)
    B = 64
    x = domain_rows(pdf_defs, B, rng)
    cond = rng.normal(size=(B, 2)) if "conditional_input_dim" in kwargs else None
    o_logp, _, o_base = oracle.forward(x, cond)
    pdf = pdf.cuda()
    tx = torch.from_numpy(x).cuda()
    tc = None if cond is None else torch.from_numpy(cond).cuda()
    with torch.no_grad():
        logp, _, base = pdf(tx, conditional_input=tc)
    ok = np.isfinite(o_logp)
    assert ok.sum() >= B - 2, what
    assert (np.abs(logp.cpu().numpy() - o_logp)[ok] / (1 + np.abs(o_logp[ok]))).max() < 2e-6, what
    assert (np.abs(base.cpu().numpy() - o_base)[ok] / (1 + np.abs(o_base[ok]))).max() < 2e-6, what
    z = rng.normal(size=(B, pdf.total_base_dim))
    o_x, o_slogp = oracle.sample_from_base(z, cond)[:2]
    with torch.no_grad():
        sx, _, slogp, _ = pdf._obtain_sample(conditional_input=tc, predefined_target_input=torch.from_numpy(z).cuda())
    oks = np.isfinite(o_x).all(axis=1) & np.isfinite(o_slogp)
    assert oks.sum() >= B - 2, what
    tol = 1e-4 if "v" in flow_defs else 2e-6                              # 'v': damped Newton on the sphere converges to ~1e-6 in the reference itself
    assert (np.abs(sx.cpu().numpy() - o_x)[oks] / (1 + np.abs(o_x[oks]))).max() < tol, what
    assert (np.abs(slogp.cpu().numpy() - o_slogp)[oks] / (1 + np.abs(o_slogp[oks]))).max() < tol, what
