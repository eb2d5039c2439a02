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
