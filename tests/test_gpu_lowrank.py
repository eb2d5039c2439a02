"""Training on a low-rank last MLP stage without the (B, P) blocks (csrc/jf_lowrank_gf.h: jf_lowrank_gf_chain_inv[_bwd]_f64,
autograd.LowRankGfChainFn) against the (B, P)-block sequence it replaces -- the path the reference-autograd fixtures pinned in rounds 2-3
(tests/test_gpu_grad.py runs the C5 fixture through the new path by default): ragged batches, every dimension / rank / reflection count /
offset option, rows in the far tails (log-space branch of the adjoint), non-contiguous inputs, and the kernels that actually ran."""
import numpy as np
import pytest
import torch

import jammy_flows_amd
from jammy_flows_amd import _hip, autograd

pytestmark = pytest.mark.gpu


def _pdf(D, flow, rank, cond=6, hidden=64, **opts):
    torch.manual_seed(D * 100 + rank)
    kw = dict(conditional_input_dim=cond, amortization_mlp_use_custom_mode=True, amortization_mlp_dims=str(hidden), amortization_mlp_ranks=rank)
    if opts:
        kw["options_overwrite"] = {"g": opts}
    pdf = jammy_flows_amd.pdf("e%d" % D, flow, **kw).double().cuda()
    with torch.no_grad():                                      # away from the initial point: every parameter section takes part
        for p in pdf.parameters():
            p.add_(0.3 * torch.randn_like(p))
    return pdf


def _grads(pdf, x, c, flag):
    pdf.lowrank_chain_training = flag
    for m in pdf.mlp_predictors:                               # flag False: round 3's sequence throughout (per-stage MLP launches too)
        if hasattr(m, "head_one_launch"):
            m.head_one_launch = flag
    for p in pdf.parameters():
        p.grad = None
    xs = x.clone().requires_grad_(True)
    cs = c.clone().requires_grad_(True)
    timer = _hip.KernelTimer()
    with torch.enable_grad(), timer:
        logp = pdf(xs, conditional_input=cs)[0]
        loss = -(logp * torch.linspace(0.5, 1.5, x.shape[0], dtype=x.dtype, device=x.device)).sum()       # a different weight on every row
        loss.backward()
    out = {"logp": logp.detach(), "x": xs.grad, "c": cs.grad}
    for k, p in pdf.named_parameters():
        if p.grad is not None:
            out["p:" + k] = p.grad.clone()
    return out, {k[0] for k in timer.summary()}


def _compare(a, b, tol):
    assert set(a) == set(b)
    worst = 0.0
    for k in b:
        scale = max(b[k].abs().max().item(), 1e-12)
        err = (a[k] - b[k]).abs().max().item() / scale
        worst = max(worst, err)
        assert err < tol, (k, err)
    return worst


@pytest.mark.parametrize("B", [1, 15, 16, 17, 129, 4099])
def test_lowrank_chain_equals_the_block_sequence_ragged_batches(B):
    pdf = _pdf(8, "gggg", 8, cond=16, hidden=128)
    x = torch.randn(B, 8, dtype=torch.float64, device="cuda") * 1.5
    c = torch.randn(B, 16, dtype=torch.float64, device="cuda")
    a, names_a = _grads(pdf, x, c, True)
    b, names_b = _grads(pdf, x, c, False)
    assert "jf_lowrank_gf_chain_inv_f64" in names_a and "jf_lowrank_gf_chain_inv_bwd_f64" in names_a, names_a
    assert not any(n.startswith("jf_gf_chain_inv") for n in names_a), names_a            # no (B, P) block anywhere
    assert "jf_gf_chain_inv_bwd_f64" in names_b and not any("lowrank" in n for n in names_b), names_b
    worst = _compare(a, b, 2e-11)
    print("B %d: worst relative difference %.2e over %d tensors" % (B, worst, len(b)))


@pytest.mark.parametrize("D,flow,rank,opts", [
    (1, "g", 1, {}), (2, "gg", 3, {}), (3, "ggg", 8, {}), (4, "gg", 5, {}), (5, "g", 2, {}), (7, "gggg", 7, {}), (8, "ggggg", 4, {}),
    (4, "gg", 8, {"skip_model_offset": 1}), (6, "gg", 6, {"num_householder_iter": 2}), (8, "g", 8, {"num_householder_iter": 1}),
    (3, "gg", 8, {"inverse_function_type": "isigmoid"}), (5, "gg", 8, {"inverse_function_type": "inormal_full_pade"}),
    (6, "g", 8, {"inverse_function_type": "inormal_partly_crude", "skip_model_offset": 1}),
])
def test_lowrank_chain_dimensions_ranks_and_layer_options(D, flow, rank, opts):
    opts = dict(opts)
    opts.setdefault("num_kde", 10)
    opts.setdefault("fit_normalization", 1)
    pdf = _pdf(D, flow, rank, **opts)
    B = 333
    x = torch.randn(B, D, dtype=torch.float64, device="cuda") * 1.5
    c = torch.randn(B, 6, dtype=torch.float64, device="cuda")
    a, names_a = _grads(pdf, x, c, True)
    b, _ = _grads(pdf, x, c, False)
    assert "jf_lowrank_gf_chain_inv_bwd_f64" in names_a, names_a
    _compare(a, b, 2e-11)


@pytest.mark.parametrize("K1,H,r1,r2,N,B", [(16, 128, 8, 8, 1224, 4099), (24, 128, 8, 8, 50, 777), (1, 16, 1, 1, 3, 1), (6, 64, 4, 3, 40, 17), (32, 128, 8, 5, 64, 130),
                                            (17, 48, 2, 8, 9, 33), (9, 112, 7, 7, 300, 16)])
def test_lowrank_head_equals_the_stage_sequence(K1, H, r1, r2, N, B):
    """AmortizableMLP in gradient mode: head in one launch forward / backward (jf_lowrank_head[_bwd]_f64, csrc/jf_lowrank_mlp.h) against the
    per-stage launches -- output, gradient of the input and of every weight; ragged batches, 1..32 inputs, 16..128 hidden units, ranks 1..8"""
    from jammy_flows_amd.amortizable_mlp import AmortizableMLP
    torch.manual_seed(K1 * 1000 + H)
    mlp = AmortizableMLP(K1, str(H), N, low_rank_approximations=[r1, r2], use_permanent_parameters=True).double().cuda()
    with torch.no_grad():
        mlp.u_v_b_pars.add_(0.2 * torch.randn_like(mlp.u_v_b_pars))
    s1, s2 = mlp.stages
    if s1["full"] or s2["full"]:
        pytest.skip("the rank is not smaller than the full matrix: stage stored full (amortizable_mlp.py:272-375)")
    x = torch.randn(B, K1, dtype=torch.float64, device="cuda")
    w = torch.randn(B, N, dtype=torch.float64, device="cuda")
    res = {}
    for flag in (True, False):
        mlp.head_one_launch = flag
        mlp.u_v_b_pars.grad = None
        xs = x.clone().requires_grad_(True)
        timer = _hip.KernelTimer()
        with torch.enable_grad(), timer:
            out = mlp(xs)
            (out * w).sum().backward()
        res[flag] = (out.detach(), xs.grad, mlp.u_v_b_pars.grad.clone(), {k[0] for k in timer.summary()})
    assert "jf_lowrank_head_f64" in res[True][3] and "jf_lowrank_head_bwd_f64" in res[True][3], res[True][3]
    assert not any("lowrank_head" in n for n in res[False][3])
    for a, b in zip(res[True][:3], res[False][:3]):
        assert (a - b).abs().max().item() <= 2e-12 * max(1.0, b.abs().max().item()), (a - b).abs().max().item()
    # the input needs no gradient: same weight gradients, no g_c written
    mlp.head_one_launch = True
    mlp.u_v_b_pars.grad = None
    with torch.enable_grad():
        (mlp(x) * w).sum().backward()
    assert (mlp.u_v_b_pars.grad - res[False][2]).abs().max().item() <= 2e-12 * max(1.0, res[False][2].abs().max().item())


def test_rows_in_the_far_tails_take_the_log_space_adjoint():
    """targets 30-60 widths away from every mixture component: the linear-space sums underflow, the forward re-evaluates with scaled sums and the
    adjoint takes its log-space branch for those waves -- same values and gradients as the (B, P)-block kernel's branch"""
    pdf = _pdf(8, "gg", 8, cond=16, hidden=128)
    B = 256
    x = torch.randn(B, 8, dtype=torch.float64, device="cuda")
    x[::3] *= 400.0
    x[5, 2] = 3.0e4
    x[70, :] = -2.5e3
    c = torch.randn(B, 16, dtype=torch.float64, device="cuda")
    a, _ = _grads(pdf, x, c, True)
    b, _ = _grads(pdf, x, c, False)
    assert torch.isfinite(b["logp"]).all() and torch.isfinite(a["logp"]).all()
    _compare(a, b, 1e-9)


def test_unsupported_configurations_keep_the_block_sequence():
    """float32, 5 mixture components, full-rank last stage: LowRankGfChainFn is not chosen (and the gradients are what they were)"""
    x = torch.randn(50, 4, dtype=torch.float64, device="cuda")
    c = torch.randn(50, 6, dtype=torch.float64, device="cuda")
    pdf5 = _pdf(4, "gg", 8, num_kde=5, fit_normalization=1)
    _, names = _grads(pdf5, x, c, True)
    assert not any("lowrank" in n for n in names), names
    pdf32 = _pdf(4, "gg", 8, num_kde=10, fit_normalization=1).float()
    _, names = _grads(pdf32, x.float(), c.float(), True)
    assert not any("lowrank" in n for n in names), names
    torch.manual_seed(1)
    full = jammy_flows_amd.pdf("e4", "gg", conditional_input_dim=6, options_overwrite={"g": {"num_kde": 10, "fit_normalization": 1}}).double().cuda()
    _, names = _grads(full, x, c, True)
    assert not any("lowrank" in n for n in names), names


def test_entry_points_directly_strided_inputs_and_argument_checks():
    pdf = _pdf(8, "gg", 8, cond=16, hidden=128)
    layers = list(pdf.layer_list[0])
    larr = _hip.gf_layer_array([l.c_struct() for l in layers])
    N = sum(l.total_param_num for l in layers)
    B = 77
    g = torch.Generator(device="cuda").manual_seed(3)
    t2w = torch.randn(B, 11, dtype=torch.float64, device="cuda", generator=g)
    t2 = t2w[:, 2:10]                                            # row stride 11
    u2 = torch.randn(N, 8, dtype=torch.float64, device="cuda", generator=g) * 0.2
    b2 = torch.randn(N, dtype=torch.float64, device="cuda", generator=g) * 0.5
    xw = torch.randn(B, 13, dtype=torch.float64, device="cuda", generator=g)
    x = xw[:, 1:9]
    ld = torch.randn(B, dtype=torch.float64, device="cuda", generator=g)
    z, ldo, blp, aux = _hip.lowrank_gf_chain_inv(t2, u2, b2, x, ld, larr, 2, 8, want_base_logp=True, want_aux=True)
    params = t2 @ u2.t() + b2
    zr, ldr, blpr = _hip.gf_chain("inv", x, ld, params, larr, 2, 8, want_base_logp=True)
    assert (z - zr).abs().max().item() < 1e-11 and (ldo - ldr).abs().max().item() < 1e-10 and (blp - blpr).abs().max().item() < 1e-10
    assert aux.shape == (2, 5, 2, B, 4)
    gz = torch.randn(B, 8, dtype=torch.float64, device="cuda", generator=g)
    gl = torch.randn(B, dtype=torch.float64, device="cuda", generator=g)
    gb = torch.randn(B, dtype=torch.float64, device="cuda", generator=g)
    for ups in ((gz, gl, gb), (None, gl, None), (gz, None, None), (None, None, gb)):
        g_x, g_t2, g_u2, g_b2 = _hip.lowrank_gf_chain_inv_bwd(t2, u2, b2, aux, z, larr, 2, 8, *ups)
        rx, rp = _hip.gf_chain_inv_bwd(x, params, larr, 2, 8, *ups)
        for got, ref in ((g_x, rx), (g_t2, rp @ u2), (g_u2, rp.t() @ t2), (g_b2, rp.sum(0))):
            assert (got - ref).abs().max().item() < 1e-10 * max(1.0, ref.abs().max().item())
    # twice the same launch: the float64 LDS atomics may order their additions differently, nothing else may move
    again = _hip.lowrank_gf_chain_inv_bwd(t2, u2, b2, aux, z, larr, 2, 8, gz, gl, gb)
    first = _hip.lowrank_gf_chain_inv_bwd(t2, u2, b2, aux, z, larr, 2, 8, gz, gl, gb)
    assert torch.equal(again[0], first[0]) and torch.equal(again[1], first[1])
    assert (again[2] - first[2]).abs().max().item() < 1e-12 * first[2].abs().max().item()
    # rank 9 is not supported (the caller keeps the block sequence), a float32 tensor is refused
    assert _hip.lowrank_gf_chain_inv(torch.zeros(B, 9, dtype=torch.float64, device="cuda"), torch.zeros(N, 9, dtype=torch.float64, device="cuda"), b2, x,
                                     None, larr, 2, 8) is None
    with pytest.raises(ValueError):
        _hip.lowrank_gf_chain_inv(t2.float(), u2, b2, x, None, larr, 2, 8)


def test_differentiable_sampling_through_the_lowrank_chain():
    """pdf.sample(allow_gradients=True) on the C5 configuration: _differentiable_sample runs D + 1 backward passes through ONE graph that contains
    LowRankGfChainFn and LowRankHeadFn (retain_graph) -- the saved layer inputs must survive them; gradients equal the block sequence's"""
    import fixture_io
    import helpers
    fx = fixture_io.load("c5_e8s2_ggggv")
    res = {}
    for flag in (True, False):
        pdf = helpers.build_product(fx, torch.float64)
        pdf.lowrank_chain_training = flag
        for m in pdf.mlp_predictors:
            if hasattr(m, "head_one_launch"):
                m.head_one_launch = flag
        c = torch.from_numpy(fx["cond"][:64]).cuda().requires_grad_(True)
        z = torch.from_numpy(fx["z"][:64]).cuda()
        with torch.enable_grad():
            x, _, logp, _ = pdf._differentiable_sample(conditional_input=c, predefined_target_input=z)
            loss = (x * torch.linspace(-1.0, 1.0, x.shape[1], dtype=x.dtype, device=x.device)).sum() + 0.1 * logp.sum()
        loss.backward()
        res[flag] = {"x": x.detach(), "c": c.grad.clone(), **{k: p.grad.clone() for k, p in pdf.named_parameters() if p.grad is not None}}
    _compare(res[True], res[False], 1e-9)


def test_full_size_training_step_c5():
    """2^17 rows of the C5 configuration (the bench's --train --workload c5 batch): loss and every gradient of the two paths agree; the gradient of
    the weights is the sum over 131072 rows -- tolerance relative to each tensor's largest entry"""
    import fixture_io
    import helpers
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from bench_configs_inputs import inputs
    fx = fixture_io.load("c5_e8s2_ggggv")
    pdf = helpers.build_product(fx, torch.float64)
    x64, c64 = inputs(fx, 1 << 17, 11)
    x, c = torch.from_numpy(x64).cuda(), torch.from_numpy(c64).cuda()
    a, names = _grads(pdf, x, c, True)
    b, _ = _grads(pdf, x, c, False)
    assert "jf_lowrank_gf_chain_inv_bwd_f64" in names
    worst = _compare(a, b, 1e-10)
    print("2^17 rows: worst relative difference %.2e" % worst)
