"""fully_amortized_pdf: host bookkeeping + oracle on the CPU, HIP parity (-m gpu) against golden vectors generated from the
real reference (tests/golden/amortized/*.npz, make_amortized_fixtures.py): forward, sampling with injected base noise, gradients."""
import json
import os

import numpy as np
import pytest
import torch

import fixture_io

DIR = os.path.join(fixture_io.GOLDEN_DIR, "amortized")
NAMES = sorted(f[:-4] for f in os.listdir(DIR) if f.endswith(".npz"))


def load(name):
    with np.load(os.path.join(DIR, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    g["meta"] = json.loads(str(g["meta"]))
    return g


def build(g, dtype=torch.float64, device="cpu"):
    import jammy_flows_amd
    m = g["meta"]
    pdf = jammy_flows_amd.fully_amortized_pdf(m["pdf_defs"], m["flow_defs"], **m["kwargs"])
    sd = {k[3:]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in g.items() if k.startswith("sd/")}
    missing, unexpected = pdf.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return pdf.to(dtype=dtype, device=device)


def oracle_forward(g):
    from oracle import OraclePdf
    from oracle.mlp import AmortizableMLPSpec
    m = g["meta"]
    kw = m["kwargs"]
    inner = OraclePdf(m["pdf_defs"], m["flow_defs"], amortize_everything=True, amortization_mlp_use_custom_mode=True,
                      amortization_mlp_dims=kw.get("inner_mlp_dims_sub_pdfs", "128"), amortization_mlp_ranks=kw.get("inner_mlp_ranks", 0),
                      amortization_mlp_highway_mode=kw.get("inner_mlp_highway_mode", 1))
    hyper = AmortizableMLPSpec(kw["conditional_input_dim"], kw.get("amortization_mlp_dims", "128"), inner.total_number_amortizable_params,
                               kw.get("amortization_mlp_ranks", 5), kw.get("amortization_mlp_highway_mode", 0))
    amort = hyper.apply(g["cond"], g["sd/amortization_mlp.u_v_b_pars"].reshape(1, -1))
    return inner.forward(g["x"], None, amortization_parameters=amort), inner, amort


@pytest.mark.parametrize("name", NAMES)
def test_bookkeeping_and_state_dict_match_the_reference(name):
    g = load(name)
    pdf = build(g)
    assert pdf.pdf_to_amortize.total_number_amortizable_params == g["meta"]["total_number_amortizable_params"]
    assert pdf.count_parameters() == g["meta"]["count_parameters"]
    assert len(list(pdf.pdf_to_amortize.parameters())) == 0          # every parameter comes from the hyper-network


@pytest.mark.parametrize("name", NAMES)
def test_oracle_vs_golden(name):
    g = load(name)
    (logp, logp_base, base), inner, amort = oracle_forward(g)
    assert inner.total_number_amortizable_params == g["meta"]["total_number_amortizable_params"]
    assert np.abs(logp - g["logp"]).max() < 1e-8 * (1 + np.abs(g["logp"]).max())
    assert np.abs(base - g["base"]).max() < 1e-8
    sx, slogp, _ = inner.sample_from_base(g["z"], None, amortization_parameters=amort)
    assert np.abs(sx - g["sample_x"]).max() < 1e-7
    assert np.abs(slogp - g["sample_logp"]).max() < 1e-7 * (1 + np.abs(g["sample_logp"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_forward_sampling_and_gradients_vs_reference(name):
    g = load(name)
    pdf = build(g, torch.float64, "cuda")
    x = torch.from_numpy(g["x"]).cuda()
    cond = torch.from_numpy(g["cond"]).cuda()
    logp, logp_base, base = pdf(x, conditional_input=cond)
    assert np.abs(logp.cpu().numpy() - g["logp"]).max() < 1e-7 * (1 + np.abs(g["logp"]).max())
    assert np.abs(base.cpu().numpy() - g["base"]).max() < 1e-7
    amort = pdf.amortization_mlp(cond)
    sx, _, slogp, _ = pdf.pdf_to_amortize._obtain_sample(predefined_target_input=torch.from_numpy(g["z"]).cuda(), amortization_parameters=amort)
    assert np.abs(sx.cpu().numpy() - g["sample_x"]).max() < 1e-6
    assert np.abs(slogp.cpu().numpy() - g["sample_logp"]).max() < 1e-6 * (1 + np.abs(g["sample_logp"]).max())
    # gradients of -mean log p w.r.t. x, conditional input and the hyper-network's weights
    xg, cg = x.clone().requires_grad_(True), cond.clone().requires_grad_(True)
    with torch.enable_grad():
        loss = -pdf(xg, conditional_input=cg)[0].mean()
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-9 * max(1.0, abs(float(g["loss"])))

    def rel(a, b):
        return float(np.abs(a.detach().cpu().numpy().reshape(b.shape) - b).max()) / max(float(np.abs(b).max()), 1e-9)
    worst = {"x": rel(xg.grad, g["x_grad"]), "cond": rel(cg.grad, g["cond_grad"])}
    for k, p in pdf.named_parameters():
        worst[k] = rel(p.grad, g["pg/" + k])
    print(name, "max relative gradient error %.2e" % max(worst.values()))
    assert max(worst.values()) < 1e-7, worst
    # float32 evaluation path
    pdf32 = build(g, torch.float32, "cuda")
    l32 = pdf32(x.float(), conditional_input=cond.float())[0]
    assert np.abs(l32.double().cpu().numpy() - g["logp"]).max() < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_gpu_permanent_amortizable_mlp_highway_modes_vs_oracle(mode):
    """pdf(..., amortization_mlp_use_custom_mode=True, amortization_mlp_highway_mode=k) with PERMANENT weights (MFMA dense launches + the
    highway sums) against the oracle built from the same state_dict (amortizable_mlp.py:581-682)"""
    import jammy_flows_amd
    from oracle import OraclePdf
    torch.manual_seed(mode)
    kw = dict(conditional_input_dim=3, amortization_mlp_use_custom_mode=True, amortization_mlp_dims="24-16", amortization_mlp_ranks=5,
              amortization_mlp_highway_mode=mode)
    pdf = jammy_flows_amd.pdf("e2+s1", "gg+o", **kw).double()
    with torch.no_grad():
        for mlp in pdf.mlp_predictors:
            nb = mlp.output_dim
            mlp.u_v_b_pars.data[0, :-nb] *= 300.0
    sd = {k: v.detach().cpu().numpy() for k, v in pdf.state_dict().items()}
    pdf = pdf.cuda()
    g = torch.Generator().manual_seed(2)
    x = torch.cat([torch.randn(128, 2, generator=g, dtype=torch.float64), torch.rand(128, 1, generator=g, dtype=torch.float64) * 6.0 + 0.1], dim=1)
    c = torch.randn(128, 3, generator=g, dtype=torch.float64)
    logp = pdf(x.cuda(), conditional_input=c.cuda())[0]
    ref = OraclePdf("e2+s1", "gg+o", state_dict=sd, **kw).forward(x.numpy(), c.numpy())[0]
    assert np.abs(logp.cpu().numpy() - ref).max() < 1e-8 * (1 + np.abs(ref).max())
    # and its gradient against finite differences of the forward
    cg = c.cuda().requires_grad_(True)
    with torch.enable_grad():
        pdf(x.cuda(), conditional_input=cg)[0].sum().backward()
    eps = 1e-6
    c2, c3 = c.clone(), c.clone()
    c2[:, 1] += eps
    c3[:, 1] -= eps
    fd = (pdf(x.cuda(), conditional_input=c2.cuda())[0] - pdf(x.cuda(), conditional_input=c3.cuda())[0]) / (2 * eps)
    assert float(((cg.grad[:, 1] - fd).abs() / (1 + fd.abs())).max()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_gradients_through_sampling_reach_the_hyper_network(name):
    """reparameterised training of a fully amortised pdf (fully_amortized.py:173-215): gradients of a loss on the SAMPLES with respect to the
    hyper-network's weights and the conditional input, against the reference's autograd through its own sampling (injected base points)"""
    g = load(name)
    pdf = build(g, torch.float64, "cuda")
    z = torch.from_numpy(g["z"]).cuda()
    cs = torch.from_numpy(g["cond"]).cuda().requires_grad_(True)
    with torch.enable_grad():
        amort = pdf.amortization_mlp(cs)
        x, _, logp, _ = pdf.pdf_to_amortize._differentiable_sample(predefined_target_input=z, amortization_parameters=amort)
        w = torch.linspace(0.5, 1.5, x.shape[1], dtype=torch.float64, device="cuda")
        loss = (x * w).sum(dim=1).mean() + 0.1 * logp.mean()
    loss.backward()
    assert abs(loss.item() - float(g["sg_loss"])) < 1e-7 * max(1.0, abs(float(g["sg_loss"])))

    def rel(a, b):
        return float(np.abs(a.detach().cpu().numpy().reshape(b.shape) - b).max()) / max(float(np.abs(b).max()), 1e-9)
    worst = {"cond": rel(cs.grad, g["sg_cond"])}
    for k, p in pdf.named_parameters():
        worst[k] = rel(p.grad, g["sg/" + k])
    assert max(worst.values()) < 1e-5, worst
    # the public entry point: sample(allow_gradients=True) must connect the samples to amortization_mlp (it ran the
    # hyper-network under no_grad and handed back samples without a path to it)
    for p in pdf.parameters():
        p.grad = None
    with torch.enable_grad():
        xs, _, lp, _ = pdf.sample(conditional_input=torch.from_numpy(g["cond"]).cuda(), seed=5, allow_gradients=True)
        assert xs.requires_grad and lp.requires_grad
        (xs.sum() + lp.sum()).backward()
    grads = [p.grad for p in pdf.parameters()]
    assert all(gr is not None for gr in grads) and max(float(gr.abs().max()) for gr in grads) > 0
    with torch.no_grad():
        xn = pdf.sample(conditional_input=torch.from_numpy(g["cond"]).cuda(), seed=5)[0]
    assert not xn.requires_grad and float((xn - xs.detach()).abs().max()) < 1e-9


NONLIN = np.load(os.path.join(fixture_io.GOLDEN_DIR, "nonlin", "amlp_nonlinearities.npz"))


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [0, 1])
@pytest.mark.parametrize("name", [str(n) for n in NONLIN["names"]])
def test_gpu_amortizable_mlp_nonlinearities_vs_reference(name, hw):
    """every nonlinearity the reference offers (extra_functions.py:81-89; only tanh is used by pdf): outputs and autograd gradients of a
    low-rank AmortizableMLP against vectors from the reference (tests/golden/make_amlp_nonlin_fixtures.py)"""
    from jammy_flows_amd.amortizable_mlp import AmortizableMLP
    k = "%s_hw%d" % (name, hw)
    mlp = AmortizableMLP(5, "12-9", 7, low_rank_approximations=3, nonlinearity=name, highway_mode=hw, use_permanent_parameters=True).double().cuda()
    with torch.no_grad():
        mlp.u_v_b_pars.copy_(torch.from_numpy(NONLIN[k + "/pars"]).cuda())
    x = torch.from_numpy(NONLIN["x"]).cuda().requires_grad_(True)
    with torch.enable_grad():
        y = mlp(x)
        loss = (y ** 2).mean()
    loss.backward()

    def rel(a, b):
        return float(np.abs(a.detach().cpu().numpy().reshape(b.shape) - b).max()) / max(float(np.abs(b).max()), 1e-30)
    assert rel(y, NONLIN[k + "/y"]) < 1e-10, k
    assert rel(x.grad, NONLIN[k + "/gx"]) < 1e-8, k
    assert rel(mlp.u_v_b_pars.grad, NONLIN[k + "/gp"]) < 1e-8, k
    with torch.no_grad():
        assert rel(mlp(x.detach()), NONLIN[k + "/y"]) < 1e-10      # the no-grad path (plain launches)


PRECISE = np.load(os.path.join(fixture_io.GOLDEN_DIR, "nonlin", "amlp_precise_structure.npz"))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["hw0", "hw2", "hw4"])
def test_gpu_amortizable_mlp_precise_structure_vs_reference(case):
    """AmortizableMLP(precise_mlp_structure=...) (amortizable_mlp.py:20, 56-62: the sub-MLP table handed in by the caller -- per-matrix ranks and
    svd modes, widths no `hidden_dims` string produces): parameter count, outputs and autograd gradients against vectors from the reference
    (tests/golden/make_amlp_precise_fixtures.py)"""
    import json
    from jammy_flows_amd.amortizable_mlp import AmortizableMLP
    spec = json.loads(str(PRECISE["structures"]))[case]
    mlp = AmortizableMLP(6, "3", 4, highway_mode=spec["highway_mode"], nonlinearity="tanh", use_permanent_parameters=True,
                         precise_mlp_structure=spec["structure"]).double().cuda()
    assert mlp.num_amortization_params == PRECISE[case + "/pars"].shape[1]
    with torch.no_grad():
        mlp.u_v_b_pars.copy_(torch.from_numpy(PRECISE[case + "/pars"]).cuda())
    x = torch.from_numpy(PRECISE["x"]).cuda().requires_grad_(True)
    with torch.enable_grad():
        y = mlp(x)
        loss = (y ** 2).mean()
    loss.backward()

    def rel(a, b):
        return float(np.abs(a.detach().cpu().numpy().reshape(b.shape) - b).max()) / max(float(np.abs(b).max()), 1e-30)
    assert rel(y, PRECISE[case + "/y"]) < 1e-10
    assert rel(x.grad, PRECISE[case + "/gx"]) < 1e-8
    assert rel(mlp.u_v_b_pars.grad, PRECISE[case + "/gp"]) < 1e-8
    with torch.no_grad():
        assert rel(mlp(x.detach()), PRECISE[case + "/y"]) < 1e-10


def test_precise_structure_refuses_unknown_activations():
    from jammy_flows_amd.amortizable_mlp import AmortizableMLP
    table = {"mlp_list": [dict(inputs=[3, 5], outputs=[5, 2], low_rank_approximations=[0, 0], add_final_bias=True, svd_mode="smart",
                               activations=[lambda t: t ** 3, lambda t: t])]}
    with pytest.raises(NotImplementedError):
        AmortizableMLP(3, "1", 2, precise_mlp_structure=table)
    table["mlp_list"][0]["activations"] = [torch.tanh, lambda t: t]
    assert AmortizableMLP(3, "1", 2, precise_mlp_structure=table).num_amortization_params == 3 * 5 + 5 + 5 * 2 + 2
