#!/usr/bin/env python3
"""Golden vectors for fully_amortized_pdf from the REAL reference: construction arguments, the hyper-network's state_dict
(with the reference's 1/1000 damping undone so that the emitted parameters really vary with the conditional input), inputs, forward outputs,
samples for injected base noise, the amortisation parameter block, and autograd gradients of -mean log p.

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/make_amortized_fixtures.py
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
with contextlib.redirect_stdout(io.StringIO()):
    import jammy_flows  # noqa: E402

CASES = [
    dict(name="fa_e2s1_ggm", pdf="e2+s1", flow="gg+m", kwargs=dict(conditional_input_dim=3, inner_mlp_dims_sub_pdfs="16", amortization_mlp_dims="32")),
    dict(name="fa_e2e1_hw0_rank", pdf="e2+e1", flow="gg+g", kwargs=dict(conditional_input_dim=2, inner_mlp_dims_sub_pdfs="12-10", inner_mlp_ranks=3,
                                                                       inner_mlp_highway_mode=0, amortization_mlp_dims="24",
                                                                       amortization_mlp_ranks=4, amortization_mlp_highway_mode=1)),
    dict(name="fa_i1s2_hw3", pdf="i1+s2", flow="r+f", kwargs=dict(conditional_input_dim=2, inner_mlp_dims_sub_pdfs="8-8", inner_mlp_highway_mode=3,
                                                                  amortization_mlp_dims="16", amortization_mlp_highway_mode=2)),
    dict(name="fa_e1e2_hw4", pdf="e1+e2", flow="g+gg", kwargs=dict(conditional_input_dim=2, inner_mlp_dims_sub_pdfs="4-3", inner_mlp_highway_mode=4,
                                                                   amortization_mlp_dims="8-6", amortization_mlp_highway_mode=3)),
]


def make(case):
    random.seed(1); numpy.random.seed(1); torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        pdf = jammy_flows.fully_amortized_pdf(case["pdf"], case["flow"], **case["kwargs"])
    with torch.no_grad():                          # undo the damping on everything but the final bias (= the desired initial parameters)
        mlp = pdf.amortization_mlp
        if hasattr(mlp, "u_v_b_pars"):
            nb = pdf.pdf_to_amortize.total_number_amortizable_params
            mlp.u_v_b_pars.data[0, :-nb] *= 300.0
        else:
            lin = [m for m in mlp if hasattr(m, "weight")]
            for i, m in enumerate(lin):
                m.weight.data *= 300.0
                if i < len(lin) - 1:
                    m.bias.data *= 300.0
    rng = numpy.random.default_rng(5)
    B = 96
    cdim = case["kwargs"]["conditional_input_dim"]
    cond = torch.from_numpy(rng.normal(size=(B, cdim)))
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        x = pdf.sample(conditional_input=cond, seed=3)[0]
    x = x.clone()
    z = torch.from_numpy(rng.normal(size=(B, pdf.pdf_to_amortize.total_base_dim)))
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        logp, logp_base, base = pdf(x, conditional_input=cond)
        amort = pdf.amortization_mlp(cond)
        sx, _, slogp, _ = pdf.pdf_to_amortize._obtain_sample(predefined_target_input=z.clone(), amortization_parameters=amort)
    xg = x.clone().requires_grad_(True)
    cg = cond.clone().requires_grad_(True)
    with contextlib.redirect_stdout(io.StringIO()):
        loss = -pdf(xg, conditional_input=cg)[0].mean()
    loss.backward()
    # gradients THROUGH SAMPLING into the hyper-network (fully_amortized.py:173-215 with allow_gradients): injected base points z, the
    # reference differentiates through its Newton iterations; loss = mean <w, x> + 0.1 mean log p
    for p_ in pdf.parameters():
        p_.grad = None
    cs = cond.clone().requires_grad_(True)
    with contextlib.redirect_stdout(io.StringIO()):
        amort_g = pdf.amortization_mlp(cs)
        gx, _, glogp, _ = pdf.pdf_to_amortize._obtain_sample(predefined_target_input=z.clone(), amortization_parameters=amort_g)
    sw = torch.linspace(0.5, 1.5, gx.shape[1], dtype=torch.float64)
    sloss = (gx * sw).sum(dim=1).mean() + 0.1 * glogp.mean()
    sloss.backward()
    sample_grads = {"sg/" + k: p_.grad.detach().numpy().copy() for k, p_ in pdf.named_parameters()}
    sample_grads["sg_cond"] = cs.grad.numpy().copy()
    sample_grads["sg_loss"] = numpy.array(sloss.item())
    for p_ in pdf.parameters():
        p_.grad = None
    xg = x.clone().requires_grad_(True)
    cg = cond.clone().requires_grad_(True)
    with contextlib.redirect_stdout(io.StringIO()):
        loss = -pdf(xg, conditional_input=cg)[0].mean()
    loss.backward()
    out = {"meta": numpy.array(json.dumps(dict(name=case["name"], pdf_defs=case["pdf"], flow_defs=case["flow"], kwargs=case["kwargs"],
                                               total_number_amortizable_params=int(pdf.pdf_to_amortize.total_number_amortizable_params),
                                               count_parameters=int(pdf.count_parameters())))),
           "x": x.numpy(), "cond": cond.numpy(), "z": z.numpy(), "logp": logp.numpy(), "logp_base": logp_base.numpy(), "base": base.numpy(),
           "sample_x": sx.numpy(), "sample_logp": slogp.numpy(), "loss": numpy.array(loss.item()),
           "x_grad": xg.grad.numpy(), "cond_grad": cg.grad.numpy()}
    for k, v in pdf.state_dict().items():
        out["sd/" + k] = v.detach().numpy()
    for k, p in pdf.named_parameters():
        out["pg/" + k] = p.grad.detach().numpy()
    out.update(sample_grads)
    path = os.path.join(HERE, "amortized", case["name"] + ".npz")
    numpy.savez_compressed(path, **out)
    print("%-18s T=%d hyper-net params=%d logp[min,max]=(%.3f, %.3f) loss=%.5f bytes=%d" % (
        case["name"], pdf.pdf_to_amortize.total_number_amortizable_params, pdf.count_parameters(), logp.min().item(), logp.max().item(), loss.item(),
        os.path.getsize(path)))


if __name__ == "__main__":
    for c in CASES:
        make(c)
