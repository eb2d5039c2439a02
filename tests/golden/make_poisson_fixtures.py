#!/usr/bin/env python3
"""Golden vectors for the Poisson head (predict_log_normalization, jammy_flows/main/default.py:51, 104-110, 467-477, 624-627, 836-877,
976-978, 1893-1896) from the REAL reference: a conditional pdf("e2", "gg") whose first MLP also emits log-lambda
(join_poisson_and_pdf_description=True), and an unconditional one whose log-lambda is a parameter.  Stored: state_dict, inputs, log-prob,
log_mean_poisson, and the gradients of  loss = -mean(log p) + mean(exp(log lambda) - 3 log lambda)  (an extended-likelihood style objective).

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/make_poisson_fixtures.py
"""
import contextlib
import io
import os
import sys

import numpy
import torch

sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
with contextlib.redirect_stdout(io.StringIO()):
    import jammy_flows  # noqa: E402

out = {}
rng = numpy.random.default_rng(9)
x = rng.normal(size=(96, 2)) * 1.2
c = rng.normal(size=(96, 3))
out["x"], out["cond"] = x, c
for name, kw, cond in (("cond_joined", dict(conditional_input_dim=3, predict_log_normalization=True, join_poisson_and_pdf_description=True), c),
                       ("uncond", dict(predict_log_normalization=True), None)):
    torch.manual_seed(4)
    numpy.random.seed(4)
    with contextlib.redirect_stdout(io.StringIO()):
        pdf = jammy_flows.pdf("e2", "gg", **kw)
    pdf.double()
    with torch.no_grad():                       # undo the MLP damping so that the parameters really vary per row
        for m in pdf.mlp_predictors:
            if m is not None:
                for mod in list(m)[:-1]:
                    if hasattr(mod, "weight"):
                        mod.weight.mul_(1000.0)
                list(m)[-1].weight.mul_(300.0)
    xt = torch.from_numpy(x)
    ct = None if cond is None else torch.from_numpy(cond)
    with contextlib.redirect_stdout(io.StringIO()):
        logp = pdf(xt, conditional_input=ct)[0]
        ll = pdf.log_mean_poisson(conditional_input=ct)
    loss = -logp.mean() + (torch.exp(ll) - 3.0 * ll).mean()
    loss.backward()
    for k, v in pdf.state_dict().items():
        out[name + "/sd/" + k] = v.detach().numpy().copy()
    out[name + "/logp"] = logp.detach().numpy()
    out[name + "/log_lambda"] = ll.detach().numpy()
    out[name + "/loss"] = numpy.array(loss.item())
    for k, p in pdf.named_parameters():
        if p.grad is not None:
            out[name + "/pg/" + k] = p.grad.numpy().copy()
    print(name, "params", pdf.count_parameters(), "loss %.6f" % loss.item(), "log_lambda", tuple(ll.shape), float(ll.mean()))
path = os.path.join(HERE, "nonlin", "poisson_head.npz")
numpy.savez_compressed(path, **out)
print(os.path.getsize(path), "bytes ->", path)
