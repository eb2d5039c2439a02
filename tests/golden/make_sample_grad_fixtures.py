#!/usr/bin/env python3
"""Gradients THROUGH SAMPLING from the REAL reference ("differentiable Newton inverse", bisection_n_newton.py:74-93, README.md:7).

For selected golden fixtures the reference pdf is rebuilt, the fixture's injected base points z (first N rows) are pushed through
pdf._obtain_sample(predefined_target_input=z) with autograd enabled -- the reference differentiates through its Newton iterations -- and

    loss = mean_b( <w, x_b> ) + 0.1 * mean_b( log_prob_b ),     w = linspace(0.5, 1.5, D_target)

is back-propagated in float64.  Stored in tests/golden/sample_grads/<name>.npz: rows, loss, the samples, d loss / d every parameter,
d loss / d conditional_input.  (mix_e2s1i1 is left out: its 'r' layer clamps to [-1, 1] on the interval [-2, 3] -- rational_quadratic_spline.py:185-186 --
so the map is not invertible there and the sampling Jacobian is singular.)

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/make_sample_grad_fixtures.py [name-substring ...]
"""
import contextlib
import io
import os
import sys

import numpy
import torch

sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
with contextlib.redirect_stdout(io.StringIO()):
    import jammy_flows  # noqa: E402
import fixture_io  # noqa: E402

CASES = ["c1_e2_gg", "g_e3_ggg_cond", "g_e1e2e1_cond", "c4_i1s1_ro", "f_s2_cond_ff", "t_e3_gggt", "c3_e4s2e4", "m_s1_cond"]
N = 48


def make(name):
    fx = fixture_io.load(name)
    with contextlib.redirect_stdout(io.StringIO()):
        pdf = jammy_flows.pdf(fx.pdf_defs, fx.flow_defs, **fx.kwargs).double()
    pdf.load_state_dict({k: torch.from_numpy(numpy.ascontiguousarray(v)) for k, v in fx.state_dict().items()}, strict=True)
    z = torch.from_numpy(fx["z"][:N])
    cond = torch.from_numpy(fx["cond"][:N]).clone().requires_grad_(True) if fx.get("cond") is not None else None
    with contextlib.redirect_stdout(io.StringIO()):
        x, _, logp, _ = pdf._obtain_sample(conditional_input=cond, predefined_target_input=z.clone())
    w = torch.linspace(0.5, 1.5, x.shape[1], dtype=torch.float64)
    loss = (x * w).sum(dim=1).mean() + 0.1 * logp.mean()
    loss.backward()
    out = {"loss": numpy.array(loss.item()), "x": x.detach().numpy(), "logp": logp.detach().numpy(), "w": w.numpy()}
    if cond is not None:
        out["cond_grad"] = cond.grad.numpy()
    for k, p in pdf.named_parameters():
        if p.grad is not None:
            out["pg/" + k] = p.grad.detach().numpy().copy()
    os.makedirs(os.path.join(HERE, "sample_grads"), exist_ok=True)
    path = os.path.join(HERE, "sample_grads", name + ".npz")
    numpy.savez_compressed(path, **out)
    gmax = max(float(numpy.abs(v).max()) for k, v in out.items() if k.startswith("pg/"))
    print("%-22s loss=%.6f max|dL/dparam|=%.3e n_param_grads=%d bytes=%d" % (name, loss.item(), gmax, sum(k.startswith("pg/") for k in out),
                                                                            os.path.getsize(path)))


if __name__ == "__main__":
    sel = sys.argv[1:]
    for c in CASES:
        if sel and not any(s in c for s in sel):
            continue
        make(c)
