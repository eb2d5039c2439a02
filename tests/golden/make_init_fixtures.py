#!/usr/bin/env python3
"""Golden vectors of the data-driven initialisation pdf.init_params(data=...) (main/default.py:1817-1952, extra_functions.py:179-409) from the
REAL reference.  Runs only in the build container:

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/make_init_fixtures.py

Two kinds of case:
  * deterministic ones (no Householder fit, no 't' fit: those start scipy.optimize from numpy.random draws): the initial parameter vector of
    every block is recorded and compared element by element;
  * the docs' recommended "gggt" (suggested_settings.rst:12-42), whose fits have no unique optimum: recorded are the data log-probabilities
    right after initialisation (mean and per-row) -- the property the procedure exists for -- to be compared statistically.
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
    import jammy_flows

CASES = [
    dict(name="init_e3_gg_angles", pdf="e3", flow="gg", kwargs=dict(options_overwrite={"g": {"rotation_mode": "angles"}})),
    dict(name="init_e2_ggg_none_skew_center", pdf="e2", flow="ggg",
         kwargs=dict(options_overwrite={"g": {"rotation_mode": "none", "add_skewness": 1, "center_mean": 1, "num_kde": 8}})),
    dict(name="init_e2e2_cond", pdf="e2+e2", flow="gg+gg", kwargs=dict(conditional_input_dim=2, options_overwrite={"g": {"rotation_mode": "triangular_combination"}})),
    dict(name="init_e3_gggt", pdf="e3", flow="gggt", kwargs={}, stochastic=True),
]


def make_data(n, d, rng):
    z = rng.normal(size=(n, d))
    a = rng.normal(size=(d, d))
    x = z @ a.T
    x[:, 0] = numpy.sinh(0.7 * x[:, 0])               # heavy tails on one axis
    x[:, -1] = x[:, -1] + 0.5 * x[:, 0] ** 2 / (1.0 + numpy.abs(x[:, 0]))
    return x + rng.normal(size=(1, d)) * 2.0


for case in CASES:
    rng = numpy.random.default_rng(77)
    numpy.random.seed(5); torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        pdf = jammy_flows.pdf(case["pdf"], case["flow"], **case["kwargs"]).double()
    data = torch.from_numpy(make_data(3000, pdf.total_target_dim, rng))
    cdim = case["kwargs"].get("conditional_input_dim")
    cond = torch.from_numpy(rng.normal(size=(3000, cdim))) if cdim else None
    numpy.random.seed(6); torch.manual_seed(6)
    with contextlib.redirect_stdout(io.StringIO()):
        pdf.init_params(data=data)
    with torch.no_grad():
        logp, _, base = pdf(data, conditional_input=cond)
    out = {"data": data.numpy(), "logp": logp.numpy(), "base": base.numpy(), "stochastic": numpy.array(bool(case.get("stochastic", False)))}
    if cond is not None:
        out["cond"] = cond.numpy()
    import json
    from fixture_io import encode_opts
    out["meta"] = numpy.array(json.dumps(dict(name=case["name"], pdf_defs=case["pdf"], flow_defs=case["flow"],
                                              kwargs={k: (encode_opts(v) if k == "options_overwrite" else v) for k, v in case["kwargs"].items()})))
    if not case.get("stochastic"):
        for k, v in pdf.state_dict().items():
            out["sd/" + k] = v.detach().numpy()
    os.makedirs(os.path.join(HERE, "init"), exist_ok=True)
    path = os.path.join(HERE, "init", case["name"] + ".npz")
    numpy.savez_compressed(path, **out)
    print("%-30s mean logp %.4f  base mean %s  base std %s  bytes %d" % (case["name"], logp.mean().item(), base.mean(0).numpy().round(3),
                                                                           base.std(0).numpy().round(3), os.path.getsize(path)))
