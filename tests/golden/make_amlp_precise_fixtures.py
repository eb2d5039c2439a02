#!/usr/bin/env python3
"""Golden vectors for AmortizableMLP(precise_mlp_structure=...) (jammy_flows/amortizable_mlp.py:20, 56-62: the caller hands in the sub-MLP
table itself -- per-matrix ranks, widths that no `hidden_dims` string produces) from the REAL reference: outputs, d loss / d input and
d loss / d u_v_b_pars for loss = mean(out^2), highway modes 0, 2 and 4.  The structures are stored in the fixture as JSON (the activations,
which the reference keeps as callables inside the table, follow its own rule: the nonlinearity everywhere but after a sub-MLP's last matrix).

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/make_amlp_precise_fixtures.py
"""
import contextlib
import io
import json
import os
import sys

import numpy
import torch

sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
with contextlib.redirect_stdout(io.StringIO()):
    from jammy_flows.amortizable_mlp import AmortizableMLP  # noqa: E402
    from jammy_flows.extra_functions import NONLINEARITIES  # noqa: E402


def sub(inputs, outputs, ranks, bias, svd_mode="smart"):
    return dict(inputs=inputs, outputs=outputs, low_rank_approximations=ranks, add_final_bias=bias, svd_mode=svd_mode)


def runnable(d, nonlinearity):
    """the table as the reference wants it: bookkeeping lists + activation callables"""
    d = dict(d)
    d.update(num_u_s=[], num_v_s=[], num_b_s=[], full_weight_matrix_flags=[], sigmas=[])
    d["activations"] = [NONLINEARITIES[nonlinearity]] * (len(d["inputs"]) - 1) + [lambda x: x]
    return d


IN, OUT = 6, 4
CASES = {
    "hw0": (0, {"mlp_list": [sub([IN, 11, 7], [11, 7, OUT], [2, 0, 3], True)]}),
    "hw2": (2, {"mlp_list": [sub([IN, 10], [10, OUT], [3, 0], False), sub([IN, 5], [5, OUT], [0, 2], False, "naive")],
                "linear_highway": sub([IN], [OUT], [2], True)}),
    "hw4": (4, {"mlp_list": [sub([IN, 9], [9, OUT], [0, 0], False), sub([IN + OUT, 8], [8, OUT], [4, 2], False)],
                "linear_highway": sub([IN], [OUT], [0], True)}),
}
rng = numpy.random.default_rng(5)
x = rng.normal(size=(48, IN))
out = {"x": x, "structures": numpy.array(json.dumps({k: {"highway_mode": hw, "structure": st} for k, (hw, st) in CASES.items()}))}
for k, (hw, st) in CASES.items():
    table = {"mlp_list": [runnable(d, "tanh") for d in st["mlp_list"]]}
    if "linear_highway" in st:
        table["linear_highway"] = runnable(st["linear_highway"], "tanh")
    torch.manual_seed(2)
    with contextlib.redirect_stdout(io.StringIO()):
        mlp = AmortizableMLP(IN, "3", OUT, highway_mode=hw, nonlinearity="tanh", use_permanent_parameters=True, precise_mlp_structure=table).double()
    with torch.no_grad():
        mlp.u_v_b_pars.data *= 300.0
        mlp.u_v_b_pars.data = mlp.u_v_b_pars.data.clamp(-1.5, 1.5)
    xt = torch.from_numpy(x).clone().requires_grad_(True)
    y = mlp(xt)
    loss = (y ** 2).mean()
    loss.backward()
    out[k + "/pars"] = mlp.u_v_b_pars.detach().numpy().copy()
    out[k + "/y"] = y.detach().numpy()
    out[k + "/gx"] = xt.grad.numpy().copy()
    out[k + "/gp"] = mlp.u_v_b_pars.grad.numpy().copy()
    print(k, "params", mlp.u_v_b_pars.numel(), "|y| max %.3f" % float(y.abs().max()), "loss %.4f" % loss.item())
path = os.path.join(HERE, "nonlin", "amlp_precise_structure.npz")
numpy.savez_compressed(path, **out)
print(os.path.getsize(path), "bytes ->", path)
