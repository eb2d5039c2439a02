#!/usr/bin/env python3
"""Golden vectors for the AmortizableMLP nonlinearities (extra_functions.py:81-89) from the REAL reference: for every name of NONLINEARITIES
an AmortizableMLP(5, "12-9", 7, low_rank_approximations=3, nonlinearity=name, highway_mode=hw) with permanent parameters (init undamped) is
evaluated on a fixed input, and loss = mean(out^2) is back-propagated: outputs, d loss / d input, d loss / d u_v_b_pars.

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/make_amlp_nonlin_fixtures.py
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
    from jammy_flows.amortizable_mlp import AmortizableMLP  # noqa: E402
    from jammy_flows.extra_functions import NONLINEARITIES  # noqa: E402

out = {"names": numpy.array(sorted(NONLINEARITIES))}
rng = numpy.random.default_rng(3)
x = rng.normal(size=(64, 5))
out["x"] = x
for hw in (0, 1):
    for name in sorted(NONLINEARITIES):
        torch.manual_seed(1)
        with contextlib.redirect_stdout(io.StringIO()):
            mlp = AmortizableMLP(5, "12-9", 7, low_rank_approximations=3, nonlinearity=name, highway_mode=hw, use_permanent_parameters=True).double()
        with torch.no_grad():
            mlp.u_v_b_pars.data *= 300.0            # the reference damps its init by 1000: undo, so that the activations see O(1) arguments
            mlp.u_v_b_pars.data = mlp.u_v_b_pars.data.clamp(-1.5, 1.5)
        xt = torch.from_numpy(x).clone().requires_grad_(True)
        y = mlp(xt)
        loss = (y ** 2).mean()
        loss.backward()
        k = "%s_hw%d" % (name, hw)
        out[k + "/pars"] = mlp.u_v_b_pars.detach().numpy().copy()
        out[k + "/y"] = y.detach().numpy()
        out[k + "/gx"] = xt.grad.numpy().copy()
        out[k + "/gp"] = mlp.u_v_b_pars.grad.numpy().copy()
        print(k, "params", mlp.u_v_b_pars.numel(), "|y| max %.3f" % float(y.abs().max()), "loss %.4f" % loss.item())
os.makedirs(os.path.join(HERE, "nonlin"), exist_ok=True)
path = os.path.join(HERE, "nonlin", "amlp_nonlinearities.npz")
numpy.savez_compressed(path, **out)
print(os.path.getsize(path), "bytes ->", path)
