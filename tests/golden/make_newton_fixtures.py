#!/usr/bin/env python3
"""Newton-stage records of the REAL reference for the sampling direction of every golden fixture that has classic 'g' layers
("for sampling: injected z -> x, log_det, Newton iteration counts").  Runs only in the build container:

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/make_newton_fixtures.py

The reference's solver (layers/bisection_n_newton.py:11-135) is wrapped, not modified: every call of the joint value/derivative function
during the Newton stage is one iteration over the rows still above tolerance, so the sizes of those calls are the per-iteration active-row
counts; after the solver returns, the residual |f(x) - z| of its result is evaluated with the same function the solver used.  Output:
tests/golden/newton_records.json = {fixture: {"solves": [{"active": [...], "row_steps": sum(active), "max_residual": r,
"n_above_1e-7": n, "rows": B, "dim": D}, ... one per g layer in sampling order], "row_steps_total": ...}}.
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
sys.path.insert(0, HERE)
with contextlib.redirect_stdout(io.StringIO()):
    import jammy_flows
    from jammy_flows.layers import bisection_n_newton as bn
import fixture_io  # noqa: E402

records = {}
orig = bn.inverse_bisection_n_newton_joint_func_and_grad
current = []


def wrapped(func, joint_func, target_arg, *args, **kw):
    active = []

    def counted(x, *a):
        active.append(int(x.shape[0]))
        return joint_func(x, *a)
    res = orig(func, counted, target_arg, *args, **kw)
    with torch.no_grad():
        resid = (func(res, *args) - target_arg).abs().max(dim=1)[0]
    current.append(dict(active=active, row_steps=int(sum(active)), max_residual=float(resid.max()), n_above_1e_7=int((resid > 1e-7).sum()),
                        rows=int(target_arg.shape[0]), dim=int(target_arg.shape[1])))
    return res


bn.inverse_bisection_n_newton_joint_func_and_grad = wrapped
for path in fixture_io.list_fixtures():
    fx = fixture_io.Fixture(path)
    opts = str(fx.kwargs.get("options_overwrite"))
    if "g" not in fx.flow_defs or "rq_splines" in opts:
        continue
    with contextlib.redirect_stdout(io.StringIO()):
        pdf = jammy_flows.pdf(fx.pdf_defs, fx.flow_defs, **fx.kwargs).double()
    pdf.load_state_dict({k: torch.from_numpy(numpy.ascontiguousarray(v)) for k, v in fx.state_dict().items()})
    cond = torch.from_numpy(fx["cond"]) if fx.get("cond") is not None else None
    current.clear()
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        sx, _, _, _ = pdf._obtain_sample(conditional_input=cond, predefined_target_input=torch.from_numpy(fx["z"]).clone(),
                                         force_embedding_coordinates=fx.meta["embedding"])
    assert numpy.abs(sx.numpy() - fx["sample_x"]).max() < 1e-12, fx.name      # the very run the fixture recorded
    records[fx.name] = dict(solves=[dict(c) for c in current], row_steps_total=int(sum(c["row_steps"] for c in current)))
    print("%-28s solves %2d  row-steps %7d  iterations %s  worst residual %.2e" % (
        fx.name, len(current), records[fx.name]["row_steps_total"], [len(c["active"]) for c in current], max(c["max_residual"] for c in current)))
json.dump(records, open(os.path.join(HERE, "newton_records.json"), "w"), indent=1, sort_keys=True)
