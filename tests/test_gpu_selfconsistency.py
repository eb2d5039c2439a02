"""The reference's own self-consistency pin, run as written (tests/test_general.py:393-588) on the HIP path: every entry of its 77-entry
configuration list (tests/golden/selfconsistency_list.json, dumped from the reference's test module by make_selfconsistency_list.py),
unconditional and with a 2-dimensional conditional input (x 100, as there), float64, 10 000 samples, intrinsic and embedding coordinates:

    sample -> forward must return the base points, the log-probs and the base log-probs to 1e-6 ('v': 1e-4), the samples must not be
    modified in place, a caller's log_det tensor must not be written to by all_layer_forward / all_layer_inverse, and (for flows without
    'g') obtain_flow_param_structure must account for every layer parameter.

Entries that cannot run are listed by name with the reason (NOT_RUN below): the continuous flow 'c' needs torchdiffeq (out of scope,
SURVEY 8 / DESIGN 1)."""
import copy
import json
import os
import random

import numpy as np
import pytest
import torch

import fixture_io

LIST = json.load(open(os.path.join(fixture_io.GOLDEN_DIR, "selfconsistency_list.json")))
ENTRIES = LIST["entries"]
NOT_RUN = {i: "continuous flow 'c' (torchdiffeq ODE layer: out of scope)" for i, e in enumerate(ENTRIES) if "c" in e["flow_defs"]}


def _kwargs(e):
    kw = copy.deepcopy(e["kwargs"])
    oo = kw.get("options_overwrite")
    if isinstance(oo, dict):
        kw["options_overwrite"] = {(int(k[1:]) if isinstance(k, str) and k.startswith("#") else k): v for k, v in oo.items()}
    return kw


def _ids():
    return ["%02d_%s_%s" % (i, e["pdf_defs"].replace("+", "-"), e["flow_defs"].replace("+", "-")) for i, e in enumerate(ENTRIES)]


def test_the_list_is_the_references_and_what_is_left_out_is_named():
    assert len(ENTRIES) == 77 and LIST["samplesize"] == 10000
    assert sorted(NOT_RUN) == [i for i, e in enumerate(ENTRIES) if "c" in e["flow_defs"]] and len(NOT_RUN) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("conditional", [False, True], ids=["uncond", "cond"])
@pytest.mark.parametrize("idx", range(len(ENTRIES)), ids=_ids())
def test_selfconsistency_as_the_reference_runs_it(idx, conditional):
    import jammy_flows_amd
    if idx in NOT_RUN:
        pytest.skip(NOT_RUN[idx])
    e = ENTRIES[idx]
    kw = _kwargs(e)
    assert "conditional_input_dim" not in kw
    if conditional:
        kw["conditional_input_dim"] = LIST["conditional_input_dim_added"]
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    n = LIST["samplesize"]
    pdf = jammy_flows_amd.pdf(e["pdf_defs"], e["flow_defs"], **kw).double().cuda()
    if len(e["pdf_defs"].split("+")) == 1 and e["pdf_defs"][0] == "e":          # pure Euclidean: the data-driven initialisation is exercised too
        pdf.init_params(data=torch.randn(100, int(e["pdf_defs"][1:]), dtype=torch.float64, device="cuda"))
    cinput = torch.from_numpy(np.random.normal(size=(n, 2)) * 100.0).cuda() if conditional else None
    tol = 1e-4 if "v" in e["flow_defs"] else 1e-6
    for emb in (False, True):
        with torch.no_grad():
            samples, base, evals, base_evals = pdf.sample(samplesize=n, conditional_input=cinput, force_embedding_coordinates=emb)
            before = samples.clone()
            evals2, base_evals2, base2 = pdf(samples, conditional_input=cinput, force_embedding_coordinates=emb)
            assert torch.equal(samples, before), "forward modified its input"
            test_sample = torch.rand(10, pdf.total_target_dim, dtype=torch.float64, device="cuda")
            log_det = torch.zeros(10, dtype=torch.float64, device="cuda")
            inp = None if cinput is None else cinput[:10]
            pdf.all_layer_forward(test_sample, log_det, inp)
            assert int((log_det == 0).sum()) == 10, "all_layer_forward wrote into the caller's log_det"
            pdf.all_layer_inverse(test_sample, log_det, inp)
            assert int((log_det == 0).sum()) == 10, "all_layer_inverse wrote into the caller's log_det"
        for name, a, b in (("base_samples", base, base2), ("evals", evals, evals2), ("base_evals", base_evals, base_evals2)):
            d = (a - b).abs()
            assert bool(torch.isfinite(d).all()), name
            assert float(d.max()) <= tol, "%s differ by %.3e (%d of %d beyond %.0e), embedding=%s" % (name, float(d.max()), int((d > tol).sum()), d.numel(), tol, emb)
        if "g" not in e["flow_defs"]:
            struct = pdf.obtain_flow_param_structure(conditional_input=None if cinput is None else cinput[:1])
            fps = sum(v.numel() for d in struct.values() for v in d.values())
            explicit = sum(l.total_param_num for block in pdf.layer_list for l in block)
            assert explicit == fps, ("explicit", explicit, "flow params", fps)
