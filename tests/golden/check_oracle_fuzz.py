#!/usr/bin/env python3
"""Build-container check (imports the REAL reference): the numpy oracle of the 'g' layer against the reference's gf_block on the SAME random
option products tests/test_gpu_fuzz.py draws (same seeds, same generator), both directions, per-sample and broadcast parameters.  Prints the
worst deviation; nothing is stored -- the GPU fuzz test then compares the kernels with the oracle on these cases.

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/check_oracle_fuzz.py
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
with contextlib.redirect_stdout(io.StringIO()):
    from jammy_flows.layers.euclidean import gaussianization_flow as ref_gf
from oracle import gf as ogf
import test_gpu_fuzz as fz

worst = 0.0
for seed in range(fz.N_CASES):
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.integers(1, 9))
    o = fz.random_options(rng, D)
    model_offset = int(rng.integers(0, 2))
    spec = ogf.GfSpec(D, o, model_offset)
    kw = {k: v for k, v in o.items() if k not in ("replace_first_sigmoid_with_icdf", "skip_model_offset")}
    with contextlib.redirect_stdout(io.StringIO()):
        layer = ref_gf.gf_block(D, use_permanent_parameters=False, model_offset=model_offset, **kw).double()
    assert layer.total_param_num == spec.total_param_num, (o, layer.total_param_num, spec.total_param_num)
    B = 96
    for pb in (B, 1):
        params = rng.normal(size=(pb, spec.total_param_num)) * 0.8
        x = rng.normal(size=(B, D)) * 2.0
        x[:4] *= 8.0
        y, ld, _ = ogf.inverse(spec, x, np.zeros(B), params)
        rparams = torch.from_numpy(np.repeat(params, B // pb, axis=0))      # the reference's extra_inputs path wants one row per sample
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            ry, rld = layer.inv_flow_mapping([torch.from_numpy(x), torch.zeros(B, dtype=torch.float64)], extra_inputs=rparams)
        ok = np.isfinite(y).all(axis=1) & np.isfinite(ld) & torch.isfinite(ry).all(dim=1).numpy() & torch.isfinite(rld).numpy()
        e = max(float((np.abs(ry.numpy() - y)[ok] / (1 + np.abs(y[ok]))).max()), float((np.abs(rld.numpy() - ld)[ok] / (1 + np.abs(ld[ok]))).max()))
        z = rng.normal(size=(B, D))
        xs, lds, _ = ogf.forward(spec, z, np.zeros(B), params)
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            rxs, rlds = layer.flow_mapping([torch.from_numpy(z), torch.zeros(B, dtype=torch.float64)], extra_inputs=rparams)
        e2 = max(float((np.abs(rxs.numpy() - xs) / (1 + np.abs(xs))).max()), float((np.abs(rlds.numpy() - lds) / (1 + np.abs(lds))).max()))
        worst = max(worst, e, e2)
        print("seed %2d D %d pb %2d rows ok %d  inv %.2e  fwd %.2e  %s/%s%s%s" % (seed, D, pb, int(ok.sum()), e, e2, o["inverse_function_type"],
                                                                                    o["rotation_mode"], " skew" if o["add_skewness"] else "",
                                                                                    " center" if o["center_mean"] else ""))
print("worst relative deviation oracle vs reference (g layer option products): %.3e" % worst)

# ---- pdf-level cases of test_random_pdf_structures_and_options_vs_oracle: the reference pdf with the product's state_dict vs the oracle
with contextlib.redirect_stdout(io.StringIO()):
    import jammy_flows
from oracle import OraclePdf
worst = 0.0
for seed in range(40):
    rng, pdf_defs, flow_defs, kwargs, pdf = fz.build_fuzz_pdf(seed)
    sd = {k: v.detach().clone() for k, v in pdf.state_dict().items()}
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            ref = jammy_flows.pdf(pdf_defs, flow_defs, **kwargs).double()
        ref.load_state_dict(sd, strict=True)
    except Exception as e:                                  # noqa: BLE001
        print("seed %2d %s / %s: reference cannot construct / load: %s" % (seed, pdf_defs, flow_defs, repr(e)[:160]))
        continue
    oracle = OraclePdf(pdf_defs, flow_defs, state_dict={k: v.numpy() for k, v in sd.items()}, **kwargs)
    B = 64
    x = fz.domain_rows(pdf_defs, B, rng)
    cond = rng.normal(size=(B, 2)) if "conditional_input_dim" in kwargs else None
    tc = None if cond is None else torch.from_numpy(cond)
    o_logp, _, o_base = oracle.forward(x, cond)
    z = rng.normal(size=(B, pdf.total_base_dim))
    o_x, o_slogp = oracle.sample_from_base(z, cond)[:2]
    try:
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            r_logp, _, r_base = ref(torch.from_numpy(x), conditional_input=tc)
            r_x, _, r_slogp, _ = ref._obtain_sample(conditional_input=tc, predefined_target_input=torch.from_numpy(z).clone())
    except Exception as e:                                  # noqa: BLE001
        print("seed %2d %s / %s: reference raised %s" % (seed, pdf_defs, flow_defs, repr(e)[:160]))
        continue
    e1 = float((np.abs(r_logp.numpy() - o_logp) / (1 + np.abs(o_logp))).max())
    e2 = float((np.abs(r_x.numpy() - o_x) / (1 + np.abs(o_x))).max())
    e3 = float((np.abs(r_slogp.numpy() - o_slogp) / (1 + np.abs(o_slogp))).max())
    if "v" not in flow_defs:
        worst = max(worst, e1, e2, e3)
    print("seed %2d %-22s %-10s %s logp %.2e  sample %.2e  sample logp %.2e" % (seed, pdf_defs, flow_defs, "cond" if cond is not None else "    ", e1, e2, e3))
print("worst relative deviation oracle vs reference (pdf-level cases without 'v'): %.3e" % worst)
