#!/usr/bin/env python3
"""C4 pdf("i1+s1", "r+o") float32: the o block as two launches (narrow MLP + o chain) against the one-launch fused block (cond_mchain_kernel<OFam>)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch, fixture_io, helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip
torch.set_grad_enabled(False)
fx = fixture_io.load("c4_i1s1_ro")
x64, _ = inputs(fx, 1 << 20, 7)
x = torch.from_numpy(x64).to("cuda", torch.float32)
out = {}
ref = None
for label, force in (("two_launches", False), ("fused", True)):
    pdf = helpers.build_product(fx, torch.float32)
    pdf.check_status = "deferred"
    pdf.force_fused_manifold_blocks = force
    pf = pdf.planned_forward(x)
    for _ in range(10): pf(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): pf(x)
    torch.cuda.synchronize(); out[label + "_ms"] = round((time.perf_counter() - t0) * 10, 4)
    t = _hip.KernelTimer()
    with t:
        for _ in range(10): pf(x)
    out[label + "_kernels"] = {k[0]: round(v["mean_ms"], 4) for k, v in t.summary().items()}
    lp = pf(x)[0]
    if ref is None: ref = lp
    else: out["max_abs_diff_logp"] = float((lp - ref).abs().max())
print(json.dumps(out))
