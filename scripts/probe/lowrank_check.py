#!/usr/bin/env python3
"""C5 training step: the low-rank chain Function against the (B, P)-block sequence (values, every gradient), and timings of both."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import numpy as np
import torch
import fixture_io
import helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip

fx = fixture_io.load("c5_e8s2_ggggv")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
x64, c64 = inputs(fx, n, 7)


def run(flag, steps=0):
    pdf = helpers.build_product(fx, torch.float64)
    pdf.lowrank_chain_training = flag
    pdf.check_status = False
    x = torch.from_numpy(x64).cuda().requires_grad_(True)
    c = torch.from_numpy(c64).cuda().requires_grad_(True)
    with torch.enable_grad():
        lp = pdf(x, conditional_input=c)[0]
        loss = -lp.mean()
    loss.backward()
    out = {"logp": lp.detach(), "x": x.grad, "c": c.grad}
    for k, p in pdf.named_parameters():
        if p.grad is not None:
            out[k] = p.grad.clone()
    if steps:
        xs, cs = x.detach(), c.detach()
        t = _hip.KernelTimer()

        def step():
            for p in pdf.parameters():
                p.grad = None
            with torch.enable_grad():
                l2 = -pdf(xs, conditional_input=cs)[0].mean()
            l2.backward()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        with t:
            for _ in range(5):
                step()
        print("flag", flag, "ms/step %.3f" % (1e3 * dt))
        for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1]["total_ms"])[:14]:
            print("    %-50s %8.4f ms x %.1f" % ("%s[%s]" % k, v["mean_ms"], v["launches"] / 5))
    return out


a = run(True, steps=20 if n >= 100000 else 0)
b = run(False, steps=20 if n >= 100000 else 0)
worst = 0.0
for k in b:
    ref = b[k].double()
    err = (a[k].double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    worst = max(worst, err)
    if err > 1e-9:
        print("MISMATCH", k, err)
print("rows", n, "tensors", len(b), "worst relative difference %.3e" % worst, "finite", all(torch.isfinite(v).all().item() for v in a.values()))
