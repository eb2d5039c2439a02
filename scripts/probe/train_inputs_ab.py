#!/usr/bin/env python3
"""training step (forward + backward) on SURVEY 8d inputs against samples of the model itself: how much of the backward kernels' time is the
log-space fallback that tail rows send whole waves to -- python3 scripts/probe/train_inputs_ab.py c3|c5"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch
import fixture_io, helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
name, dtype, n = {"c3": ("c3_e4s2e4", torch.float32, 1 << 18), "c5": ("c5_e8s2_ggggv", torch.float64, 1 << 17)}[wl]
fx = fixture_io.load(name)
pdf = helpers.build_product(fx, dtype)
pdf.check_status = False
x64, c64 = inputs(fx, n, 7)
xs = torch.from_numpy(x64).to(device="cuda", dtype=dtype)
c = None if c64 is None else torch.from_numpy(c64).to(device="cuda", dtype=dtype)
with torch.no_grad():
    xm = pdf._obtain_sample(conditional_input=c, predefined_target_input=torch.randn((n, pdf.total_base_dim), dtype=dtype, device="cuda"))[0]
for label, x in (("survey inputs", xs), ("model samples", xm)):
    def step():
        for p in pdf.parameters():
            p.grad = None
        with torch.enable_grad():
            loss = -pdf(x, conditional_input=c)[0].mean()
        loss.backward()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20 * 1e3
    t = _hip.KernelTimer()
    with t:
        for _ in range(5): step()
    top = sorted(t.summary().items(), key=lambda kv: -kv[1]["total_ms"])[:5]
    print("%s %-14s %.3f ms/step   " % (wl, label, dt) + "  ".join("%s %.3f" % (k[0].replace("jf_", ""), v["total_ms"] / 5) for k, v in top))
