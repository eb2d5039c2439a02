#!/usr/bin/env python3
"""C4 pdf("i1+s1","r+o") float32, 2^20 rows: the default launch sequence (r chain, jf_mlp2, o chain) against the one-launch
MLP + 'o' chain (pdf.force_fused_manifold_blocks).  Prints step time and the per-kernel table of both."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch
import fixture_io
import helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip

torch.set_grad_enabled(False)
fx = fixture_io.load("c4_i1s1_ro")
B = 1 << 20
x_np, _ = inputs(fx, B, 7)
x, c = torch.from_numpy(x_np).to(device="cuda", dtype=torch.float32), None
for forced in (False, True):
    pdf = helpers.build_product(fx, torch.float32)
    pdf.force_fused_manifold_blocks = forced
    pdf.check_status = False
    for _ in range(5):
        lp = pdf(x, conditional_input=c)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        lp = pdf(x, conditional_input=c)[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    timer = _hip.KernelTimer()
    with timer:
        for _ in range(10):
            pdf(x, conditional_input=c)
    print("forced" if forced else "default", "%.4f ms" % (dt * 1e3), {k[0] + "[" + k[1] + "]": round(v["mean_ms"], 4) for k, v in timer.summary().items()},
          float(lp.double().mean()))
