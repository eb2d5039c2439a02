#!/usr/bin/env python3
"""Per-kernel HIP-event times of the C3 log-prob step at 2^20 rows on the bench.py inputs (for A/B runs of kernel variants selected by
environment variables): prints `tag  step_ms  {kernel: mean_ms}`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import bench
import fixture_io
import helpers
from jammy_flows_amd import _hip

torch.set_grad_enabled(False)
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
W = bench.WORKLOADS[wl]
B = int(sys.argv[2]) if len(sys.argv) > 2 else W["rows"]
dt = torch.float32 if W["dtype"] == "f32" else torch.float64
fx = fixture_io.load(W["fixture"])
pdf = helpers.build_product(fx, dt, "cuda")
pdf.check_status = False
x64, c64 = bench.make_inputs(wl, B, W["seed"])
x = torch.from_numpy(x64).to("cuda", dt)
c = None if c64 is None else torch.from_numpy(c64).to("cuda", dt)
for _ in range(5):
    pdf(x, conditional_input=c)
torch.cuda.synchronize()
t = _hip.KernelTimer()
n = 30
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with t:
    e0.record()
    for _ in range(n):
        logp = pdf(x, conditional_input=c)[0]
    e1.record()
torch.cuda.synchronize()
o = helpers.build_oracle(fx).forward(x64[:2048], None if c64 is None else c64[:2048])[0]
err = float(np.max(np.abs(logp[:2048].double().cpu().numpy() - o)))
print(os.environ.get("TAG", ""), "step %.4f ms" % (e0.elapsed_time(e1) / n), "err %.2e" % err,
      {k[0].replace("jf_", "") + ("[" + k[1] + "]" if k[1] else ""): round(v["mean_ms"], 4) for k, v in sorted(t.summary().items())}, flush=True)
