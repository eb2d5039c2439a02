"""per-kernel HIP-event times of one C3 log-prob step at 2^20 rows (status check off): python scripts/probe/step_kernels.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import numpy as np
import torch
import fixture_io, helpers
from jammy_flows_amd import _hip
import bench

fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32)
pdf.check_status = False
x64, _ = bench.make_inputs(fx, 1 << 20, 1234) if hasattr(bench, "make_inputs") else (None, None)
if x64 is None:
    rng = np.random.default_rng(7)
    n = 1 << 20
    x64 = np.concatenate([rng.normal(size=(n, 4)) * 1.5, np.arccos(rng.uniform(-1, 1, size=(n, 1))).clip(1e-3, np.pi - 1e-3),
                          rng.uniform(0, 2 * np.pi, size=(n, 1)), rng.normal(size=(n, 4)) * 1.5], axis=1)
x = torch.from_numpy(x64).to(device="cuda", dtype=torch.float32)
with torch.no_grad():
    for _ in range(3):
        pdf(x)
    t = _hip.KernelTimer()
    with t:
        for _ in range(10):
            pdf(x)
for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
    print("%-50s %.4f ms" % (k[0] + "[" + k[1] + "]", v["mean_ms"]))
