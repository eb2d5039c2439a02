#!/usr/bin/env python3
"""Sampling step of C2 / C3 (float32, 2^20 rows) and C1 (float64, 2^18) with and without the broadcast sampler's start table
(_hip.FWD_TABLE_MIN_ROWS): step time, kernel table, Newton row-steps."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import fixture_io
import helpers
from jammy_flows_amd import _hip

torch.set_grad_enabled(False)
for name, dtype, B in (("c2_e4_gggg", torch.float32, 1 << 20), ("c3_e4s2e4", torch.float32, 1 << 20), ("c1_e2_gg", torch.float64, 1 << 18)):
    fx = fixture_io.load(name)
    pdf = helpers.build_product(fx, dtype)
    pdf.check_status = False
    z = torch.randn(B, pdf.total_base_dim, device="cuda", dtype=dtype)
    for min_rows in (1 << 62, 8192):
        _hip.FWD_TABLE_MIN_ROWS = min_rows
        for _ in range(3):
            pdf._obtain_sample(predefined_target_input=z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pdf._obtain_sample(predefined_target_input=z)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        timer = _hip.KernelTimer()
        with timer:
            for _ in range(5):
                pdf._obtain_sample(predefined_target_input=z)
        print(name, "table" if min_rows == 8192 else "plain", "%.3f ms" % (dt * 1e3),
              {k[0] + "[" + k[1] + "]": round(v["mean_ms"], 4) for k, v in timer.summary().items() if v["mean_ms"] > 0.02})
