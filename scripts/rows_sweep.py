#!/usr/bin/env python3
"""Step time of a BASELINE configuration against the batch size on one MI355X: eager pdf.forward (deferred status), the C-side step plan
(pdf.planned_forward, when the library has it), HIP-graph replay, and the per-kernel HIP-event times of the eager step.
What strong scaling over 8 GPUs needs is t(2^17) <= t(2^20) / (8 x 0.85).

    python3 scripts/rows_sweep.py [fixture] [f32|f64] [log2 min] [log2 max]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import numpy as np
import torch
import fixture_io
import helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip

torch.set_grad_enabled(False)
name = sys.argv[1] if len(sys.argv) > 1 else "c3_e4s2e4"
dtype = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 12
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 20
fx = fixture_io.load(name)
pdf = helpers.build_product(fx, dtype)
pdf.check_status = "deferred"


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    pdf.flush_status()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


rows = []
for lg in range(hi, lo - 1, -1):
    n = 1 << lg
    x64, c64 = inputs(fx, n, 7)
    x = torch.from_numpy(x64).to(device="cuda", dtype=dtype)
    c = torch.from_numpy(c64).to(device="cuda", dtype=dtype) if c64 is not None else None
    steps = 50 if lg >= 18 else 200
    r = {"log2_rows": lg, "eager_ms": 1e3 * timeit(lambda: pdf(x, conditional_input=c), steps)}
    if hasattr(pdf, "planned_forward"):
        try:
            pf = pdf.planned_forward(x, conditional_input=c)
            r["plan_ms"] = 1e3 * timeit(lambda: pf(x, conditional_input=c), steps)
            ref = pdf(x, conditional_input=c)[0]
            r["plan_bit_identical"] = bool(torch.equal(pf(x, conditional_input=c)[0], ref))
        except Exception as e:                               # noqa: BLE001
            r["plan_error"] = repr(e)[:160]
    g = pdf.graphed_forward(x, conditional_input=c)
    r["graph_ms"] = 1e3 * timeit(lambda: g.graph.replay(), steps)
    del g
    t = _hip.KernelTimer()
    with t:
        for _ in range(10):
            (pf if "plan_ms" in r else pdf)(x, conditional_input=c)       # a plan records its events in C, back to back with the launches
    r["kernels_ms"] = {"%s[%s]" % k: round(v["mean_ms"], 4) for k, v in sorted(t.summary().items())}
    r["kernel_sum_ms"] = round(sum(v for v in r["kernels_ms"].values()), 4)
    rows.append(r)
    print(json.dumps(r), flush=True)
t20 = [r for r in rows if r["log2_rows"] == 20]
if t20:
    for key in ("eager_ms", "plan_ms", "graph_ms"):
        if key in t20[0]:
            print(key, "efficiency vs 2^20:", {r["log2_rows"]: round(t20[0][key] / (r[key] * (1 << (20 - r["log2_rows"]))), 3) for r in rows if key in r})
