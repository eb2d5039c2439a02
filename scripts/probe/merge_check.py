#!/usr/bin/env python3
"""Merged log-prob step (csrc/merged_kernels.hip) against the separate launches: bit identity at ragged batch sizes (eager and plan),
repeat launches, and step times at 2^15 .. 2^20 rows for merge off / on.

    python3 scripts/probe/merge_check.py [fixture] [times-only]
"""
import json
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

torch.set_grad_enabled(False)
name = sys.argv[1] if len(sys.argv) > 1 else "c3_e4s2e4"
fx = fixture_io.load(name)
pdf = helpers.build_product(fx, torch.float32)
pdf.check_status = "deferred"


def dev(n, seed=7):
    x64, c64 = inputs(fx, n, seed)
    x = torch.from_numpy(x64).to(device="cuda", dtype=torch.float32)
    c = torch.from_numpy(c64).to(device="cuda", dtype=torch.float32) if c64 is not None else None
    return x, c


def timeit(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


if len(sys.argv) <= 2:
    for n in (1, 63, 64, 65, 127, 255, 256, 257, 1000, 4096 + 77, (1 << 15) + 129, 1 << 17):
        x, c = dev(n)
        pdf.merge_max_rows = 0
        _hip.lib().jf_gf_bcast_lane_rows(0)
        ref = pdf(x, conditional_input=c)
        for thr in (0, 1 << 40):
            _hip.lib().jf_gf_bcast_lane_rows(thr)
            pdf.merge_max_rows = 1 << 30
            pf = pdf.planned_forward(x, conditional_input=c)
            for rep in range(3):
                for fn in (pdf, pf):
                    got = fn(x, conditional_input=c)
                    if not all(bool(torch.equal(a, b)) for a, b in zip(got, ref)):
                        print("MISMATCH n=%d rep=%d plan=%s lanes=%d" % (n, rep, fn is pf, thr), flush=True)
            ops = pf.plan.n_ops
        print("n=%d ok, plan ops %d" % (n, ops), flush=True)
    pdf.flush_status()
    _hip.lib().jf_gf_bcast_lane_rows(-1)

for lg in (20, 19, 18, 17, 16, 15, 13):
    n = 1 << lg
    x, c = dev(n)
    steps = 50 if lg >= 18 else 200
    r = {"log2_rows": lg}
    for label, mx, thr in (("separate", 0, 0), ("separate_gl", 0, 1 << 40), ("side", 1 << 30, 0), ("side_gl", 1 << 30, 1 << 40)):
        pdf.merge_max_rows = mx
        _hip.lib().jf_gf_bcast_lane_rows(thr)
        pf = pdf.planned_forward(x, conditional_input=c)
        r[label + "_ms"] = round(1e3 * min(timeit(lambda: pf(x, conditional_input=c), steps) for _ in range(3)), 4)
        t = _hip.KernelTimer()
        with t:
            for _ in range(10):
                pf(x, conditional_input=c)
        r[label + "_kernels"] = {"%s[%s]" % k: round(v["mean_ms"], 4) for k, v in sorted(t.summary().items())}
        del pf
    print(json.dumps(r), flush=True)
