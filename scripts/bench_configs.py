#!/usr/bin/env python3
"""Throughput of the five BASELINE.json configurations (SURVEY 8d inputs) on one MI355X: log-prob evals/s and samples/s.
Not the contract benchmark (bench.py is, on configs[2]); this is the table in DESIGN.md section 5."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import fixture_io
import helpers
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_configs_inputs import inputs

torch.set_grad_enabled(False)

CONFIGS = [  # (fixture, dtype, rows, label)
    ("c1_e2_gg", torch.float64, 4096, 'C1 pdf("e2","gg") f64'),
    ("c2_e4_gggg", torch.float32, 1 << 20, 'C2 pdf("e4","gggg") f32'),
    ("c3_e4s2e4", torch.float32, 1 << 20, 'C3 pdf("e4+s2+e4","gggg+f+gggg") f32'),
    ("c3_e4s2e4", torch.float64, 1 << 20, 'C3 f64'),
    ("c3b_e4s2e4_fsplines", torch.float32, 1 << 20, 'C3 with f splines (vertical rr + circular oo) f32'),
    ("c4_i1s1_ro", torch.float32, 1 << 20, 'C4 pdf("i1+s1","r+o") f32'),
    ("c4_i1s1_ro", torch.float64, 1 << 20, 'C4 f64'),
    ("c5_e8s2_ggggv", torch.float64, 1 << 19, 'C5 conditional pdf("e8+s2","gggg+v"), AmortizableMLP rank 8, f64, 2^19 rows (one GPU share)'),
]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


print("| configuration | rows | log-prob ms | evals/s | HIP-graph replay ms | sampling ms | samples/s |")
print("|---|---|---|---|---|---|---|")
for name, dtype, n, label in CONFIGS:
    fx = fixture_io.load(name)
    pdf = helpers.build_product(fx, dtype)
    pdf.check_status = False
    x, cond = inputs(fx, n, 7)
    x = torch.from_numpy(x).to(device="cuda", dtype=dtype)
    cond = torch.from_numpy(cond).to(device="cuda", dtype=dtype) if cond is not None else None
    t_lp = timeit(lambda: pdf(x, conditional_input=cond))
    g = pdf.graphed_forward(x, conditional_input=cond)
    t_g = timeit(lambda: g(x, conditional_input=cond, check=False), n=20)
    z = torch.randn(n, pdf.total_base_dim, device="cuda", dtype=dtype)
    try:
        t_s = timeit(lambda: pdf._obtain_sample(conditional_input=cond, predefined_target_input=z), n=2)
        s_txt = "%.2f | %.3g" % (1e3 * t_s, n / t_s)
    except Exception as e:            # noqa: BLE001 -- report, do not hide
        s_txt = "%s | -" % type(e).__name__
    print("| %s | %d | %.3f | %.3g | %.3f | %s |" % (label, n, 1e3 * t_lp, n / t_lp, 1e3 * t_g, s_txt))
