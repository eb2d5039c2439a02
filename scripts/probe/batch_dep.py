#!/usr/bin/env python3
"""does a row's result depend on the batch it sits in?  C3 float32: rows of a 2^20 batch against the same rows evaluated alone, per block"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import numpy as np, torch, fixture_io, helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip
torch.set_grad_enabled(False)
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32)
pdf.check_status = False
n = 1 << 20
x64, _ = inputs(fx, n, 1234)
x = torch.from_numpy(x64).to("cuda", torch.float32)
full = pdf(x)
for m in (1 << 16, 1 << 17, 1 << 15, 5000):
    sel = torch.arange(0, n, n // m, device="cuda")[:m]
    for rg in (0, 1, 2):
        _hip.lib().jf_cond_gf_split_row_groups(rg)
        alone = pdf(x[sel].contiguous())
        d = [(a != b[sel]) & ~(a.isnan() & b[sel].isnan()) for a, b in zip(alone, full)]
        print("rows", m, "rg", rg, "logp diff rows", int(d[0].sum()), "base cols differing", [int(d[2][:, j].sum()) for j in range(d[2].shape[1])], flush=True)
_hip.lib().jf_cond_gf_split_row_groups(0)
