#!/usr/bin/env python3
"""average Newton row-steps per (row, g layer) of the sampling direction: python3 scripts/probe/newton_steps.py [fixture] [f32|f64] [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch
import fixture_io, helpers
from bench_configs_inputs import inputs
name = sys.argv[1] if len(sys.argv) > 1 else "c3_e4s2e4"
dtype = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
fx = fixture_io.load(name)
pdf = helpers.build_product(fx, dtype)
_, c64 = inputs(fx, n, 7)
c = None if c64 is None else torch.from_numpy(c64).to(device="cuda", dtype=dtype)
z = torch.randn((n, pdf.total_base_dim), dtype=dtype, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
with torch.no_grad():
    try:
        pdf._obtain_sample(conditional_input=c, predefined_target_input=z)
    except Exception as e:                                   # (a probe build may count wave iterations in the out-of-range word)
        print("raised:", str(e)[:80])
w = pdf.last_status_words
n_g = sum(1 for blk in pdf.layer_list for l in blk if type(l).__name__ == "gf_block")
print(name, dtype, "rows", n, "g layers", n_g, "status", w, "row-steps per (row, layer): %.2f" % (w["newton_row_steps"] / (n * n_g)))
