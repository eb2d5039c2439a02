"""Newton row-steps of the sampling direction (status word 3) per row and g layer: python3 scripts/probe/sample_steps.py [c3|c5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import fixture_io, helpers
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
W = bench.WORKLOADS[wl]
dtype = torch.float32 if W["dtype"] == "f32" else torch.float64
fx = fixture_io.load(W["fixture"])
pdf = helpers.build_product(fx, dtype, torch.device("cuda"))
B = 1 << 16
x64, c64 = bench.make_inputs(wl, B, W["seed"])
c = None if c64 is None else torch.from_numpy(c64).to(device="cuda", dtype=dtype)
with torch.no_grad():
    out = pdf.sample(conditional_input=c, samplesize=B if c is None else 1)
torch.cuda.synchronize()
print(wl, "rows", B, "status", pdf.last_status_words, "newton row-steps per row: %.2f" % (pdf.last_status_words["newton_row_steps"] / B))
