#!/usr/bin/env python3
"""Determinism stress of the log-prob path at full batch size: N launches of the same 2^20-row evaluation, every result compared bit for bit
with the first (and the first with the small-batch values of every replica).  python scripts/probe/stress.py [fixture] [f32|f64] [launches]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import numpy as np
import torch
from test_gpu_parity import ALL_FIXTURES, build_product, to_dev

name = sys.argv[1] if len(sys.argv) > 1 else "c3_e4s2e4"
dtype = torch.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else torch.float32
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 200
torch.set_grad_enabled(False)
fx = [f for f in ALL_FIXTURES if f.name == name][0]
n = fx["x"].shape[0]
reps = (1 << 20) // n + 2
big = reps * n - 41
pdf = build_product(fx, dtype)
pdf.check_status = False
x = to_dev(np.tile(fx["x"], (reps, 1))[:big], dtype)
cond = to_dev(np.tile(fx["cond"], (reps, 1))[:big], dtype) if fx.get("cond") is not None else None
emb = bool(fx.meta["embedding"])
first = pdf(x, conditional_input=cond, force_embedding_coordinates=emb)
small = pdf(x[:n], conditional_input=None if cond is None else cond[:n], force_embedding_coordinates=emb)
ref = small[0].repeat(reps)[:big]
fin = torch.isfinite(ref)
err = ((first[0] - ref).abs() / (1 + ref.abs()))[fin].max().item()
differing = 0
for i in range(launches):
    out = pdf(x, conditional_input=cond, force_embedding_coordinates=emb)
    same = all(bool(((a == b) | (a.isnan() & b.isnan())).all()) for a, b in zip(out, first))
    differing += 0 if same else 1
print("%s %s rows %d: %d of %d launches differ from the first; first vs replicated small batch: worst relative deviation %.3g"
      % (name, str(dtype).split(".")[-1], big, differing, launches, err))
