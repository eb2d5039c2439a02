"""probe: per-kernel times of pdf.sample() for C3 (f32, 2^20) and C5 (f64, 2^19)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import fixture_io, helpers
from jammy_flows_amd import _hip
for name, dtype, n in (("c3_e4s2e4", torch.float32, 1 << 20), ("c5_e8s2_ggggv", torch.float64, 1 << 19)):
    fx = fixture_io.load(name)
    pdf = helpers.build_product(fx, dtype)
    c = fx.get("cond")
    cond = None
    if c is not None:
        cond = torch.from_numpy(np.random.default_rng(3).normal(size=(n, c.shape[1]))).to(device="cuda", dtype=dtype)
    with torch.no_grad():
        for _ in range(2):
            pdf.sample(conditional_input=cond, samplesize=n) if cond is None else pdf.sample(conditional_input=cond)
        t = _hip.KernelTimer()
        with t:
            pdf.sample(conditional_input=cond, samplesize=n) if cond is None else pdf.sample(conditional_input=cond)
    print(name, dtype)
    for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
        print("  %-50s x%d %.3f ms" % (k[0] + "[" + k[1] + "]", v["launches"], v["total_ms"]))
