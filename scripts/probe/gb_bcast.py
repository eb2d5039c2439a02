"""probe: time of jf_gf_chain_inv_bwd (broadcast regime) alone: C2 e4/gggg float32 (and float64), 2^18 rows"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import fixture_io, helpers
from jammy_flows_amd import _hip
for dtype in (torch.float32, torch.float64):
    fx = fixture_io.load("c2_e4_gggg")
    pdf = helpers.build_product(fx, dtype)
    rng = np.random.default_rng(7)
    x = torch.from_numpy(rng.normal(size=(1 << 18, 4)) * 1.5).to(device="cuda", dtype=dtype)
    for _ in range(3):
        pdf.zero_grad(); (-pdf(x)[0].mean()).backward()
    t = _hip.KernelTimer()
    with t:
        for _ in range(5):
            pdf.zero_grad(); (-pdf(x)[0].mean()).backward()
    for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
        print(dtype, k, "%.3f ms" % (v["total_ms"] / v["launches"]))
    g = torch.cat([p.grad.flatten() for p in pdf.parameters()])
    print("grad checksum %.9e" % float(g.double().abs().sum()))
