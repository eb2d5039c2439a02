import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import fixture_io, helpers
from jammy_flows_amd import _hip
fx = fixture_io.load("c5_e8s2_ggggv")
for dtype in (torch.float64,):
    pdf = helpers.build_product(fx, dtype); pdf.check_status = False
    n = 1 << 19
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.normal(size=(n, 8)) * 1.5, np.arccos(rng.uniform(-1, 1, size=(n, 1))).clip(1e-3, np.pi - 1e-3), rng.uniform(0, 2 * np.pi, size=(n, 1))], axis=1)
    x = torch.from_numpy(x).to(device="cuda", dtype=dtype); c = torch.randn(n, 16, device="cuda", dtype=dtype)
    for _ in range(2): pdf(x, conditional_input=c)
    t = _hip.KernelTimer()
    with t:
        for _ in range(5): pdf(x, conditional_input=c)
    for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
        print("%-40s %-24s n=%d mean %.3f ms" % (k[0], k[1], v["launches"], v["mean_ms"]))
