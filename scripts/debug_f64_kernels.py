import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch, fixture_io, helpers, bench
from jammy_flows_amd import _hip
fx = fixture_io.load("c3_e4s2e4")
for dtype in (torch.float64, torch.float32):
    pdf = helpers.build_product(fx, dtype); pdf.check_status = False
    x = torch.from_numpy(bench.make_inputs(1 << 20, 3)).to(device="cuda", dtype=dtype)
    for _ in range(2): pdf(x)
    t = _hip.KernelTimer()
    with t:
        for _ in range(5): pdf(x)
    print(dtype)
    for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
        print("   %-32s %-22s mean %.3f ms" % (k[0], k[1], v["mean_ms"]))
