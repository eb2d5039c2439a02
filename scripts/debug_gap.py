import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch, fixture_io, helpers, bench
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32)
x = torch.from_numpy(bench.make_inputs(1 << 20, 3)).to(device="cuda", dtype=torch.float32)
for cs in (True, False):
    pdf.check_status = cs
    for _ in range(3): pdf(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pdf(x)
    torch.cuda.synchronize(); print("check_status", cs, "%.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
