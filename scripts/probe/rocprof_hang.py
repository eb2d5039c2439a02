import os, sys, faulthandler
faulthandler.dump_traceback_later(35, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch, fixture_io, helpers
from jammy_flows_amd import _hip
v = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
if len(sys.argv) > 3 and sys.argv[3] == "dup":
    sys.stdout.flush(); os.dup(1); os.dup2(2, 1)
torch.set_grad_enabled(False)
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32, "cuda")
pdf.check_status = "deferred"
x = torch.randn(N, 10, device="cuda")
x[:, 4] = torch.rand(N, device="cuda") * 3.0 + 0.05
x[:, 5] = torch.rand(N, device="cuda") * 6.0 + 0.05
print(v, "start", flush=True)
if v == "eager":
    for _ in range(2): pdf(x)
    pdf.flush_status()
elif v == "pool":
    h, s = _hip.mapped_status(torch.device("cuda"))
    pool = torch.cuda.MemPool()
    for _ in range(2): pdf(x)
    pdf.flush_status()
elif v == "mapped_only":
    h, s = _hip.mapped_status(torch.device("cuda"))
    for _ in range(2): pdf(x)
    pdf.flush_status()
elif v == "unfused":
    pdf.fuse_conditional_blocks = False
    for _ in range(2): pdf(x)
    pdf.flush_status()
elif v == "c2":
    fx2 = fixture_io.load("c2_e4_gggg")
    p2 = helpers.build_product(fx2, torch.float32, "cuda")
    x2 = torch.randn(N, 4, device="cuda")
    for _ in range(2): p2(x2)
    p2.flush_status()
elif v == "c2f64":
    fx2 = fixture_io.load("c2_e4_gggg")
    p2 = helpers.build_product(fx2, torch.float64, "cuda")
    x2 = torch.randn(N, 4, device="cuda", dtype=torch.float64)
    for _ in range(2): p2(x2)
    p2.flush_status()
elif v == "plan":
    pdf.use_step_plans = True
    for _ in range(3): pdf(x)
    pdf.flush_status()
torch.cuda.synchronize()
print(v, "done", flush=True)
