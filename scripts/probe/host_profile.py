"""cProfile of eager training steps of one fixture (host side): python3 scripts/probe/host_profile.py <fixture> [rows] [f32|f64]"""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import fixture_io, helpers

name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dtype = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
fx = fixture_io.load(name)
pdf = helpers.build_product(fx, dtype, torch.device("cuda"))
reps = (B + fx["x"].shape[0] - 1) // fx["x"].shape[0]
x = torch.from_numpy(np.tile(fx["x"][:-8], (reps + 1, 1))[:B]).to(device="cuda", dtype=dtype)
c = None if fx.get("cond") is None else torch.from_numpy(np.tile(fx["cond"][:-8], (reps + 1, 1))[:B]).to(device="cuda", dtype=dtype)


def step():
    for p in pdf.parameters():
        p.grad = None
    with torch.enable_grad():
        (-pdf(x, conditional_input=c)[0].mean()).backward()


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
