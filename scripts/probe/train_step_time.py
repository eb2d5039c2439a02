"""forward + backward of one fixture's pdf at a given batch size, per-kernel HIP-event times of the backward launches:
python3 scripts/probe/train_step_time.py <fixture> [rows] [f32|f64]        (e.g. c4_i1s1_ro 262144 f32; A/B two libraries with JF_LIB_PATH)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import fixture_io, helpers
from jammy_flows_amd import _hip

name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
dtype = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
fx = fixture_io.load(name)
pdf = helpers.build_product(fx, dtype, torch.device("cuda"))
import bench
wl = {W["fixture"]: k for k, W in bench.WORKLOADS.items()}.get(name)
if wl is not None:                                   # a BASELINE configuration: the bench's own generated inputs
    x64, c64 = bench.make_inputs(wl, B, bench.WORKLOADS[wl]["seed"])
else:                                                # any other fixture: its rows, tiled
    reps = (B + fx["x"].shape[0] - 1) // fx["x"].shape[0]
    x64 = np.tile(fx["x"], (reps, 1))[:B]
    c64 = None if fx.get("cond") is None else np.tile(fx["cond"], (reps, 1))[:B]
x = torch.from_numpy(x64).to(device="cuda", dtype=dtype)
c = None if c64 is None else torch.from_numpy(c64).to(device="cuda", dtype=dtype)


def step():
    for p in pdf.parameters():
        p.grad = None
    with torch.enable_grad():
        (-pdf(x, conditional_input=c)[0].mean()).backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
with _hip.KernelTimer() as kt:
    for _ in range(5):
        step()
summ = kt.summary()
print("%s %s rows %d: %.4f ms per step;" % (name, "f64" if dtype == torch.float64 else "f32", B, ms),
      {k[0] + "[" + k[1] + "]": round(v["mean_ms"] * v["launches"] / 5, 4) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[:8]})
