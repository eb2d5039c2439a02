#!/usr/bin/env python3
"""time of a reparameterised-sampling training step (pdf._differentiable_sample + backward) -- python3 scripts/probe/sample_grad_time.py [fixture] [rows] [old]
`old`: import the package from scripts/probe/oldtree/ instead (an A/B against an earlier commit: `git archive <rev> jammy_flows_amd | tar -x -C
scripts/probe/oldtree` and copy the current libjammy_hip.so next to its _hip.py; the directory is not kept in the tree)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
old = len(sys.argv) > 3 and sys.argv[3] == "old"
sys.path[:0] = [os.path.join(ROOT, "scripts", "probe", "oldtree") if old else ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch
import fixture_io, helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip
name = sys.argv[1] if len(sys.argv) > 1 else "c3_e4s2e4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
fx = fixture_io.load(name)
dtype = torch.float64
pdf = helpers.build_product(fx, dtype)
pdf.check_status = False
_, c64 = inputs(fx, n, 7)
c = None if c64 is None else torch.from_numpy(c64).to(device="cuda", dtype=dtype)
z = torch.randn((n, pdf.total_base_dim), dtype=dtype, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
w = torch.linspace(-1, 1, pdf.total_target_dim, dtype=dtype, device="cuda")
def step():
    for p in pdf.parameters():
        p.grad = None
    with torch.enable_grad():
        x, _, logp, _ = pdf._differentiable_sample(conditional_input=c, predefined_target_input=z)
        loss = (x * w).sum(dim=1).mean() + 0.1 * logp.mean()
    loss.backward()
    return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print(name, n, "old" if old else "new", "%.3f ms per step" % ((time.perf_counter() - t0) / 10 * 1e3))
t = _hip.KernelTimer()
with t:
    step()
names = sorted({k[0] for k in t.summary()})
print("   launches per step:", sum(v["launches"] for v in t.summary().values()), "cot kernel used:", any("cot" in k for k in names))
