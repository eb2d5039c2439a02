import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import numpy as np, torch, fixture_io, helpers
from bench_configs_inputs import inputs
torch.set_grad_enabled(False)
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32)
pdf.check_status = "deferred"
def say(*a):
    torch.cuda.synchronize(); print(*a, flush=True)
keep = []
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
for lg in (20, 19, 18):
    x = torch.from_numpy(inputs(fx, 1 << lg, 7)[0]).to(device="cuda", dtype=torch.float32)
    say("inputs", lg, hex(x.data_ptr()))
    pdf(x); say("eager ok")
    pf = pdf.planned_forward(x); say("plan recorded", pf.plan.n_ops, "ops", [hex(t.data_ptr()) for t in pf.out_like])
    for i in range(3):
        o = pf(x); say("replay", i, float(o[0][0]))
    if mode == "keep":
        keep.append(pf)
    if mode != "nograph":
        g = pdf.graphed_forward(x); say("graph captured")
        g.graph.replay(); say("graph replayed")
        del g; say("graph deleted")
    pdf(x); say("eager after ok")
print("done")
