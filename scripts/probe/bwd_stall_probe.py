"""times the C3 conditional block's adjoint launch (cond_gf_split_bwd_kernel) under the timing-only library variants of
scripts/probe/bwd_stall_variants.sh (one child process per library, one training stream, no optimizer step: the variants' gradients are wrong by
construction).  python3 scripts/probe/bwd_stall_probe.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time, json
ROOT = %r
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import numpy as np, torch
import fixture_io, helpers
from benchlib.workloads import make_inputs
from jammy_flows_amd import _hip
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32, torch.device("cuda"))
pdf.check_status = False
x = torch.from_numpy(make_inputs("c3", 1 << 18, 0)[0]).to(device="cuda", dtype=torch.float32)
def step():
    for p in pdf.parameters():
        p.grad = None
    with torch.enable_grad():
        (-pdf(x)[0].mean()).backward()
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20 * 1e3
timer = _hip.KernelTimer()
with timer:
    for _ in range(10):
        step()
torch.cuda.synchronize()
tab = {"%%s[%%s]" %% k: round(v["mean_ms"], 4) for k, v in timer.summary().items()}
print(json.dumps({"step_ms": round(dt, 4), "kernels": tab}))
''' % ROOT
libs = [("product", None)] + [(v, os.path.join(ROOT, "jammy_flows_amd", "_probe", "libjammy_hip_%s.so" % v)) for v in ("nodma", "nobarrier", "both", "occ3")]
for rnd in range(2):
    for name, path in libs:
        env = dict(os.environ, JF_TRAIN_STREAMS="1")
        if path:
            if not os.path.exists(path):
                print(name, "not built"); continue
            env["JF_LIB_PATH"] = path
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        try:
            r = json.loads(out.stdout.strip().splitlines()[-1])
            k = {a: b for a, b in r["kernels"].items() if "split_bwd" in a or "wgrad_split" in a or "split2" in a or "split3" in a}
            print("%-10s step %.3f ms (eager, one stream)  %s" % (name, r["step_ms"], k), flush=True)
        except Exception as e:
            print(name, "FAILED", out.stderr[-400:], flush=True)
