#!/usr/bin/env python3
"""consecutive (independent) log-prob steps alternating between two streams, each through its own recorded plan: does the tail of one step's
fused block overlap with the next step's side blocks?"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch, fixture_io, helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip
torch.set_grad_enabled(False)
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32)
pdf.check_status = "deferred"
for lg in (20, 19, 18, 17, 16, 15):
    n = 1 << lg
    x64, c64 = inputs(fx, n, 7)
    x = torch.from_numpy(x64).to("cuda", torch.float32)
    r = {"log2_rows": lg}
    for depth in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(depth)]
        plans = []
        for s in streams:
            with torch.cuda.stream(s):
                plans.append(pdf.planned_forward(x))
        torch.cuda.synchronize()
        steps = 60 if lg >= 18 else 240

        def run(k):
            for i in range(k):
                j = i % depth
                with torch.cuda.stream(streams[j]):
                    plans[j](x)
        run(12)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            run(steps)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps)
        r["depth%d_ms" % depth] = round(best * 1e3, 4)
        ref = pdf(x)[0]
        with torch.cuda.stream(streams[-1]):
            got = plans[-1](x)[0]
        torch.cuda.synchronize()
        r["same%d" % depth] = bool(torch.equal(ref, got))
        del plans
    print(json.dumps(r), flush=True)
