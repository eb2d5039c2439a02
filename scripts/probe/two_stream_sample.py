#!/usr/bin/env python3
"""sampling steps (independent base batches) alternating between streams: C3 f32 2^20, C5 f64 2^19"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch, fixture_io, helpers
from bench_configs_inputs import inputs
torch.set_grad_enabled(False)
for name, dt, lg in (("c3_e4s2e4", torch.float32, 20), ("c5_e8s2_ggggv", torch.float64, 19), ("c3_e4s2e4", torch.float32, 17)):
    fx = fixture_io.load(name)
    pdf = helpers.build_product(fx, dt)
    pdf.check_status = False
    n = 1 << lg
    x64, c64 = inputs(fx, n, 7)
    c = None if c64 is None else torch.from_numpy(c64).to("cuda", dt)
    z = torch.randn((n, pdf.total_base_dim), dtype=dt, device="cuda")
    ref = pdf._obtain_sample(conditional_input=c, predefined_target_input=z)[0]
    r = {"cfg": name, "log2_rows": lg}
    for depth in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(depth)]
        def run(k):
            outs = None
            for i in range(k):
                with torch.cuda.stream(streams[i % depth]):
                    outs = pdf._obtain_sample(conditional_input=c, predefined_target_input=z)
            return outs
        run(6); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); out = run(30); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 30)
        r["depth%d_ms" % depth] = round(best * 1e3, 4)
        r["same%d" % depth] = bool(torch.equal(out[0], ref))
    print(json.dumps(r), flush=True)
