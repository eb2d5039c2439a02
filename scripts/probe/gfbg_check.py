#!/usr/bin/env python3
"""lane = (row, coordinate) broadcast g chain against lane = row: bit identity (C1 e2 f64, C2 e4 f32/f64, D = 3) and kernel times."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import numpy as np, torch, fixture_io, helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip
torch.set_grad_enabled(False)
L = _hip.lib()
for name, dt in (("c2_e4_gggg", torch.float32), ("c2_e4_gggg", torch.float64), ("c1_e2_gg", torch.float64), ("c1_e2_gg", torch.float32)):
    fx = fixture_io.load(name)
    pdf = helpers.build_product(fx, dt)
    pdf.check_status = "deferred"
    for n in (1, 17, 64, 100, 256, 1000, 4099, 1 << 15, (1 << 17) + 3):
        x64, c64 = inputs(fx, n, 11)
        x = torch.from_numpy(x64).to("cuda", dt)
        L.jf_gf_bcast_lane_rows(0)
        ref = pdf(x)
        L.jf_gf_bcast_lane_rows(1 << 40)
        got = pdf(x)
        same = all(bool(torch.equal(a, b)) for a, b in zip(got, ref))
        if not same:
            print("MISMATCH", name, dt, n, [float((a - b).abs().max()) for a, b in zip(got, ref)], flush=True)
    print(name, dt, "ok", flush=True)
    for lg in (20, 18, 17, 16, 13):
        x64, _ = inputs(fx, 1 << lg, 7)
        x = torch.from_numpy(x64).to("cuda", dt)
        r = {"cfg": name, "dtype": str(dt), "log2_rows": lg}
        for label, thr in (("lane_row", 0), ("lane_coord", 1 << 40)):
            L.jf_gf_bcast_lane_rows(thr)
            pf = pdf.planned_forward(x)
            for _ in range(10): pf(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): pf(x)
            torch.cuda.synchronize(); r[label + "_ms"] = round((time.perf_counter() - t0) * 10, 4)
            del pf
        print(json.dumps(r), flush=True)
    L.jf_gf_bcast_lane_rows(-1)
