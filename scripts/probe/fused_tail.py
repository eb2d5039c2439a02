#!/usr/bin/env python3
"""fused conditional block at the strong-scaling shard size: one round of workgroups vs 1.33 (2^17 rows), both row-group variants"""
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
for n in (768 * 64, 768 * 128, 768 * 128 + 256 * 64, 1 << 17, 768 * 192, 768 * 256, 1 << 18):
    x64, c64 = inputs(fx, n, 7)
    x = torch.from_numpy(x64).to("cuda", torch.float32)
    r = {"rows": n}
    for rg in (0, 1, 2):
        _hip.lib().jf_cond_gf_split_row_groups(rg)
        pf = pdf.planned_forward(x)
        t = _hip.KernelTimer()
        with t:
            for _ in range(50):
                pf(x)
        k = {kk[0]: round(v["mean_ms"], 4) for kk, v in t.summary().items()}
        r["rg%d" % rg] = k.get("jf_cond_gf_chain_split3_f32")
        del pf
    print(json.dumps(r), flush=True)
