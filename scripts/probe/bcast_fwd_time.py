#!/usr/bin/env python3
"""time of the broadcast-regime sampling chain (C3 block 0: e4 gggg, float32) alone"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch
import fixture_io, helpers
from jammy_flows_amd import _hip
fx = fixture_io.load("c3_e4s2e4")
dtype = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else torch.float32
pdf = helpers.build_product(fx, dtype)
layers = list(pdf.layer_list[0])
larr = _hip.gf_layer_array([l.c_struct() for l in layers])
from jammy_flows_amd.layers.euclidean import gaussianization_flow as gfl
params = gfl.chain_permanent_row(layers, torch.zeros(1, dtype=dtype, device="cuda"))
for lg in (20, 18, 16):
    n = 1 << lg
    z = torch.randn((n, 4), dtype=dtype, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    def fn():
        return _hip.gf_chain("fwd", z, None, params, larr, len(layers), 4)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print("bcast fwd 2^%d rows %.4f ms" % (lg, (time.perf_counter() - t0) / 50 * 1e3))
    t = _hip.KernelTimer()
    with t:
        for _ in range(10): fn()
    print("   ", {"%s[%s]" % k: round(v["mean_ms"], 4) for k, v in t.summary().items()})
