#!/usr/bin/env python3
"""broadcast-regime g adjoint (C3 block 0, float32, 2^18 rows): SURVEY 8d inputs against samples of the model itself (no tail rows)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch
import fixture_io, helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip
from jammy_flows_amd.layers.euclidean import gaussianization_flow as gfl
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32)
layers = list(pdf.layer_list[0])
larr = _hip.gf_layer_array([l.c_struct() for l in layers])
params = gfl.chain_permanent_row(layers, torch.zeros(1, dtype=torch.float32, device="cuda"))
n = 1 << 18
x64, _ = inputs(fx, n, 7)
xs = torch.from_numpy(x64[:, :4]).to(device="cuda", dtype=torch.float32).contiguous()
z = torch.randn((n, 4), dtype=torch.float32, device="cuda")
xm = _hip.gf_chain("fwd", z, None, params, larr, len(layers), 4)[0]
g = torch.full((n,), -1.0 / n, dtype=torch.float32, device="cuda")
for name, x in (("survey inputs", xs), ("model samples", xm)):
    def fn():
        return _hip.gf_chain_inv_bwd(x, params, larr, len(layers), 4, None, g, g)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.7:
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize()
    print("%-14s %.4f ms   |x| max %.1f" % (name, (time.perf_counter() - t0) / 50 * 1e3, x.abs().max().item()))
