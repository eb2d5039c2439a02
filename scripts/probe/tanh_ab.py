#!/usr/bin/env python3
"""kernel times of the float64 MLP kernels whose hidden layer is a tanh (HIP events, 2^20 rows), and their error against torch float64"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
from jammy_flows_amd import _hip
torch.manual_seed(0)
B = 1 << 20
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = {}
for K1, H, N in ((7, 128, 548), (4, 128, 10)):
    x = torch.randn(B, K1, dtype=torch.float64, device=dev)
    w1 = torch.randn(H, K1, dtype=torch.float64, device=dev) * 0.5; b1 = torch.randn(H, dtype=torch.float64, device=dev) * 0.1
    w2 = torch.randn(N, H, dtype=torch.float64, device=dev) * 0.2; b2 = torch.randn(N, dtype=torch.float64, device=dev) * 0.1
    ref = torch.tanh(x[:4096] @ w1.t() + b1) @ w2.t() + b2
    o = torch.empty(B, N, dtype=torch.float64, device=dev)
    out["mlp2_f64_K%d_N%d_ms" % (K1, N)] = round(t(lambda: _hip.mlp2(x, w1, b1, w2, b2, out=o)), 4)
    out["mlp2_f64_K%d_N%d_err" % (K1, N)] = float((o[:4096] - ref).abs().max())
    if N > 64:
        for S in (5, 6):
            img = _hip.mlp2_i8_pack(w2, b2, S)
            out["mlp2_i8x%d_ms" % S] = round(t(lambda: _hip.mlp2_i8(x, w1, b1, img, N, S, out=o)), 4)
            out["mlp2_i8x%d_err" % S] = float((o[:4096] - ref).abs().max())
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import fixture_io, helpers
from bench_configs_inputs import inputs
torch.set_grad_enabled(False)
fx = fixture_io.load("c5_e8s2_ggggv")
pdf = helpers.build_product(fx, torch.float64)
pdf.check_status = "deferred"
x64, c64 = inputs(fx, 1 << 19, 7)
x = torch.from_numpy(x64).cuda(); c = torch.from_numpy(c64).cuda()
pf = pdf.planned_forward(x, conditional_input=c)
tm = _hip.KernelTimer()
with tm:
    for _ in range(10): pf(x, c)
out["c5_kernels_ms"] = {k[0]: round(v["mean_ms"], 4) for k, v in tm.summary().items()}
o = helpers.build_oracle(fx).forward(x64[:2048], c64[:2048])[0]
out["c5_err"] = float(abs(pf(x, c)[0][:2048].cpu().numpy() - o).max())
print(json.dumps(out))
