#!/usr/bin/env python3
"""float64 amortisation MLP on int8 digit slices (csrc/mlp_i8_kernels.hip) against the float64-MFMA path (jf_mlp2_f64):
agreement of log p / base point on a golden fixture and on 2^20 replicated rows, sampling agreement, and timings of the three arithmetic
choices.  python scripts/probe/i8_check.py [fixture]"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import numpy as np
import torch
from test_gpu_parity import ALL_FIXTURES, build_product, to_dev
from jammy_flows_amd import _hip
from jammy_flows_amd.main import default as jf_default

name = sys.argv[1] if len(sys.argv) > 1 else "c3_e4s2e4"
torch.set_grad_enabled(False)
fx = [f for f in ALL_FIXTURES if f.name == name][0]
n = fx["x"].shape[0]
pdf = build_product(fx, torch.float64)
emb = bool(fx.meta["embedding"])
xs = to_dev(fx["x"], torch.float64)
cs = to_dev(fx["cond"], torch.float64) if fx.get("cond") is not None else None


def run(mode, x, c):
    jf_default.MLP_MATRIX_ARITHMETIC_F64[0] = mode
    timer = _hip.KernelTimer()
    with timer:
        out = pdf(x, conditional_input=c, force_embedding_coordinates=emb)
    summ = timer.summary()
    return out, {k[0]: round(v["mean_ms"], 4) for k, v in summ.items()}


ref, names_ref = run("f64", xs, cs)
gold = torch.as_tensor(np.asarray(fx["logp"]), dtype=torch.float64, device=xs.device)
print("two-launch path vs golden log p: max |d| %.3g   kernels %s" % ((ref[0] - gold).abs().max().item(), names_ref))
for mode in ("i8x6", "i8x5"):
    out, names = run(mode, xs, cs)
    print("%s vs two-launch: log p max |d| %.3g, base point max |d| %.3g; vs golden %.3g   i8 kernel ran: %s"
          % (mode, (out[0] - ref[0]).abs().max().item(), (out[2] - ref[2]).abs().max().item(), (out[0] - gold).abs().max().item(),
             any("i8" in str(k) for k in names)))

reps = (1 << 20) // n + 2
big = reps * n - 41
x = to_dev(np.tile(fx["x"], (reps, 1))[:big], torch.float64)
c = to_dev(np.tile(fx["cond"], (reps, 1))[:big], torch.float64) if cs is not None else None
pdf.check_status = False
base = None
for mode in ("f64", "i8x6", "i8x5"):
    jf_default.MLP_MATRIX_ARITHMETIC_F64[0] = mode
    for _ in range(3):
        out = pdf(x, conditional_input=c, force_embedding_coordinates=emb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = pdf(x, conditional_input=c, force_embedding_coordinates=emb)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    if base is None:
        base = out
    fin = torch.isfinite(base[0])
    print("%-5s %d rows: %.3f ms per evaluation; log p max |d| vs f64 path %.3g" % (mode, big, ms, (out[0] - base[0])[fin].abs().max().item()))
    print("      ", run(mode, x, c)[1])

# sampling: same base noise through the three arithmetic choices
torch.manual_seed(5)
m = 1 << 18
cc = None if c is None else c[:m]
zz = torch.randn(m, np.asarray(fx["z"]).shape[1], dtype=torch.float64, device=x.device)
res = {}
for mode in ("f64", "i8x6", "i8x5"):
    jf_default.MLP_MATRIX_ARITHMETIC_F64[0] = mode
    for it in range(3):
        torch.manual_seed(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = pdf._obtain_sample(conditional_input=cc, predefined_target_input=zz, force_embedding_coordinates=emb)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    res[mode] = s
    print("%-5s sampling %d rows: %.3f ms; sample max |d| vs f64 path %.3g, log p max |d| %.3g"
          % (mode, m, ms, (s[0] - res["f64"][0]).abs().max().item(), (s[2] - res["f64"][2]).abs().max().item()))
