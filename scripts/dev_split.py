"""dev check: split-bf16 fused block vs exact-f32 fused block vs two-launch path vs golden float64 values."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import helpers
from helpers import ALL_FIXTURES, build_product, to_dev
from jammy_flows_amd import _hip

for name in ["c3_e4s2e4", "c3b_e4s2e4_fsplines", "g_e3_ggg_cond"]:
    fx = [f for f in ALL_FIXTURES if f.name == name][0]
    pdf = build_product(fx, torch.float32)
    pdf.check_status = False
    x = to_dev(fx["x"], torch.float32); cond = to_dev(fx.get("cond"), torch.float32)
    emb = bool(fx.meta["embedding"])
    out = {}
    for mode in ("two", "f32", "split_bf16"):
        pdf.fuse_conditional_blocks = mode != "two"
        pdf.fused_matrix_arithmetic = mode
        t = _hip.KernelTimer()
        with t:
            out[mode] = pdf(x, conditional_input=cond, force_embedding_coordinates=emb)
        kern = sorted(set(k[0] for k in t.summary()))
        ref = fx["logp"]
        got = out[mode][0].double().cpu().numpy()
        fin = np.isfinite(got)
        err = np.abs(got - ref)[fin]
        print(name, mode, "finite", fin.sum(), "/", len(fin), "max|dlogp| vs golden %.3e" % err.max(), "median %.2e" % np.median(err),
              [k for k in kern if "cond" in k])
    d = (out["split_bf16"][0] - out["f32"][0]).abs()
    d = d[torch.isfinite(d)]
    print(name, "split vs f32 fused: max %.3e" % d.max().item(), " base: %.3e" % (out["split_bf16"][2] - out["f32"][2]).abs().nan_to_num(0).max().item())
