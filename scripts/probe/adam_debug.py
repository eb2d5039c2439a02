"""one Adam step on the C3 gradients: default vs fused=True vs capturable=True"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import numpy as np
import torch
import fixture_io, helpers

fx = fixture_io.load("c3_e4s2e4")
rng = np.random.default_rng(7)
n = 1 << 14
x = np.concatenate([rng.normal(size=(n, 4)) * 1.5, np.arccos(rng.uniform(-1, 1, size=(n, 1))).clip(1e-3, np.pi - 1e-3),
                    rng.uniform(0, 2 * np.pi, size=(n, 1)), rng.normal(size=(n, 4)) * 1.5], axis=1)
x = torch.from_numpy(x).to(device="cuda", dtype=torch.float32)

res = {}
for mode in ("default", "fused", "capturable", "foreach_off"):
    pdf = helpers.build_product(fx, torch.float32)
    kw = dict(fused=True) if mode == "fused" else dict(capturable=True) if mode == "capturable" else dict(foreach=False) if mode == "foreach_off" else {}
    opt = torch.optim.Adam(pdf.parameters(), lr=1e-2, **kw)
    losses = []
    for it in range(5):
        opt.zero_grad(set_to_none=True)
        loss = -pdf(x)[0].mean()
        loss.backward()
        if it == 0 and mode == "default":
            for k, p in pdf.named_parameters():
                print("%-50s shape %-16s stride %-12s grad stride %-12s contiguous %s dtype %s" % (k, tuple(p.shape), p.stride(), None if p.grad is None else p.grad.stride(), p.is_contiguous(), p.dtype))
        opt.step()
        losses.append(loss.item())
    res[mode] = (losses, {k: p.detach().clone() for k, p in pdf.named_parameters()})
    print(mode, ["%.5f" % l for l in losses])
for mode in ("fused", "capturable", "foreach_off"):
    for k, v in res[mode][1].items():
        d = (v - res["default"][1][k]).abs().max().item() if v.numel() else 0.0
        if d > 1e-5:
            print(mode, "param differs", k, d)
