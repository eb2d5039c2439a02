#!/usr/bin/env python3
"""Row groups per wave of the split-bf16 block kernel (JF_CS_RG = 1 | 2, read once per process) against batch size: run as
    JF_CS_RG=1 python scripts/probe/rg_sweep.py; JF_CS_RG=2 python scripts/probe/rg_sweep.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import jammy_flows_amd

torch.manual_seed(0)
pdf = jammy_flows_amd.pdf("e4+s2+e4", "gggg+f+gggg").float().cuda()
pdf.check_status = False
torch.set_grad_enabled(False)
out = []
for lg in range(13, 21):
    B = 1 << lg
    x = torch.randn(B, 10, device="cuda")
    x[:, 4] = torch.rand(B, device="cuda") * 3.0 + 0.07
    x[:, 5] = torch.rand(B, device="cuda") * 6.2
    for _ in range(5):
        pdf(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        pdf(x)
    e1.record()
    torch.cuda.synchronize()
    out.append("2^%d %.4f" % (lg, e0.elapsed_time(e1) / n))
print("RG", os.environ.get("JF_CS_RG"), " ".join(out))
