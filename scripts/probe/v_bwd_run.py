"""A few training steps of C5's 'v' block alone (jf_v_chain_inv_bwd_f64), for rocprofv3 runs.  Usage: python3 scripts/probe/v_bwd_run.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
import jammy_flows_amd as jf

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
torch.manual_seed(0)
pdf = jf.pdf("s2", "v", conditional_input_dim=16).double().cuda()
g = torch.Generator().manual_seed(1)
x = torch.cat([torch.rand(B, 1, generator=g, dtype=torch.float64) * 2.8 + 0.15, torch.rand(B, 1, generator=g, dtype=torch.float64) * 6.0 + 0.1], 1).cuda()
c = torch.randn(B, 16, generator=g, dtype=torch.float64).cuda()
from jammy_flows_amd import _hip
for it in range(8):
    if it == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    for p in pdf.parameters():
        p.grad = None
    with torch.enable_grad():
        (-pdf(x, conditional_input=c)[0].mean()).backward()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
with _hip.KernelTimer() as kt:
    for it in range(5):
        for p in pdf.parameters():
            p.grad = None
        with torch.enable_grad():
            (-pdf(x, conditional_input=c)[0].mean()).backward()
summ = kt.summary()
vb = [v["mean_ms"] for k, v in summ.items() if "v_chain_inv_bwd" in k[0]]
print("rows %d: step %.4f ms (forward + backward of the block, MLP included); jf_v_chain_inv_bwd_f64 %.4f ms" % (B, ms, vb[0] if vb else float("nan")))
