import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from jammy_flows_amd import _hip
B = 1 << 20
x = torch.randn(B, 7, device="cuda"); w1 = torch.randn(128, 7, device="cuda"); b1 = torch.randn(128, device="cuda")
w2 = torch.randn(548, 128, device="cuda") * 0.1; b2 = torch.randn(548, device="cuda")
out = torch.empty(B, 576 if os.environ.get("PAD") else 548, device="cuda")[:, :548]
for _ in range(2): _hip.mlp2(x, w1, b1, w2, b2, out=out)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): _hip.mlp2(x, w1, b1, w2, b2, out=out)
e1.record(); torch.cuda.synchronize()
print("JF_DBG=%s  mlp2 7->128->548: %.3f ms" % (os.environ.get("JF_DBG", "0"), e0.elapsed_time(e1) / 5))
