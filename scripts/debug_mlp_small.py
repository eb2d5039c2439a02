import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from jammy_flows_amd import _hip
B = 1 << 20
xx = torch.randn(B, 10, device="cuda"); x = xx[:, :4]
w1 = torch.randn(128, 4, device="cuda"); b1 = torch.randn(128, device="cuda")
w2 = torch.randn(10, 128, device="cuda") * 0.1; b2 = torch.randn(10, device="cuda")
for _ in range(2): _hip.mlp2(x, w1, b1, w2, b2)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): _hip.mlp2(x, w1, b1, w2, b2)
e1.record(); torch.cuda.synchronize()
print("grid %s  mlp2 4->128->10: %.3f ms" % (os.environ.get("JF_GRID", "auto"), e0.elapsed_time(e1) / 10))
