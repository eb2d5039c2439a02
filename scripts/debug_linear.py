import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from jammy_flows_amd import _hip
torch.manual_seed(0)
for dt in (torch.float64, torch.float32):
    for (B, K, N) in [(300, 7, 128), (300, 128, 548), (1000, 16, 8), (257, 128, 1224), (64, 1, 10)]:
        x = torch.randn(B, K, dtype=dt, device="cuda"); w = torch.randn(N, K, dtype=dt, device="cuda") / K ** 0.5; b = torch.randn(N, dtype=dt, device="cuda")
        for act in (0, 1):
            y = _hip.linear(x, w, b, act)
            ref = x.double() @ w.double().T + b.double()
            if act: ref = torch.tanh(ref)
            print(dt, (B, K, N), "act", act, "max err %.3e" % (y.double() - ref).abs().max().item())
