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
print("--- fused mlp2")
for dt in (torch.float64, torch.float32):
    for (B, K1, H, N) in [(300, 7, 128, 548), (1000, 4, 128, 10), (257, 24, 128, 50), (129, 1, 128, 8), (64, 16, 64, 100), (5000, 32, 96, 1224)]:
        x = torch.randn(B, K1, dtype=dt, device="cuda"); w1 = torch.randn(H, K1, dtype=dt, device="cuda") / K1 ** 0.5; b1 = torch.randn(H, dtype=dt, device="cuda")
        w2 = torch.randn(N, H, dtype=dt, device="cuda") / H ** 0.5; b2 = torch.randn(N, dtype=dt, device="cuda")
        y = _hip.mlp2(x, w1, b1, w2, b2)
        ref = torch.tanh(x.double() @ w1.double().T + b1.double()) @ w2.double().T + b2.double()
        print(dt, (B, K1, H, N), "max err %.3e" % (y.double() - ref).abs().max().item())
