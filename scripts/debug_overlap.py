"""Do the MFMA-bound MLP kernel and the VALU/HBM-bound g-chain kernel overlap when launched on two streams?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import fixture_io, helpers
from jammy_flows_amd import _hip
from jammy_flows_amd.layers.euclidean import gaussianization_flow as gfl
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32); pdf.check_status = False
B = 1 << 20
layers = list(pdf.layer_list[2])
x = torch.randn(B, 4, device="cuda") * 1.5
inp = torch.randn(B, 7, device="cuda")
mlp = pdf.mlp_predictors[2]
ps = [mlp[0].weight.detach(), mlp[0].bias.detach(), mlp[2].weight.detach(), mlp[2].bias.detach()]
params = _hip.mlp2(inp, *ps)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    torch.cuda.synchronize(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def seq():
    _hip.mlp2(inp, *ps, out=params); gfl.run_chain(layers, "inv", x, None, params)
def par():
    with torch.cuda.stream(s1): _hip.mlp2(inp, *ps, out=params)
    with torch.cuda.stream(s2): gfl.run_chain(layers, "inv", x, None, params)
def par_rev():
    with torch.cuda.stream(s2): gfl.run_chain(layers, "inv", x, None, params)
    with torch.cuda.stream(s1): _hip.mlp2(inp, *ps, out=params)
print("sequential mlp2 + gchain: %.3f ms" % timeit(seq))
print("two streams (mlp first): %.3f ms" % timeit(par))
print("two streams (gchain first): %.3f ms" % timeit(par_rev))
