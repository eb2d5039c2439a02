import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import fixture_io, helpers
from jammy_flows_amd.layers.euclidean import gaussianization_flow as gfl
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32); pdf.check_status=False
B = 1 << 20
layers = list(pdf.layer_list[2])
x = torch.randn(B, 4, device="cuda") * 1.5
params = torch.randn(B, 548, device="cuda") * 0.3
for _ in range(3): gfl.run_chain(layers, "inv", x, None, params)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): gfl.run_chain(layers, "inv", x, None, params)
e1.record(); torch.cuda.synchronize()
print("JF_DBG=%s  gf per-sample chain: %.3f ms" % (os.environ.get("JF_DBG", "0"), e0.elapsed_time(e1) / 10))
# broadcast regime: the first sub-pdf's chain with permanent parameters
layers0 = list(pdf.layer_list[0])
row = gfl.chain_permanent_row(layers0, x)
if row is not None:
    for _ in range(3): gfl.run_chain(layers0, "inv", x, None, row)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10): gfl.run_chain(layers0, "inv", x, None, row)
    e1.record(); torch.cuda.synchronize()
    print("JF_TPB=%s  gf broadcast chain: %.3f ms" % (os.environ.get("JF_TPB", "-"), e0.elapsed_time(e1) / 10))
