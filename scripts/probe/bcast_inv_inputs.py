#!/usr/bin/env python3
"""jf_gf_chain_inv (broadcast, float32, C2's e4 gggg) per 2^20 rows on (a) the SURVEY inputs (N(0, 1.5^2) per coordinate), (b) samples of the
model itself, (c) the SURVEY inputs with the tail rows (|x| > 4) replaced by copies of central rows: what the tail rows cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch
import fixture_io, helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip
from jammy_flows_amd.layers.euclidean import gaussianization_flow as gfl
torch.set_grad_enabled(False)
fx = fixture_io.load("c2_e4_gggg")
pdf = helpers.build_product(fx, torch.float32)
layers = list(pdf.layer_list[0])
larr = _hip.gf_layer_array([l.c_struct() for l in layers])
B = 1 << 20
x_np, _ = inputs(fx, B, 7)
xa = torch.from_numpy(x_np).to(device="cuda", dtype=torch.float32)
params = gfl.chain_permanent_row(layers, xa)
z = torch.randn(B, 4, device="cuda")
xb, _ = _hip.gf_chain("fwd", z, None, params, larr, len(layers), 4)
xc = xa.clone()
far = (xa.abs() > 4).any(dim=1)
xc[far] = xa[~far][:int(far.sum())]
print("rows with a coordinate beyond 4: %d of %d" % (int(far.sum()), B))
fn0 = lambda: _hip.gf_chain("inv", xa, None, params, larr, len(layers), 4)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    fn0()
for name, x in (("survey", xa), ("model samples", xb), ("survey without tail rows", xc)):
    st = _hip.new_status(x.device)
    fn = lambda: _hip.gf_chain("inv", x, None, params, larr, len(layers), 4)
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): fn()
    torch.cuda.synchronize()
    print("%-28s %.4f ms" % (name, (time.perf_counter() - t0) / 100 * 1e3))
