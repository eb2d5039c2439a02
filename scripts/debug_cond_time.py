import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import fixture_io, helpers, bench
from jammy_flows_amd import _hip
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32); pdf.check_status = False
B = 1 << 20
x = torch.from_numpy(bench.make_inputs(B, 3)).to(device="cuda", dtype=torch.float32)
layers = list(pdf.layer_list[2])
mlp = pdf.mlp_predictors[2]
ps = [mlp[0].weight.detach(), mlp[0].bias.detach(), mlp[2].weight.detach(), mlp[2].bias.detach()]
emb = _hip.sphere_embedding(x[:, 4:6], None, 2, True)[0]
inp = torch.cat([x[:, :4], emb], dim=1).contiguous()
tgt = x[:, 6:10]
arr = _hip.gf_layer_array([l.c_struct() for l in layers])
def run(): return _hip.cond_gf_chain_inv(inp, *ps, tgt, None, arr, len(layers), 4)
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("fused cond g-chain (7->128->548, gggg, D=4, 2^20 rows): %.3f ms" % (e0.elapsed_time(e1) / 10))
