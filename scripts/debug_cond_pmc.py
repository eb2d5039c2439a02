import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import fixture_io, helpers, bench
from jammy_flows_amd import _hip
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32); pdf.check_status = False
B = 1 << 20
nl = int(os.environ.get("NL", "4"))
x = torch.from_numpy(bench.make_inputs(B, 3)).to(device="cuda", dtype=torch.float32)
layers = list(pdf.layer_list[2])[:nl]
mlp = pdf.mlp_predictors[2]
n = sum(l.total_param_num for l in layers)
ps = [mlp[0].weight.detach(), mlp[0].bias.detach(), mlp[2].weight.detach()[:n].contiguous(), mlp[2].bias.detach()[:n].contiguous()]
inp = torch.randn(B, 7, device="cuda")
tgt = x[:, 6:10]
arr = _hip.gf_layer_array([l.c_struct() for l in layers])
for _ in range(3): _hip.cond_gf_chain_inv(inp, *ps, tgt, None, arr, len(layers), 4)
torch.cuda.synchronize()
