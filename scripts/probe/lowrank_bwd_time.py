#!/usr/bin/env python3
"""time of jf_lowrank_gf_chain_inv / _bwd alone at C5's shapes (random parameters of the reference's initial scale)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch
import fixture_io, helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip
fx = fixture_io.load("c5_e8s2_ggggv")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
x64, c64 = inputs(fx, n, 7)
pdf = helpers.build_product(fx, torch.float64)
x = torch.from_numpy(x64).cuda(); c = torch.from_numpy(c64).cuda()
mlp = pdf.mlp_predictors[0]
layers = list(pdf.layer_list[0])
with torch.no_grad():
    t2, u2, b2 = mlp.forward_to_last_rank(c)
larr = _hip.gf_layer_array([l.c_struct() for l in layers])
tgt = x[:, :8].contiguous()
def fwd():
    return _hip.lowrank_gf_chain_inv(t2, u2, b2, tgt, None, larr, len(layers), 8, want_base_logp=True, want_aux=True)
z, ld, blp, aux = fwd()
g_ld = torch.full((n,), -1.0 / n, dtype=torch.float64, device="cuda")
def bwd():
    return _hip.lowrank_gf_chain_inv_bwd(t2, u2, b2, aux, z, larr, len(layers), 8, None, g_ld, g_ld)
def fwd_noaux():
    return _hip.lowrank_gf_chain_inv(t2, u2, b2, tgt, None, larr, len(layers), 8, want_base_logp=True, want_aux=False)
def fused_inference():
    lr = pdf._fusable_lowrank_block(0, layers, False, None, x)
    return _hip.amlp_gf_chain_inv(c, *lr, tgt, None, larr, len(layers), 8, want_base_logp=True)
for name, fn in (("fwd", fwd), ("fwd without aux", fwd_noaux), ("fused inference kernel (with the MLP)", fused_inference), ("bwd", bwd)):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:                     # clocks ramp for ~1 s under load
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): fn()
    torch.cuda.synchronize(); print(name, "%.4f ms" % ((time.perf_counter() - t0) / 100 * 1e3))
