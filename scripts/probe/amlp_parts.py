#!/usr/bin/env python3
"""Where does jf_amlp_gf_chain_inv_f64 spend its time?  Builds variants of csrc/amlp_gf_kernels.hip with one part of the kernel removed (edited
copies under /tmp, compiled on the GPU box) and times each on the C5 block-0 shape (2^19 rows, K1 16, H 128, rank 8, 4 g layers, D 8).
    python scripts/probe/amlp_parts.py
"""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from jammy_flows_amd import _hip

SRC = open(os.path.join(ROOT, "jammy_flows_amd", "csrc", "amlp_gf_kernels.hip")).read()
VARIANTS = {
    "full": lambda s: s,
    "no_stage1": lambda s: s.replace("for (int j0 = g; j0 < a.H; j0 += 4 * G) {", "for (int j0 = g; j0 < 0; j0 += 4 * G) {"),
    "no_gen_fma": lambda s: s.replace("for (int q = 0; q < RM; ++q) if (q < a.r2) acc += u[q] * t2[q];", "for (int q = 0; q < 1; ++q) acc += t2[0];"),
    "no_mixture": lambda s: s.replace("const MixQ<T> q = ag_mixture<T>(P, o, x, live);",
                                      "MixQ<T> q; { T ss = T(0);\n#pragma unroll\n for (int i_ = 0; i_ < AG_SLOTS; ++i_) ss += P[i_]; q.lc = ss * T(1e-3) - T(1); q.ls = T(-0.5); q.lp = T(-1); q.cdf = T(0.4); q.sf = T(0.6); }"),
}
torch.manual_seed(0)
B, K1, H, r, D, L = 1 << 19, 16, 128, 8, 8, 4
N = L * (3 * 10 * D + D * D) + D
dev = "cuda"
f64 = torch.float64
inp = torch.randn(B, K1, dtype=f64, device=dev)
V1 = torch.randn(r, K1, dtype=f64, device=dev) * 0.3; U1 = torch.randn(H, r, dtype=f64, device=dev) * 0.3; b1 = torch.randn(H, dtype=f64, device=dev) * 0.1
V2 = torch.randn(r, H, dtype=f64, device=dev) * 0.1; U2 = torch.randn(N, r, dtype=f64, device=dev) * 0.1; b2 = torch.randn(N, dtype=f64, device=dev) * 0.5
x = torch.randn(B, D, dtype=f64, device=dev) * 1.5
xo = torch.empty_like(x); ldo = torch.empty(B, dtype=f64, device=dev); blp = torch.empty(B, dtype=f64, device=dev)
layers = (_hip.jf_gf_layer * L)()
for i in range(L):
    s = layers[i]
    s.num_kde, s.hh_iter, s.model_offset, s.fit_normalization, s.regulate_normalization = 10, D, 1 if i == L - 1 else 0, 1, 1
    s.inverse_function_type = 0 if i else 1
    s.width_mode, s.clamp_widths, s.nonlinear_stretch_type = _hip.GF_WIDTH_SMOOTH, 0, 0
    s.width_min, s.width_max, s.norm_min, s.norm_max = 0.01, 100.0, 1.0, 10.0
P = ctypes.c_void_p
for name, edit in VARIANTS.items():
    src = edit(SRC)
    assert name == "full" or src != SRC, name
    d = "/tmp/amlp_%s" % name
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "k.hip"), "w").write(src)
    so = os.path.join(d, "lib.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-shared",
                           "-I", os.path.join(ROOT, "jammy_flows_amd", "csrc"), "-I", os.path.join(ROOT, "include"), os.path.join(d, "k.hip"), "-o", so])
    fn = ctypes.CDLL(so).jf_amlp_gf_chain_inv_f64
    args = (P(inp.data_ptr()), ctypes.c_int64(K1), P(V1.data_ptr()), P(U1.data_ptr()), P(b1.data_ptr()), P(V2.data_ptr()), P(U2.data_ptr()), P(b2.data_ptr()),
            ctypes.c_int32(K1), ctypes.c_int32(H), ctypes.c_int32(r), ctypes.c_int32(r), P(x.data_ptr()), ctypes.c_int64(D), None, ctypes.c_int64(B),
            ctypes.c_int32(D), ctypes.c_int32(L), layers, P(xo.data_ptr()), ctypes.c_int64(D), P(ldo.data_ptr()), None, P(blp.data_ptr()), None,
            P(torch.cuda.current_stream().cuda_stream))
    for _ in range(2):
        rc = fn(*args)
        assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn(*args)
    e1.record(); torch.cuda.synchronize()
    print("%-12s %.3f ms" % (name, e0.elapsed_time(e1) / 5))
