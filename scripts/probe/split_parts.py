#!/usr/bin/env python3
"""Where does jf_cond_gf_chain_inv_split_f32 spend its time?  Variants of csrc/cond_split_kernels.hip with one part removed (edited copies under
/tmp, compiled on the GPU box), timed on the C3 block-2 shape (2^20 rows, K1 7, H 128, 4 g layers, D 4).
    python scripts/probe/split_parts.py
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from jammy_flows_amd import _hip

SRC = open(os.path.join(ROOT, "jammy_flows_amd", "csrc", "cond_split_kernels.hip")).read()
FAKE_MIX = ("MixQ<float> q; { float ss = x;\n#pragma unroll\n for (int i_ = 0; i_ < CS_SLOTS; ++i_) ss += P[i_]; q.lc = ss * 1e-3f - 1.f; q.ls = -0.5f; q.lp = -1.f; "
            "q.cdf = 0.4f; q.sf = 0.6f; }")
VARIANTS = {
    "full": lambda s: s,
    "no_mixture": lambda s: s.replace("const MixQ<float> q = cs_mixture(P, o, x, live);", FAKE_MIX),
    "no_icdf": lambda s: s.replace("const IcdfOut<float> sy = gf_icdf<float>(o.inv_type, q);", "IcdfOut<float> sy; sy.y = q.lc - q.ls; sy.logd = q.lp;"),
    "no_tanh": lambda s: s.replace("const float h = M<float>::tanh_fast(acc[j][r] + b1s[j * MT + 4 * lq + r]);", "const float h = acc[j][r] + b1s[j * MT + 4 * lq + r];"),
    "no_mfma2": lambda s: s.replace("for (int s = 0; s < CS_KSTEPS; ++s) {\n                bf16x8 A[CS_CT][CS_NP];", "for (int s = 0; s < 1; ++s) {\n                bf16x8 A[CS_CT][CS_NP];"),
    "no_rotation": lambda s: s.replace("if (i < o.hh) {                                        // x <- Q^T x", "if (i < 0) {                                        // x <- Q^T x"),
}
torch.manual_seed(0)
B, K1, H, D, L = 1 << 20, 7, 128, 4, 4
N = L * (3 * 10 * D + D * D) + D
dev, f32 = "cuda", torch.float32
inp = torch.randn(B, K1, dtype=f32, device=dev)
W1 = torch.randn(H, K1, dtype=f32, device=dev) * 0.3; b1 = torch.randn(H, dtype=f32, device=dev) * 0.1
W2 = torch.randn(N, H, dtype=f32, device=dev) * 0.05; b2 = torch.randn(N, dtype=f32, device=dev) * 0.5
x = torch.randn(B, D, dtype=f32, device=dev) * 1.5
xo = torch.empty_like(x); ldo = torch.empty(B, dtype=f32, device=dev); blp = torch.empty(B, dtype=f32, device=dev)
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
    d = "/tmp/split_%s" % name
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "k.hip"), "w").write(src)
    so = os.path.join(d, "lib.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-shared",
                           "-I", os.path.join(ROOT, "jammy_flows_amd", "csrc"), "-I", os.path.join(ROOT, "include"), os.path.join(d, "k.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.jf_cond_gf_packed_bytes.restype = ctypes.c_int64
    nbytes = lib.jf_cond_gf_packed_bytes(ctypes.c_int32(D), ctypes.c_int32(L), layers)
    packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = P(torch.cuda.current_stream().cuda_stream)
    rc = lib.jf_cond_gf_pack_f32(P(W2.data_ptr()), ctypes.c_int64(H), P(b2.data_ptr()), ctypes.c_int32(H), ctypes.c_int32(D), ctypes.c_int32(L), layers,
                                 P(packed.data_ptr()), st)
    assert rc == 0, rc
    fn = lib.jf_cond_gf_chain_inv_split_f32
    args = (P(inp.data_ptr()), ctypes.c_int64(K1), P(W1.data_ptr()), ctypes.c_int64(K1), P(b1.data_ptr()), P(packed.data_ptr()), ctypes.c_int32(K1),
            ctypes.c_int32(H), P(x.data_ptr()), ctypes.c_int64(D), None, ctypes.c_int64(B), ctypes.c_int32(D), ctypes.c_int32(L), layers, P(xo.data_ptr()),
            ctypes.c_int64(D), P(ldo.data_ptr()), None, P(blp.data_ptr()), None, st)
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
