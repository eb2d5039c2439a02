"""CPU-side tests of the host logic: pdf construction rules, parameter bookkeeping, state_dict compatibility with the reference,
the C-ABI library exports, and the "no CPU fallback" contract.  No kernel is launched here (no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import helpers
from helpers import ALL_FIXTURES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUPPORTED = [fx for fx in ALL_FIXTURES if helpers.product_supports(fx)]


def test_library_exports_every_declared_symbol():
    from jammy_flows_amd import _hip
    header = open(os.path.join(ROOT, "include", "jammy_hip.h")).read()
    declared = set(re.findall(r"\bint(?:64_t|32_t)?\s+(jf_[a-z0-9_]+)\s*\(", header))
    for fam, suffix in re.findall(r"^JF_DECLARE_MCHAIN\((\w+),\s*\w+,\s*(\w+)\)", header, flags=re.M):   # macro-declared chain entry points
        declared |= {"jf_%s_chain_inv_%s" % (fam, suffix), "jf_%s_chain_fwd_%s" % (fam, suffix), "jf_%s_chain_inv_sum_%s" % (fam, suffix)}
    for suffix in re.findall(r"^JF_DECLARE_T\(\w+,\s*(\w+)\)", header, flags=re.M):
        declared |= {"jf_t_layer_inv_" + suffix, "jf_t_layer_fwd_" + suffix, "jf_t_layer_inv_bwd_" + suffix}
    for suffix in re.findall(r"^JF_DECLARE_AMLP\(\w+,\s*(\w+)\)", header, flags=re.M):
        declared |= {"jf_amlp_stage_" + suffix, "jf_amlp_stage_bwd_" + suffix}
    for fam, suffix in re.findall(r"^JF_DECLARE_COND_MCHAIN\((\w+),\s*\w+,\s*(\w+)\)", header, flags=re.M):
        declared |= {"jf_cond_%s_chain_inv_%s" % (fam, suffix), "jf_cond_%s_chain_fwd_%s" % (fam, suffix)}
    for fam, suffix in re.findall(r"^JF_DECLARE_MCHAIN_BWD\((\w+),\s*\w+,\s*(\w+)\)", header, flags=re.M):
        declared.add("jf_%s_chain_inv_bwd_%s" % (fam, suffix))
    assert declared == set(_hip.exported_symbols()), declared ^ set(_hip.exported_symbols())
    lib = _hip.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.jf_abi_version() >= 1


def test_gf_layer_struct_matches_header():
    from jammy_flows_amd import _hip
    # 12 int32 + 4 double, no padding surprises: sizeof must be 12*4 + 4*8
    assert ctypes.sizeof(_hip.jf_gf_layer) == 80
    assert _hip.lib().jf_abi_version() >= 2           # the struct grew with ABI 2 (rotation_mode / center_mean / add_skewness)


@pytest.mark.parametrize("fx", SUPPORTED, ids=[f.name for f in SUPPORTED])
def test_construction_matches_reference_bookkeeping(fx):
    import jammy_flows_amd
    pdf = jammy_flows_amd.pdf(fx.pdf_defs, fx.flow_defs, **fx.kwargs)
    assert [[l.total_param_num for l in blk] for blk in pdf.layer_list] == fx.meta["layer_param_nums"]
    assert [[type(l).__name__ for l in blk] for blk in pdf.layer_list] == fx.meta["layer_types"]
    assert pdf.total_base_dim == fx.meta["total_base_dim"]
    assert pdf.total_target_dim == fx.meta["total_target_dim"]
    assert pdf.total_target_dim_embedded == fx.meta["total_target_dim_embedded"]
    assert pdf.count_parameters() == fx.meta["count_parameters"]
    ref_sd = fx.state_dict()
    sd = pdf.state_dict()
    assert set(sd.keys()) == set(ref_sd.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref_sd[k].shape), k
    pdf.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ref_sd.items()}, strict=True)


def test_first_layer_rules():
    """first g layer of a multi-layer e-block -> inormal_partly_precise, last one gets the offset (main/default.py:440-448)."""
    import jammy_flows_amd
    pdf = jammy_flows_amd.pdf("e4+e1", "gggg+g", conditional_input_dim=2)
    types = [l.inverse_function_type for l in pdf.layer_list[0]]
    assert types == ["inormal_partly_precise", "isigmoid", "isigmoid", "isigmoid"]
    assert [l.model_offset for l in pdf.layer_list[0]] == [0, 0, 0, 1]
    assert pdf.layer_list[1][0].model_offset == 1 and pdf.layer_list[1][0].inverse_function_type == "isigmoid"
    assert pdf.mlp_predictors[1][0].in_features == 2 + 4


def test_options_overwrite_precedence():
    import jammy_flows_amd
    ow = {"g": {"num_kde": 7}, 1: {"g": {"num_kde": 3}}, (1, 1): {"g": {"num_kde": 4}}}
    pdf = jammy_flows_amd.pdf("e2+e2", "gg+gg", options_overwrite=ow)
    assert [[l.num_kde for l in blk] for blk in pdf.layer_list] == [[7, 7], [3, 4]]
    with pytest.raises(AssertionError):
        jammy_flows_amd.pdf("e2", "gg", options_overwrite={"g": {"num_kde": -1}})
    with pytest.raises(AssertionError):
        jammy_flows_amd.pdf("e2", "gg", options_overwrite={"g": {"no_such_option": 1}})


def test_no_cpu_fallback():
    import jammy_flows_amd
    from jammy_flows_amd._hip import HipUnavailable
    pdf = jammy_flows_amd.pdf("e2", "gg").double()
    with pytest.raises(HipUnavailable):
        pdf(torch.zeros(4, 2, dtype=torch.float64))
    with pytest.raises(HipUnavailable):
        pdf.sample(samplesize=4)
    layer = pdf.layer_list[0][0]
    with pytest.raises(HipUnavailable):
        layer.inv_flow_mapping([torch.zeros(4, 2, dtype=torch.float64), torch.zeros(4, dtype=torch.float64)])


def test_unsupported_things_fail_loudly():
    import jammy_flows_amd
    with pytest.raises(NotImplementedError):
        jammy_flows_amd.pdf("e2", "c")                      # continuous manifold flow: needs torchdiffeq, outside the hot path
    p = jammy_flows_amd.pdf("e3", "gggt")                   # the docs' recommended Euclidean setting constructs (suggested_settings.rst:12-42)
    assert [type(l).__name__ for l in p.layer_list[0]] == ["gf_block"] * 3 + ["mvn_block"]
    p = jammy_flows_amd.pdf("e2", "gg", options_overwrite={"g": {"add_skewness": 1, "center_mean": 1, "rotation_mode": "cayley"}})
    assert [l.total_param_num for l in p.layer_list[0]] == [1 + 18 + 60, 2 + 1 + 18 + 60] and all(l.has_extended_options for l in p.layer_list[0])
    # round 4: the Poisson head constructs (main/default.py:467-477, 624-627) -- and keeps the reference's own restrictions
    p = jammy_flows_amd.pdf("e2", "gg", predict_log_normalization=True)
    assert tuple(p.log_normalization.shape) == (1, 1) and "log_normalization" in p.state_dict()
    p = jammy_flows_amd.pdf("e2", "gg", conditional_input_dim=3, predict_log_normalization=True, join_poisson_and_pdf_description=True)
    assert p.mlp_predictors[0][2].out_features == sum(p.num_parameter_list[0]) + 1
    with pytest.raises(AssertionError):
        jammy_flows_amd.pdf("e2+e1", "gg+g", predict_log_normalization=True)          # one sub-pdf only (:471-472)


def test_data_init_host_math_matches_the_oracle():
    """the host-side pieces of init_params(data=...) (init_fns.py: Householder product, 't' lower-triangular matrix at default width options,
    reverse-KL loss) against the oracle's restatements of the same reference functions"""
    from jammy_flows_amd import init_fns
    from oracle import mvn as omvn
    from oracle.special import householder_matrix
    rng = np.random.default_rng(3)
    for D, n_iter in ((2, 2), (3, 3), (5, 2)):
        vs = rng.normal(size=(n_iter, D))
        assert np.abs(init_fns._householder_matrix(vs) - householder_matrix(vs[None])[0]).max() < 1e-13
    for D, cov in ((3, "full"), (4, "diagonal"), (2, "diagonal_symmetric")):
        n = {"full": D + D * (D - 1) // 2, "diagonal": D, "diagonal_symmetric": 1}[cov]
        a = rng.normal(size=n)
        spec = omvn.TSpec(D, {"cov_type": cov, "softplus_for_width": 0, "width_smooth_saturation": 1, "lower_bound_for_widths": 0.01,
                              "upper_bound_for_widths": 100, "clamp_widths": 0, "skip_model_offset": 0}, 0)
        L = init_fns._mvn_lower_triangular(D, a, cov)
        # the oracle applies L to base points in the sampling direction: x = L z
        z = rng.normal(size=(7, D))
        x, ld = omvn.forward(spec, z, np.zeros(7), a[None, :])[:2]
        assert np.abs(z @ L.T - x).max() < 1e-12, cov
        assert np.abs(ld - np.log(np.abs(np.linalg.det(L)))).max() < 1e-12, cov
        target = L @ L.T
        assert abs(init_fns._mvn_loss(target, cov)(a)) < 1e-10          # reverse KL of a distribution with itself


def test_float64_log_of_the_mixture_sums_formula():
    """csrc/jf_math.h M<double>::log_fast restated in numpy (same reduction, same constants, without the fused multiply-adds): within 1.5 ulp
    of numpy's log over 600 decades, denormals included"""
    rng = np.random.default_rng(0)
    x = np.concatenate([10.0 ** rng.uniform(-307, 300, 200000), rng.uniform(0.5, 2.0, 200000), 1.0 + rng.uniform(-1e-6, 1e-6, 50000),
                        np.array([5e-324, 2.2e-308, 1.0, 0.7071067811865475, 0.7071067811865476, 1.7976931348623157e308])])
    m, e = np.frexp(x)
    low = m < 0.70710678118654752440
    m = np.where(low, m + m, m)
    e = np.where(low, e - 1, e).astype(np.float64)
    f = m - 1.0
    s = f / (2.0 + f)
    z = s * s
    w = z * z
    t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01))
    t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)))
    R, hfsq = t1 + t2, 0.5 * f * f
    r = e * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + e * 1.90821492927058770002e-10)) - f)
    ref = np.log(x)
    ulp = np.spacing(np.abs(ref))
    assert float((np.abs(r - ref) / np.maximum(ulp, 5e-324)).max()) <= 1.5


def test_build_flags_really_disable_packed_f32(tmp_path):
    """csrc/Makefile passes -packed-fp32-ops through -Xclang; clang prints "not a recognized feature" for it, so check that the
    backend honours it: a loop that hipcc packs into v_pk_*_f32 by default must come out without them under the library's own CXXFLAGS"""
    import re
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("hipcc not available")
    mk = open(os.path.join(ROOT, "jammy_flows_amd", "csrc", "Makefile")).read()
    flags = re.search(r"^CXXFLAGS := (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    assert "-packed-fp32-ops" in flags
    src = tmp_path / "pk.hip"
    src.write_text("#include <hip/hip_runtime.h>\n__global__ void k(float2* o, const float2* a, const float2* b) {\n"
                   "  const int i = threadIdx.x; float2 x = a[i], y = b[i];\n"
                   "  for (int j = 0; j < 8; ++j) { x.x = x.x * y.x + y.x; x.y = x.y * y.y + y.y; y.x *= x.x; y.y *= x.y; }\n  o[i] = x; }\n")
    def count(extra):
        out = tmp_path / "pk.s"
        subprocess.check_call([hipcc] + extra + ["-S", "--cuda-device-only", str(src), "-o", str(out)], stderr=subprocess.DEVNULL)
        return len(re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", out.read_text()))
    assert count(["--offload-arch=gfx950", "-O3"]) > 0, "the probe loop is no longer packed by default: pick another one"
    assert count([f for f in flags if f not in ("-fPIC", "-Wall", "-Wno-unused-function")]) == 0


def test_dense_kernel_selection_rules():
    """host-side choice between the split-bf16 / streaming / tiled dense kernels (no launches: shapes, dtypes, alignment only)"""
    from jammy_flows_amd import _hip
    x = torch.zeros((4096, 128), dtype=torch.float32)
    w = torch.zeros((548, 128), dtype=torch.float32)
    b = torch.zeros(548, dtype=torch.float32)
    assert _hip.linear_split_ok(x, w, b)
    assert _hip.linear_split_ok(torch.zeros((5, 548)), w.t())                 # a transposed weight view is fine (packed with its strides)
    assert not _hip.linear_split_ok(x.double(), w.double())                  # float32 only
    assert not _hip.linear_split_ok(torch.zeros((8, 6)), torch.zeros((8, 6)))  # K, N must be multiples of 4
    assert not _hip.linear_split_ok(x[:, 1:125], w[:, 1:125])                # unaligned rows
    assert not _hip.linear_split_ok(x, torch.zeros((16384, 128)))            # the bias row of such a layer would not fit next to the chunks
    lib = _hip.lib()
    assert lib.jf_abi_version() >= 3
    # the partial-slab count depends on the shape (streaming kernel for min(N, K) <= 16, tiled MFMA kernel otherwise)
    assert lib.jf_linear_wgrad_splits_f32(1 << 18, 128, 548) >= 1 and lib.jf_linear_wgrad_splits_f64(1 << 17, 8, 1224) >= 1
    assert lib.jf_linear_wgrad_split_splits(1 << 18, 548) * ((548 + 127) // 128) <= 1030
    assert lib.jf_mlp2_small_bwd_slabs(1 << 18) == 4096 and lib.jf_mlp2_small_bwd_slabs(10) == 1
    assert lib.jf_linear_split_packed_bytes(548, 128) == 12 * 3 * 4 * 3 * 1024 and lib.jf_linear_split_packed_bytes(128, 548) == 18 * 8 * 3 * 1024


def test_cpu_quota_parsing(tmp_path, monkeypatch):
    import builtins
    import bench
    real_open = builtins.open

    def fake(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            p = tmp_path / "cpu.max"
            p.write_text("1600000 100000\n")
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake)
    assert bench.cpu_quota() == 16.0


def _device_code_objects(lib_path):
    """the gfx950 code objects inside a HIP shared library: .hip_fatbin holds one uncompressed clang offload bundle per translation unit
    (magic, u64 entry count, then per entry u64 offset / u64 size / u64 triple length / triple)"""
    import struct
    import subprocess
    import tempfile
    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([objcopy, "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
        data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    pos = data.find(magic)
    while pos >= 0:
        n, = struct.unpack_from("<Q", data, pos + len(magic))
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(magic, pos + len(magic))
    return out


def test_shipped_library_has_no_packed_f32_instructions(tmp_path):
    """with v_pk_{fma,mul,add}_f32 in the instruction stream the fused conditional block produced rare wrong log-dets of whole
    16-row groups.  The remedy is a build flag clang calls "not a recognized feature", so the SHIPPED libjammy_hip.so itself is disassembled:
    every gfx950 code object in it must be free of packed-f32 arithmetic (a toolchain update that drops the flag fails here, on the CPU box)."""
    import re
    import subprocess
    lib = os.path.join(ROOT, "jammy_flows_amd", "libjammy_hip.so")
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(lib) and os.path.exists(objdump)):
        pytest.skip("library or llvm-objdump not available")
    cos = _device_code_objects(lib)
    csrc = os.path.join(ROOT, "jammy_flows_amd", "csrc")
    n_src = len([f for f in os.listdir(csrc) if f.endswith(".hip") and "__global__" in open(os.path.join(csrc, f)).read()])   # (plan.hip is host code only)
    assert len(cos) == n_src, "expected one gfx950 code object per .hip translation unit (%d), found %d" % (n_src, len(cos))
    pat = re.compile(rb"\bv_pk_(?:fma|mul|add)_f32\b")
    mfma = 0
    for i, co in enumerate(cos):
        p = tmp_path / ("dev%d.co" % i)
        p.write_bytes(co)
        asm = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(p)], stdout=subprocess.PIPE, check=True).stdout
        assert len(asm) > 1000
        hits = pat.findall(asm)
        assert not hits, "code object %d of libjammy_hip.so contains %d packed-f32 instructions" % (i, len(hits))
        mfma += len(re.findall(rb"\bv_mfma_f32_16x16x32_bf16\b", asm))
    assert mfma > 0, "the disassembly did not see the fused block's bf16 MFMAs: wrong code objects?"


def test_packed_gradient_rows_index_every_parameter_column_once():
    """host side of the one-launch block adjoint: cond_gf_packed_rows maps the natural column order of a g block's parameter row (offset,
    Householder vectors, means, log-widths, log-weights: gaussianization_flow.py:63-215) into the packed rows [layer][coordinate][slot] the
    kernel writes (csrc/jf_cond_regs.h) -- one packed column per parameter, the rest of the 144 per layer padding"""
    from jammy_flows_amd import _hip
    import jammy_flows_amd
    pdf = jammy_flows_amd.pdf("e4+e4", "gggg+gggg")
    for layers in (list(pdf.layer_list[1]), list(pdf.layer_list[1])[:2]):
        D = layers[0].dimension
        arr = _hip.gf_layer_array([l.c_struct() for l in layers])
        idx = _hip.cond_gf_packed_rows(arr, len(layers), D)
        n_params = sum(l.total_param_num for l in layers)
        assert len(idx) == n_params and len(set(idx)) == n_params
        assert min(idx) >= 0 and max(idx) < len(layers) * 4 * _hip.COND_GF_SLOTS
        col = 0
        for li, l in enumerate(layers):
            c = l.c_struct()
            base = li * 4 * _hip.COND_GF_SLOTS
            if c.model_offset:
                assert idx[col:col + D] == [base + d * _hip.COND_GF_SLOTS + 34 for d in range(D)]      # slot 34: offset
                col += D
            col += c.hh_iter * D
            first_mean = idx[col:col + D]
            assert first_mean == [base + d * _hip.COND_GF_SLOTS + 0 for d in range(D)]               # slot 0: mean of component 0
            col += 3 * c.num_kde * D
        assert col == n_params


def test_one_launch_adam_host_side():
    """jammy_flows_amd.optim.Adam without a device: torch's defaults and state layout, hyper-parameter checks, and no CPU fallback -- a host
    parameter with a gradient is refused loudly (the update itself is csrc/misc_kernels.hip: jf_adam_step, tests/test_gpu_grad.py)"""
    import torch
    from jammy_flows_amd import optim as jf_optim
    p = torch.nn.Parameter(torch.zeros(5))
    opt = jf_optim.Adam([p])
    ref = torch.optim.Adam([torch.nn.Parameter(torch.zeros(5))])
    for k in ("lr", "betas", "eps"):
        assert opt.param_groups[0][k] == ref.param_groups[0][k]
    opt.step()                                                   # no gradient anywhere: nothing to do, nothing raised
    assert len(opt.state) == 0
    p.grad = torch.ones(5)
    with pytest.raises(RuntimeError, match="HIP device"):
        opt.step()
    for bad in (dict(lr=-1.0), dict(betas=(1.0, 0.9)), dict(betas=(0.9, -0.1)), dict(eps=-1e-8)):
        with pytest.raises(ValueError):
            jf_optim.Adam([p], **bad)
    # a torch state_dict loads (same keys: step / exp_avg / exp_avg_sq)
    q = torch.nn.Parameter(torch.zeros(3))
    topt = torch.optim.Adam([q], lr=2e-3)
    q.grad = torch.ones(3)
    topt.step()
    mine = jf_optim.Adam([q], lr=1e-3)
    mine.load_state_dict(topt.state_dict())
    st = mine.state[q]
    assert set(st) == {"step", "exp_avg", "exp_avg_sq"} and int(st["step"]) == 1 and mine.param_groups[0]["lr"] == 2e-3


def test_kernel_caps_raise_loudly_without_a_gpu():
    """the caps the reference does not have (flow_options.py:38, spline_fns.py:45-186) are errors at construction / descriptor
    time, never silent truncation: 't' beyond 32 dimensions, 'r' / 'o' / nested splines / 'g' with rq_splines beyond 64 bins (16 until round 6), more than 4 nested
    f sub-layers"""
    import jammy_flows_amd as jf
    with pytest.raises(NotImplementedError, match="32 dimensions"):
        jf.pdf("e33", "t")
    p = jf.pdf("i1", "r", options_overwrite={"r": {"num_basis_functions": 65}})
    with pytest.raises(NotImplementedError, match="64 bins"):
        p.layer_list[0][0].c_struct()
    jf.pdf("i1", "r", options_overwrite={"r": {"num_basis_functions": 64}}).layer_list[0][0].c_struct()
    with pytest.raises(NotImplementedError, match="nested"):
        jf.pdf("s2", "f", options_overwrite={"f": {"add_vertical_rq_spline_flow": 1, "vertical_flow_defs": "rrrrr"}})
    with pytest.raises(NotImplementedError, match="64 bins|at most 64"):
        jf.pdf("e2", "g", options_overwrite={"g": {"nonlinear_stretch_type": "rq_splines", "num_kde": 65}})
    jf.pdf("e2", "g", options_overwrite={"g": {"nonlinear_stretch_type": "rq_splines", "num_kde": 64}})


def test_flat_pieces_gradient_equals_slicing():
    """autograd.FlatPieces (one graph node for all u / v / b blocks of an AmortizableMLP's parameter vector) gives the gradient of plain
    slicing, including pieces nobody used and a slice off the cut points"""
    from jammy_flows_amd import autograd as jfa
    torch.manual_seed(0)
    cuts = [0, 6, 6 + 8, 17, 30]
    w = [torch.randn(b - a, dtype=torch.float64) for a, b in zip(cuts[:-1], cuts[1:])]

    def loss(get):
        return (get(0, 6).view(2, 3) * w[0].view(2, 3)).sum() + (get(6, 14) ** 2 * w[1]).sum() + (get(17, 30) * w[3]).sum() + 3.0 * get(2, 9).sum()

    with torch.enable_grad():
        p1 = torch.randn(30, dtype=torch.float64, requires_grad=True)
        p2 = p1.detach().clone().requires_grad_(True)
        fp = jfa.FlatPieces(p1, cuts)
        loss(lambda a, b: fp[a:b]).backward()
        loss(lambda a, b: p2[a:b]).backward()
    assert torch.equal(p1.grad, p2.grad)
    assert float(p1.grad[14:17].abs().max()) == 0.0
