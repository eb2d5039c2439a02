"""jf_linear_split_f32 (split-bf16 MFMA dense layer, float32-equivalent accuracy) against a float64 product: both loop shapes (wide: N > 128,
deep: N <= 128), ragged row counts, K not a multiple of 32, a transposed weight view, with and without bias."""
import numpy as np
import pytest
import torch

from jammy_flows_amd import _hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,K,N,transposed,bias", [(70001, 128, 548, False, True), (5000, 548, 128, True, False), (257, 36, 20, False, True),
                                                   (1, 4, 4, False, False), (129, 128, 132, True, True), (4097, 548, 128, False, True)])
def test_linear_split_matches_float64_product(B, K, N, transposed, bias):
    rng = np.random.default_rng(B + K + N)
    x = torch.from_numpy(rng.normal(size=(B, K)) * np.exp(rng.normal(size=(B, 1)))).to(device="cuda", dtype=torch.float32)
    w_full = torch.from_numpy(rng.normal(size=(K, N) if transposed else (N, K)) / np.sqrt(K)).to(device="cuda", dtype=torch.float32)
    w = w_full.t() if transposed else w_full
    b = torch.from_numpy(rng.normal(size=(N,))).to(device="cuda", dtype=torch.float32) if bias else None
    assert _hip.linear_split_ok(x, w, b)
    out = _hip.linear_split(x, w, b)
    ref = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    bound = x.double().abs() @ w.double().abs().t() + (b.double().abs() if bias else 0.0) + 1e-30     # sum |x||w| + |b|: the scale of the rounding of a K-term float32 dot product
    rel = (out.double() - ref).abs() / bound
    err, rms = rel.max().item(), rel.pow(2).mean().sqrt().item()
    exact = x @ w.t() + (b if bias else 0.0)                       # the library's float32 product, for scale
    rel_lib = (exact.double() - ref).abs() / bound
    print("B %d K %d N %d: error / sum|x||w|: max %.2e rms %.2e (library float32 GEMM: max %.2e rms %.2e)" % (
        B, K, N, err, rms, rel_lib.max().item(), rel_lib.pow(2).mean().sqrt().item()))
    # 6 piece products x K / 32 k-steps, each one float32 accumulation (<= 2^-24 of the running sum): worst case 6 K / 32 * 6e-8
    assert err < 6 * ((K + 31) // 32) * 6.0e-8 + 1e-7
    assert err < 4 * rel_lib.max().item() + 1e-7 and rms < 1.0e-7


def test_linear_split_rejects_unaligned_shapes():
    x = torch.zeros((8, 6), device="cuda")
    w = torch.zeros((8, 6), device="cuda")
    assert not _hip.linear_split_ok(x, w)
    with pytest.raises(ValueError):
        _hip.linear_split(x, w)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("B,K,N,act", [(5000, 8, 1224, 0), (4099, 1224, 8, 0), (300, 8, 128, 1), (300, 128, 8, 0), (7, 3, 50, 1), (1001, 16, 40, 0),
                                       (130, 548, 10, 1), (1, 300, 1, 0), (70000, 8, 36, 0)])
def test_linear_skinny_shapes(dtype, B, K, N, act):
    """jf_linear with one tiny dimension (the rank-8 stages of AmortizableMLP and their backward: streaming kernels instead of MFMA tiles)"""
    rng = np.random.default_rng(B + K + N)
    x = torch.from_numpy(rng.normal(size=(B, K))).to(device="cuda", dtype=dtype)
    w = torch.from_numpy(rng.normal(size=(N, K)) / np.sqrt(K)).to(device="cuda", dtype=dtype)
    b = torch.from_numpy(rng.normal(size=(N,))).to(device="cuda", dtype=dtype)
    out = _hip.linear(x, w, b, act)
    ref = x.double() @ w.double().t() + b.double()
    if act:
        ref = torch.tanh(ref)
    err = (out.double() - ref).abs().max().item()
    assert err < (1e-12 if dtype == torch.float64 else 3e-6) * (1 + K ** 0.5), err


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("B,K1,H,N", [(1000, 4, 128, 10), (4099, 7, 64, 3), (300, 32, 128, 16), (1, 1, 4, 1), (70001, 4, 128, 10),
                                      (70001, 4, 128, 46), (999, 8, 64, 64), (513, 2, 128, 33), (200, 1, 100, 17), (3000, 4, 128, 48)])
def test_mlp2_small_backward_vs_autograd(dtype, B, K1, H, N):
    """jf_mlp2_small_bwd (whole backward of a narrow Linear-tanh-Linear head in one launch) against torch autograd in float64; the wide
    variants (17 .. 64 outputs behind <= 8 inputs: the 4 -> 128 -> 46 head of an 'f' layer with spline flows) included"""
    if not _hip.mlp2_small_shape_ok(K1, H, N, 8 if dtype == torch.float64 else 4):
        assert dtype == torch.float64 and N > 48
        with pytest.raises(RuntimeError, match="unsupported"):
            _hip.mlp2_small_bwd(*(torch.zeros(s, device="cuda", dtype=dtype) for s in ((B, K1), (H, K1), (H,), (N, H), (B, N))))
        return
    rng = np.random.default_rng(B + K1 + N)
    mk = lambda *s: torch.from_numpy(rng.normal(size=s)).to(device="cuda", dtype=torch.float64)
    x, w1, b1, w2, b2, g = mk(B, K1), mk(H, K1) / np.sqrt(K1), mk(H), mk(N, H) / np.sqrt(H), mk(N), mk(B, N)
    ps = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    with torch.enable_grad():
        out = torch.tanh(x @ ps[0].t() + ps[1]) @ ps[2].t() + ps[3]
        out.backward(g)
    got = _hip.mlp2_small_bwd(x.to(dtype), w1.to(dtype), b1.to(dtype), w2.to(dtype), g.to(dtype))
    tol = 1e-11 if dtype == torch.float64 else 3e-5
    for name, a, p in zip(("w1", "b1", "w2", "b2"), got, ps):
        err = (a.double() - p.grad).abs().max().item() / max(p.grad.abs().max().item(), 1e-30)
        assert err < tol * (1 + B ** 0.5 / 30), (name, err)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("B,K1,H", [(1000, 7, 128), (4099, 4, 64), (70001, 7, 128), (3, 32, 8)])
def test_mlp_hidden_backward_vs_autograd(dtype, B, K1, H):
    """jf_mlp_hidden_bwd (tanh derivative + first-layer weight / bias gradient in one launch) against torch autograd in float64"""
    rng = np.random.default_rng(B + K1 + H)
    mk = lambda *s: torch.from_numpy(rng.normal(size=s)).to(device="cuda", dtype=torch.float64)
    x, w1, b1, gh = mk(B, K1), mk(H, K1) / np.sqrt(K1), mk(H), mk(B, H)
    ps = [t.clone().requires_grad_(True) for t in (w1, b1)]
    with torch.enable_grad():
        torch.tanh(x @ ps[0].t() + ps[1]).backward(gh)
    got = _hip.mlp_hidden_bwd(x.to(dtype), w1.to(dtype), b1.to(dtype), gh.to(dtype))
    tol = 1e-11 if dtype == torch.float64 else 3e-5
    for name, a, p in zip(("w1", "b1"), got, ps):
        err = (a.double() - p.grad).abs().max().item() / max(p.grad.abs().max().item(), 1e-30)
        assert err < tol * (1 + B ** 0.5 / 30), (name, err)


def test_split_kernels_repeat_bit_identically_at_full_size():
    """2^18 rows, three launches each: the split-bf16 products and the weight gradient have no atomics and no launch-order dependence, so
    repeated launches must agree bit for bit (the check that exposed the packed-f32 fault of the fused forward block)"""
    B = 1 << 18
    g_ = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, 128, device="cuda", generator=g_)
    w = torch.randn(548, 128, device="cuda", generator=g_) / 11.0
    b = torch.randn(548, device="cuda", generator=g_)
    gp = torch.randn(B, 548, device="cuda", generator=g_)
    first = (_hip.linear_split(x, w, b), _hip.linear_split(gp, w.t()), *_hip.linear_wgrad(gp, x))
    for _ in range(2):
        again = (_hip.linear_split(x, w, b), _hip.linear_split(gp, w.t()), *_hip.linear_wgrad(gp, x))
        for a, c in zip(first, again):
            assert torch.equal(a, c)
    assert all(torch.isfinite(t).all() for t in first)


@pytest.mark.parametrize("B,K1,H,N", [(70001, 4, 128, 46), (1000, 1, 128, 8), (333, 4, 128, 10), (4099, 3, 100, 17), (1, 2, 128, 32), (5000, 4, 128, 64),
                                      (777, 4, 64, 47), (129, 1, 8, 1), (2048, 4, 128, 16), (2048, 4, 128, 68), (900, 7, 128, 46)])
def test_mlp2_float32_vs_float64_product(B, K1, H, N):
    """jf_mlp2_f32 against a float64 evaluation: the narrow kernel (<= 4 inputs, <= 64 outputs in one .. four column tiles of 16,
    mlp_narrow_kernels.hip: vector-unit first layer, split-f16 second layer), and the exact-float32 MFMA kernel for the shapes beyond it.
    Strided output rows (the wrapper pads rows to 128-byte lines), rows not a multiple of 16, widths that are not multiples of 4 / 2."""
    rng = np.random.default_rng(B + 7 * K1 + H + 13 * N)
    dev = lambda a: torch.from_numpy(a).to(device="cuda", dtype=torch.float32)
    x = dev(rng.normal(size=(B, K1)) * 1.5)
    w1, b1 = dev(rng.normal(size=(H, K1)) / np.sqrt(K1)), dev(rng.normal(size=(H,)) * 0.3)
    w2, b2 = dev(rng.normal(size=(N, H)) / np.sqrt(H)), dev(rng.normal(size=(N,)))
    out = _hip.mlp2(x, w1, b1, w2, b2)
    assert out.shape == (B, N) and torch.isfinite(out).all()
    ref = torch.tanh(x.double() @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    bound = w2.double().abs().sum(dim=1) + b2.double().abs()           # |h| <= 1: the scale of the rounding of the H-term product
    err = ((out.double() - ref).abs() / bound).max().item()
    print("B %d K1 %d H %d N %d: max error / (sum|w2| + |b2|) %.2e" % (B, K1, H, N, err))
    assert err < 2e-6, err
    # an odd output width into an unpadded buffer (scalar stores), and a caller's buffer whose neighbours must stay untouched
    if N <= 64:
        buf = torch.full((B, N + 3), 7.0, dtype=torch.float32, device="cuda")
        _hip.mlp2(x, w1, b1, w2, b2, out=buf[:, 1:N + 1])
        assert torch.equal(buf[:, 1:N + 1], out) and (buf[:, 0] == 7.0).all() and (buf[:, N + 1:] == 7.0).all()
