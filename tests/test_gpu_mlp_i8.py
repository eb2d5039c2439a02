"""jf_mlp2_i8_f64 (float64 Linear-tanh-Linear with the second product as int8 digit-slice products on the matrix cores, csrc/mlp_i8_kernels.hip)
against a float64 torch product: ragged row counts, every edge of the column / hidden / input ranges, weight rows of very different scale,
strided inputs and outputs; the dispatch in HipLinearStack (thresholds, the cached digit image follows weight updates, non-finite weights
take the exact kernel); the C3 golden fixture through the int8 path."""
import numpy as np
import pytest
import torch

from jammy_flows_amd import _hip
from jammy_flows_amd.main import default as jf_default

pytestmark = pytest.mark.gpu


def _case(B, K1, H, N, seed, row_scales=False):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.normal(size=(B, K1))).cuda()
    w1 = torch.from_numpy(rng.normal(size=(H, K1)) / np.sqrt(K1)).cuda()
    b1 = torch.from_numpy(rng.normal(size=(H,))).cuda()
    w2 = rng.normal(size=(N, H)) / np.sqrt(H)
    if row_scales:
        w2 *= np.exp(rng.uniform(-40, 40, size=(N, 1)))             # every row has its own power-of-two scale in the digit image
    w2 = torch.from_numpy(w2).cuda()
    b2 = torch.from_numpy(rng.normal(size=(N,))).cuda()
    return x, w1, b1, w2, b2


@pytest.mark.parametrize("slices,tol", [(6, 4e-12), (5, 5e-10)])
@pytest.mark.parametrize("B,K1,H,N", [(70003, 7, 128, 548), (1, 1, 4, 1), (129, 28, 128, 137), (4097, 3, 64, 16), (300, 7, 100, 600), (128, 16, 8, 49),
                                      (5000, 12, 128, 1224)])
def test_mlp2_i8_matches_float64_product(slices, tol, B, K1, H, N):
    x, w1, b1, w2, b2 = _case(B, K1, H, N, B + K1 + H + N)
    packed = _hip.mlp2_i8_pack(w2, b2, slices)
    out = _hip.mlp2_i8(x, w1, b1, packed, N, slices)
    h = torch.tanh(x @ w1.t() + b1)
    ref = h @ w2.t() + b2
    # scale of one output's error: the largest weight of its row times the number of terms (truncation of digits and levels is relative to that)
    bound = w2.abs().max(dim=1).values * H + 1e-300
    err = ((out - ref).abs() / bound).max().item()
    exact = _hip.mlp2(x, w1, b1, w2, b2)
    err_exact = ((exact - ref).abs() / bound).max().item()
    print("B %d K1 %d H %d N %d slices %d: max error / (H max|w|) %.2e (float64 MFMA kernel: %.2e)" % (B, K1, H, N, slices, err, err_exact))
    assert err < tol, err
    assert torch.isfinite(out).all()


def test_mlp2_i8_rows_of_very_different_scale():
    x, w1, b1, w2, b2 = _case(2000, 7, 128, 200, 11, row_scales=True)
    out = _hip.mlp2_i8(x, w1, b1, _hip.mlp2_i8_pack(w2, b2, 6), 200, 6)
    ref = torch.tanh(x @ w1.t() + b1) @ w2.t() + b2
    bound = 4e-12 * w2.abs().max(dim=1).values * 128 + 4.5e-16 * ref.abs()          # the product's own scale + the rounding of adding the bias
    assert ((out - ref).abs() <= bound).all(), ((out - ref).abs() / bound).max().item()


def test_mlp2_i8_extreme_weight_magnitudes():
    """every finite double can be cut into digits: rows near the overflow threshold and denormal rows"""
    x, w1, b1, w2, b2 = _case(500, 7, 128, 48, 13)
    w2 = w2.clone()
    w2[0] *= 1e300
    w2[1] *= 1e-300
    w2[2] *= 1e-320                                                 # denormal weights: the row is zero to every digit, like its exact product to 1e-300
    b2 = torch.zeros_like(b2)
    out = _hip.mlp2_i8(x, w1, b1, _hip.mlp2_i8_pack(w2, b2, 6), 48, 6)
    ref = torch.tanh(x @ w1.t() + b1) @ w2.t()
    assert torch.isfinite(out).all()
    for r in (0, 1, 3):
        assert ((out[:, r] - ref[:, r]).abs() <= 4e-12 * w2[r].abs().max() * 128).all(), r
    assert (out[:, 2].abs() <= 1e-300).all()


def test_mlp2_i8_strided_input_and_output_and_zero_rows():
    x, w1, b1, w2, b2 = _case(777, 7, 128, 137, 3)
    wide_x = torch.zeros((777, 19), dtype=torch.float64, device="cuda")
    wide_x[:, 2:9] = x
    wide_out = torch.full((777, 300), -7.0, dtype=torch.float64, device="cuda")
    packed = _hip.mlp2_i8_pack(w2, b2, 6)
    _hip.mlp2_i8(wide_x[:, 2:9], w1, b1, packed, 137, 6, out=wide_out[:, 5:142])
    dense = _hip.mlp2_i8(x, w1, b1, packed, 137, 6)
    assert torch.equal(wide_out[:, 5:142], dense)
    assert (wide_out[:, :5] == -7.0).all() and (wide_out[:, 142:] == -7.0).all()      # nothing outside the requested columns is touched
    empty = _hip.mlp2_i8(x[:0], w1, b1, packed, 137, 6)
    assert empty.shape == (0, 137)
    with pytest.raises(ValueError):
        _hip.mlp2_i8(x, w1, b1, packed, 90, 6)                      # an image of another size (the chunk count is what can be checked)


def test_mlp2_i8_is_bit_deterministic_at_full_size():
    x, w1, b1, w2, b2 = _case(1 << 20, 7, 128, 548, 5)
    packed = _hip.mlp2_i8_pack(w2, b2, 6)
    first = _hip.mlp2_i8(x, w1, b1, packed, 548, 6)
    small = _hip.mlp2_i8(x[:4096], w1, b1, packed, 548, 6)
    assert torch.equal(first[:4096], small)
    for _ in range(5):
        assert torch.equal(_hip.mlp2_i8(x, w1, b1, packed, 548, 6), first)
    ref = torch.tanh(x[-1000:] @ w1.t() + b1) @ w2.t() + b2
    assert (first[-1000:] - ref).abs().max().item() < 1e-11


def _stack(K1, H, N, seed):
    torch.manual_seed(seed)
    st = jf_default.HipLinearStack(torch.nn.Linear(K1, H), torch.nn.Tanh(), torch.nn.Linear(H, N)).double().cuda()
    return st


def test_linear_stack_dispatch_and_image_cache():
    # error bound of the digit-slice product in use (default i8x5 since round 5: operands kept to 2^-34; i8x6: 2^-41)
    tol = 1e-11 if jf_default.MLP_MATRIX_ARITHMETIC_F64[0] == "i8x6" else 2e-9
    st = _stack(7, 128, 548, 0)
    x = torch.randn(5000, 7, dtype=torch.float64, device="cuda")
    small = torch.randn(100, 7, dtype=torch.float64, device="cuda")

    def ran(inp):
        timer = _hip.KernelTimer()
        with timer, torch.no_grad():
            out = st(inp)
        return out, {k[0] for k in timer.summary()}

    out, names = ran(x)
    assert "jf_mlp2_i8_f64" in names and "jf_mlp2_i8_pack_f64" in names, names
    ref = torch.tanh(x @ st[0].weight.t() + st[0].bias) @ st[2].weight.t() + st[2].bias
    assert (out - ref).abs().max().item() < tol
    out2, names = ran(x)
    assert names == {"jf_mlp2_i8_f64"} and torch.equal(out, out2)   # image reused
    _, names = ran(small)
    assert names == {"jf_mlp2_f64"}, names                          # small batches: the exact kernel
    with torch.no_grad():
        st[2].weight.mul_(1.5)                                      # in-place update (an optimiser step): the image follows
    out3, names = ran(x)
    assert "jf_mlp2_i8_pack_f64" in names
    ref = torch.tanh(x @ st[0].weight.t() + st[0].bias) @ st[2].weight.t() + st[2].bias
    assert (out3 - ref).abs().max().item() < tol
    prev = jf_default.MLP_MATRIX_ARITHMETIC_F64[0]
    try:
        jf_default.MLP_MATRIX_ARITHMETIC_F64[0] = "f64"
        out4, names = ran(x)
        assert names == {"jf_mlp2_f64"}
        assert (out4 - out3).abs().max().item() < tol
    finally:
        jf_default.MLP_MATRIX_ARITHMETIC_F64[0] = prev
    with torch.no_grad():
        st[2].weight[3, 5] = float("inf")                           # cannot be cut into digits: the exact kernel propagates it
    out5, names = ran(x)
    assert names == {"jf_mlp2_f64"}, names
    assert not torch.isfinite(out5[:, 3]).any() and torch.isfinite(out5[:, 4]).all()


def test_nan_rows_propagate_through_the_int8_path():
    """(int)rint(NaN) is 0, so a NaN hidden activation used to become digit 0 and the row came out as b2 -- finite, no status
    flag, and only above the 4096-row switch to this kernel.  A NaN input row, a NaN in W1 and an inf - inf in the first layer must give NaN
    outputs for exactly the affected rows, as jf_mlp2_f64 and nn.Linear do; every other row is untouched."""
    st = _stack(7, 128, 548, 3)
    x = torch.randn(6000, 7, dtype=torch.float64, device="cuda")
    with torch.no_grad():
        clean = st(x)
        xb = x.clone()
        xb[17, 2] = float("nan")
        xb[4100, 0] = float("inf")                                   # inf * w + (-inf) * w' in the first layer: NaN pre-activations
        xb[4100, 1] = float("-inf")
        timer = _hip.KernelTimer()
        with timer:
            out = st(xb)
        assert any(k[0] == "jf_mlp2_i8_f64" for k in timer.summary())
        bad = torch.zeros(6000, dtype=torch.bool, device="cuda")
        bad[17] = True
        bad[4100] = True
        assert torch.isnan(out[bad]).all(), "rows with a NaN hidden activation must be NaN in every column"
        assert torch.equal(out[~bad], clean[~bad])
        ref = torch.tanh(xb @ st[0].weight.t() + st[0].bias) @ st[2].weight.t() + st[2].bias
        assert torch.equal(torch.isnan(ref).all(dim=1), bad)
        # an infinite (not NaN) pre-activation is a legitimate h = +-1
        xi = x.clone()
        xi[5, 3] = float("inf")
        oi = st(xi)
        ri = torch.tanh(xi @ st[0].weight.t() + st[0].bias) @ st[2].weight.t() + st[2].bias
        assert torch.isfinite(oi[5]).all() and (oi[5] - ri[5]).abs().max().item() < 1e-10


def test_c3_golden_fixture_through_the_int8_path():
    from helpers import ALL_FIXTURES, build_product, max_rel, to_dev
    fx = [f for f in ALL_FIXTURES if f.name == "c3_e4s2e4"][0]
    pdf = build_product(fx, torch.float64)
    x, cond = to_dev(fx["x"], torch.float64), to_dev(fx.get("cond"), torch.float64)
    prev = jf_default.MLP_I8_MIN_ROWS[0]
    try:
        jf_default.MLP_I8_MIN_ROWS[0] = 1
        timer = _hip.KernelTimer()
        with timer:
            logp, logp_base, base = pdf(x, conditional_input=cond, force_embedding_coordinates=fx.meta["embedding"])
        assert any(k[0] in ("jf_mlp2_i8_f64", "jf_mlp2_i8_seg_f64") for k in timer.summary()), sorted(timer.summary())
        assert max_rel(logp, fx["logp"]) < 1e-7 and max_rel(base, fx["base"]) < 1e-6
        z = to_dev(fx["z"], torch.float64)
        xs, _, slogp, _ = pdf._obtain_sample(conditional_input=cond, predefined_target_input=z, force_embedding_coordinates=fx.meta["embedding"])
        assert max_rel(xs, fx["sample_x"]) < 1e-6 and max_rel(slogp, fx["sample_logp"]) < 1e-6
    finally:
        jf_default.MLP_I8_MIN_ROWS[0] = prev
