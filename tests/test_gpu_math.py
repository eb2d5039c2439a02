"""The device math functions of the flow kernels (csrc/jf_math.h) against torch's float64 functions: exp_fast (ulp error
over the whole range, special values, gradual underflow), log_fast, the absolute-accuracy tanh_fast of the hidden layers, rcp."""
import numpy as np
import pytest
import torch

from jammy_flows_amd import _hip

pytestmark = pytest.mark.gpu


def _ulps(got, ref):
    """error in units of the last place of ref (float64)"""
    spacing = torch.from_numpy(np.spacing(np.abs(ref.cpu().numpy()))).to(ref.device)
    return ((got - ref).abs() / spacing)


def test_exp_fast_float64_ulp_error_and_special_values():
    fn = _hip.MATH_EXP_FAST
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.cat([torch.empty(1 << 20, dtype=torch.float64).uniform_(-745.0, 709.0, generator=g),
                   torch.empty(1 << 20, dtype=torch.float64).uniform_(-40.0, 40.0, generator=g),
                   torch.empty(1 << 18, dtype=torch.float64).uniform_(-1e-3, 1e-3, generator=g),
                   torch.linspace(-0.0108304246962 * 3, 0.0108304246962 * 3, 4097, dtype=torch.float64)]).cuda()      # around the reduction's break points
    got = _hip.device_math(x, fn)
    ref = torch.exp(x)
    normal = ref > 2.3e-308
    u = _ulps(got[normal], ref[normal])
    print("exp (fn %d) float64: max %.2f ulp, mean %.3f ulp over %d normal-range results" % (fn, u.max().item(), u.mean().item(), int(normal.sum())))
    assert u.max().item() < 2.0
    # gradual underflow: absolute error below one denormal step
    sub = ~normal
    assert ((got[sub] - ref[sub]).abs() <= 1.5 * 4.94e-324).all()
    special = torch.tensor([float("-inf"), float("inf"), float("nan"), 0.0, -0.0, 709.78, 710.0, 800.0, -745.2, -746.0, -1e300, 1e300], dtype=torch.float64).cuda()
    gs, rs = _hip.device_math(special, fn), torch.exp(special)
    assert torch.equal(torch.isnan(gs), torch.isnan(rs))
    ok = ~torch.isnan(rs)
    fin = ok & torch.isfinite(rs)
    assert torch.equal(torch.isinf(gs[ok]), torch.isinf(rs[ok]))
    assert ((gs[fin] - rs[fin]).abs() <= 4e-16 * rs[fin].abs() + 1e-323).all()
    assert gs[3].item() == 1.0 and gs[4].item() == 1.0


def test_tanh_fast_float64_absolute_accuracy():
    g = torch.Generator(device="cpu").manual_seed(2)
    x = torch.cat([torch.empty(1 << 20, dtype=torch.float64).uniform_(-25.0, 25.0, generator=g),
                   torch.empty(1 << 18, dtype=torch.float64).uniform_(-1e-6, 1e-6, generator=g),
                   torch.tensor([0.0, -0.0, float("inf"), float("-inf"), 1e300, -1e300], dtype=torch.float64)]).cuda()
    got = _hip.device_math(x, _hip.MATH_TANH_FAST)
    err = (got - torch.tanh(x)).abs().max().item()
    print("tanh_fast float64: max absolute error %.2e" % err)
    assert err < 4e-16
    assert torch.isnan(_hip.device_math(torch.tensor([float("nan")], dtype=torch.float64).cuda(), _hip.MATH_TANH_FAST)).all()
    assert (got.abs() <= 1.0).all() and (torch.sign(got) == torch.sign(x)).all()


def test_log_fast_and_rcp_float64():
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.exp(torch.empty(1 << 20, dtype=torch.float64).uniform_(-700.0, 700.0, generator=g)).cuda()
    u = _ulps(_hip.device_math(x, _hip.MATH_LOG_FAST), torch.log(x))
    print("log_fast float64: max %.2f ulp" % u.max().item())
    assert u.max().item() < 2.0
    u = _ulps(_hip.device_math(x, _hip.MATH_RCP), 1.0 / x)
    print("rcp float64: max %.2f ulp" % u.max().item())
    assert u.max().item() < 2.0


def test_float32_math_functions_are_the_hardware_ones():
    x = torch.empty(1 << 16, dtype=torch.float32).uniform_(-80.0, 80.0).cuda()
    got = _hip.device_math(x, _hip.MATH_EXP_FAST).double()
    ref = torch.exp(x.double())
    assert ((got - ref).abs() / ref).max().item() < 5e-6          # v_exp_f32 after the multiplication by log2(e): ~2e-6 at |x| = 80
