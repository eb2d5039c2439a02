"""Step plans (include/jammy_hip.h "step plans", csrc/plan.hip, pdf.planned_forward): the whole log-prob step of a pdf -- the loop over
sub-manifolds and layers the reference runs in Python on every call (jammy_flows/main/default.py:879-1117) -- recorded once and re-issued
from C by one call.  A replay must be bit-identical to the eager step, on inputs it was not recorded with, for every fixture."""
import numpy as np
import pytest
import torch

import helpers
from helpers import ALL_FIXTURES, build_product, max_rel, to_dev

pytestmark = pytest.mark.gpu

SUPPORTED = [fx for fx in ALL_FIXTURES if helpers.product_supports(fx)]
IDS = [fx.name for fx in SUPPORTED]


def same(a, b):
    return bool(((a == b) | (a.isnan() & b.isnan())).all())


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("fx", SUPPORTED, ids=IDS)
def test_plan_replay_equals_eager_forward(fx, dtype):
    from jammy_flows_amd.main.default import PlanNotApplicable
    if dtype == torch.float32 and any(part[0] == "v" for part in fx.flow_defs.split("+")):
        pytest.skip("'v' is float64 only (exponential_map_s2.py:450)")
    pdf = build_product(fx, dtype)
    pdf.check_status = False                       # adversarial fixture rows (interval ends, poles) set status words in float32; parity is the subject here
    x = to_dev(fx["x"], dtype)
    cond = to_dev(fx.get("cond"), dtype)
    kw = dict(force_embedding_coordinates=bool(fx.meta["embedding"]))
    try:
        pdf(x, conditional_input=cond, **kw)
    except RuntimeError as e:                      # e.g. add_skewness in float32: the reference asserts float64, the library returns JF_ERR_UNSUPPORTED
        pytest.skip("the eager step itself is refused: %s" % e)
    try:
        pf = pdf.planned_forward(x, conditional_input=cond, **kw)
    except PlanNotApplicable as e:
        pytest.skip("not plannable: %s" % e)
    assert pf.plan.n_ops >= 1                      # (the status words live in host memory: no copy-back op, a one-block pdf is one launch)
    # other rows than the recorded ones, in other buffers
    perm = torch.randperm(x.shape[0], device=x.device)
    x2 = x[perm].clone()
    c2 = None if cond is None else cond[perm].clone()
    x2_before = x2.clone()
    want = pdf(x2, conditional_input=c2, **kw)
    got = pf(x2, c2)
    assert torch.equal(x2, x2_before), "inputs must not be modified (tests/test_general.py:519)"
    for g, w in zip(got, want):
        assert g.data_ptr() != w.data_ptr() and same(g, w)
    # and against the golden vectors of the reference itself (float32 has rows outside its domain in the adversarial tail: test_gpu_parity.py)
    if dtype == torch.float64:
        logp = pf(x, cond)[0]
        fin = np.isfinite(fx["logp"])
        assert max_rel(logp[torch.from_numpy(fin).to(logp.device)], fx["logp"][fin]) < 1e-6


def test_forward_uses_plans_when_enabled_and_follows_parameter_updates():
    fx = [f for f in SUPPORTED if f.name == "c3_e4s2e4"][0]
    pdf = build_product(fx, torch.float32)
    x = to_dev(np.tile(fx["x"], (40, 1)), torch.float32)
    eager = pdf(x)
    pdf.use_step_plans = True
    a = pdf(x)
    assert len(pdf._step_plans) == 1 and all(pdf._step_plans.values())
    for g, w in zip(a, eager):
        assert same(g, w)
    plan = next(iter(pdf._step_plans.values()))
    n_records = plan.plan.handle
    # an in-place parameter update (what an optimizer step does): the next call must see the new weights
    with torch.no_grad():
        for p in pdf.parameters():
            p.mul_(1.01)
    b = pdf(x)
    assert plan.plan.handle != n_records, "the plan must have been recorded again"
    pdf.use_step_plans = False
    want = pdf(x)
    for g, w in zip(b, want):
        assert same(g, w)
    assert not same(b[0], a[0])
    # a second input signature gets its own plan
    pdf.use_step_plans = True
    pdf(x[:100])
    assert len(pdf._step_plans) == 2


def test_plan_reports_status_like_the_eager_path():
    """spline input outside its interval: the eager path raises (spline_fns.py:57-59); a plan raises at the flush / next call"""
    fx = [f for f in SUPPORTED if f.name == "c4_i1s1_ro"][0]
    pdf = build_product(fx, torch.float64)
    x = to_dev(fx["x"], torch.float64)[:64].clone()
    x[:, 0] = x[:, 0].clamp(0.01, 0.99)
    pdf.check_status = "deferred"
    pf = pdf.planned_forward(x)
    pf(x)
    pf.flush()
    bad = x.clone()
    bad[3, 0] = 1.5
    pf(bad)
    with pytest.raises(Exception, match="outside boundaries|nonfinite values"):   # (x = 1.5 is clamped by the r layer, the interval chart then has no finite image)
        pf.flush()
    pf(x)                      # the words were cleared with the report
    pf.flush()


def test_kernel_timer_sees_the_kernels_of_a_replayed_plan():
    from jammy_flows_amd import _hip
    fx = [f for f in SUPPORTED if f.name == "c3_e4s2e4"][0]
    pdf = build_product(fx, torch.float32)
    pdf.check_status = False
    x = to_dev(np.tile(fx["x"], (400, 1)), torch.float32)
    pf = pdf.planned_forward(x)
    t = _hip.KernelTimer()
    with t:
        for _ in range(5):
            pf(x)
    table = t.summary()
    names = {k[0] for k in table}
    # (the broadcast g chain and the `f` block leave as one launch: csrc/merged_kernels.hip)
    assert "jf_cond_gf_chain_split3_f32" in names and "jf_merge_end" in names, names
    assert all(v["launches"] == 5 and v["mean_ms"] > 0 for v in table.values()), table


def test_kernel_timer_sampling_scales_to_all_replays():
    """KernelTimer(plan_every=4) (bench.py): events on every 4th replay only; launches / total_ms are scaled to all replays, mean_ms is measured"""
    from jammy_flows_amd import _hip
    fx = [f for f in SUPPORTED if f.name == "c3_e4s2e4"][0]
    pdf = build_product(fx, torch.float32)
    pdf.check_status = False
    x = to_dev(np.tile(fx["x"], (400, 1)), torch.float32)
    pf = pdf.planned_forward(x)
    t = _hip.KernelTimer(plan_every=4)
    with t:
        for _ in range(12):
            pf(x)
    table = t.summary()
    assert table and all(abs(v["launches"] - 12) < 1e-9 and v["mean_ms"] > 0 and abs(v["total_ms"] - 12 * v["mean_ms"]) < 1e-9 for v in table.values()), table


def test_plan_sees_a_replaced_parameter_object():
    """the plan's parameter key walks a kept tensor list (the nn.Module walk was half of a shard step's host time) and looks at the module again
    every 256th call: a Parameter OBJECT that was replaced is seen within that many steps; invalidate_packed_caches() is immediate"""
    fx = [f for f in SUPPORTED if f.name == "c2_e4_gggg"][0]
    pdf = build_product(fx, torch.float64)
    x = to_dev(fx["x"], torch.float64)
    pdf.use_step_plans = True
    a = pdf(x)[0].clone()
    name, old_p = next(iter(pdf.named_parameters()))
    owner = pdf
    for part in name.split(".")[:-1]:
        owner = getattr(owner, part)
    setattr(owner, name.split(".")[-1], torch.nn.Parameter(old_p.detach() * 1.02))
    for _ in range(260):
        b = pdf(x)[0]
    pdf.use_step_plans = False
    want = pdf(x)[0]
    assert same(b, want) and not same(b, a)
    # ... and at once after invalidate_packed_caches()
    pdf.use_step_plans = True
    setattr(owner, name.split(".")[-1], torch.nn.Parameter(old_p.detach() * 0.97))
    pdf.invalidate_packed_caches()
    c = pdf(x)[0]
    pdf.use_step_plans = False
    assert same(c, pdf(x)[0]) and not same(c, b)


def test_mlp_inputs_read_in_place_equal_the_materialised_rows():
    """_hip.SegInput: cat[conditional_input, embed(x_0), ...] (main/default.py:946-962) read by the consumer kernel from the segments themselves
    (csrc/jf_cond_in.h).  float32: angles embedded with the hardware sine / cosine -> within 1e-4 of the materialised path on log p (bar 1e-2);
    plain column ranges in float64: identical."""
    from jammy_flows_amd import _hip
    fx = [f for f in SUPPORTED if f.name == "c3_e4s2e4"][0]
    pdf = build_product(fx, torch.float32)
    pdf.check_status = False
    x = to_dev(np.tile(fx["x"], (30, 1)), torch.float32)
    t = _hip.KernelTimer()
    with t:
        a = pdf(x)[0]
    assert any(k[0] == "jf_cond_gf_chain_split3_f32" for k in t.summary()) and not any(k[0].startswith("jf_conditioning_rows") for k in t.summary())
    saved = _hip.SegInput.in_place_ok
    try:
        _hip.SegInput.in_place_ok = property(lambda self: False)
        t = _hip.KernelTimer()
        with t:
            b = pdf(x)[0]
        assert any(k[0].startswith("jf_conditioning_rows") for k in t.summary())
    finally:
        _hip.SegInput.in_place_ok = saved
    fin = torch.isfinite(a) & torch.isfinite(b)
    assert bool((torch.isfinite(a) == torch.isfinite(b)).all())
    assert float(((a - b).abs() / (1 + b.abs()))[fin].max()) < 2e-6       # (one float32 ulp of the largest |log p| is 1.2e-4)
    # float64, two plain column ranges, the int8-slice MLP
    g = torch.Generator(device="cuda").manual_seed(3)
    B = 5000
    u, v = torch.randn(B, 9, generator=g, device="cuda", dtype=torch.float64), torch.randn(B, 6, generator=g, device="cuda", dtype=torch.float64)
    w1, b1 = torch.randn(128, 7, generator=g, device="cuda", dtype=torch.float64) * 0.3, torch.randn(128, generator=g, device="cuda", dtype=torch.float64)
    w2, b2 = torch.randn(200, 128, generator=g, device="cuda", dtype=torch.float64), torch.randn(200, generator=g, device="cuda", dtype=torch.float64)
    img = _hip.mlp2_i8_pack(w2, b2)
    seg = _hip.SegInput([(u[:, 2:5], 0), (v[:, 1:5], 0)], B, torch.float64, u.device)
    assert seg.in_place_ok
    got = _hip.mlp2_i8(seg, w1, b1, img, 200)
    want = _hip.mlp2_i8(torch.cat([u[:, 2:5], v[:, 1:5]], dim=1), w1, b1, img, 200)
    assert torch.equal(got, want)
