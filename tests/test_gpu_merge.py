"""Round 5 additions to the small-batch log-prob step (DESIGN "small shards"):
  * two side blocks in ONE launch (csrc/merged_kernels.hip; main/default.py:946-962 is why the blocks are independent),
  * the broadcast g chain with one lane per (row, coordinate) (csrc/jf_gfb.h) -- bit-identical to the lane = row kernel, so the kernel choice may
    follow the batch size without a row's result depending on the batch it sits in,
  * a recorded step's status words in pinned host memory (no copy-back launch),
  * a step plan whose slots cannot be declared (x and conditional_input views of one tensor, empty batch) falls back to eager."""
import numpy as np
import pytest
import torch

import helpers
from helpers import ALL_FIXTURES, build_product, to_dev

pytestmark = pytest.mark.gpu
BY_NAME = {fx.name: fx for fx in ALL_FIXTURES}


def same(a, b):
    return bool(((a == b) | (a.isnan() & b.isnan())).all())


def rows(fx, n, dtype, seed=0):
    """n rows drawn (with repetition, shuffled) from the fixture's own targets: valid points of every sub-manifold"""
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, fx["x"].shape[0], size=n)
    return to_dev(fx["x"][idx], dtype), to_dev(None if fx.get("cond") is None else fx["cond"][idx], dtype)


@pytest.mark.parametrize("n", [1, 63, 255, 256, 257, 4173, (1 << 15) + 129])
def test_side_blocks_in_one_launch_are_bit_identical(n):
    from jammy_flows_amd import _hip
    fx = BY_NAME["c3_e4s2e4"]
    pdf = build_product(fx, torch.float32)
    pdf.check_status = False
    x, c = rows(fx, n, torch.float32)
    pdf.merge_max_rows = 0
    ref = pdf(x, conditional_input=c)
    pf0 = pdf.planned_forward(x, conditional_input=c)
    pdf.merge_max_rows = 1 << 40
    pf1 = pdf.planned_forward(x, conditional_input=c)
    assert pf1.plan.n_ops == pf0.plan.n_ops - 1, (pf0.plan.calls, pf1.plan.calls)
    assert any(name == "jf_merge_end" and b > a for name, _, a, b in pf1.plan.calls)
    for rep in range(3):
        for fn in (pdf, pf1):
            got = fn(x, conditional_input=c)
            assert all(same(g, w) for g, w in zip(got, ref)), (n, rep)
    t = _hip.KernelTimer()
    with t:
        pdf(x, conditional_input=c)
    assert {k[0] for k in t.summary()} >= {"jf_merge_end", "jf_cond_gf_chain_split3_f32"}


def test_a_merge_the_library_declines_is_issued_launch_by_launch():
    """two broadcast g chains are not a combination the merged kernel carries: jf_merge_end replays them in order"""
    from jammy_flows_amd import _hip
    from jammy_flows_amd.layers.euclidean import gaussianization_flow as gfl
    fx = BY_NAME["c2_e4_gggg"]
    pdf = build_product(fx, torch.float32)
    x, _ = rows(fx, 1000, torch.float32)
    layers = list(pdf.layer_list[0])
    params = gfl.chain_permanent_row(layers, x)
    want = gfl.run_chain(layers, "inv", x, None, params)
    _hip.merge_begin()
    try:
        a = gfl.run_chain(layers, "inv", x, None, params)
        b = gfl.run_chain(layers, "inv", x, None, params)
        assert int(_hip.lib().jf_merge_captured()) == 2
        merged = _hip.merge_end(x)
    except BaseException:
        _hip.merge_abort()
        raise
    assert merged is False
    torch.cuda.synchronize()
    for got in (a, b):
        assert same(got[0], want[0]) and same(got[1], want[1])
    # an aborted capture launches nothing and leaves the thread's launches going to the GPU again
    _hip.merge_begin()
    gfl.run_chain(layers, "inv", x, None, params)
    _hip.merge_abort()
    again = gfl.run_chain(layers, "inv", x, None, params)
    assert same(again[0], want[0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("name", ["c2_e4_gggg", "c1_e2_gg", "g_e3_norot_noreg", "g_e3_nonorm_hh2"])
def test_lane_per_coordinate_chain_equals_lane_per_row_bit_for_bit(name, dtype):
    """incl. rows far from every mixture component (the scaled-sum fallback of both kernels) and a batch that is not a multiple of the tile"""
    from jammy_flows_amd import _hip
    fx = BY_NAME[name]
    if not helpers.product_supports(fx):
        pytest.skip("fixture not supported")
    pdf = build_product(fx, dtype)
    pdf.check_status = False
    x, c = rows(fx, 5000 + 37, dtype, seed=3)
    far = torch.linspace(-60.0, 60.0, 64, device=x.device, dtype=dtype)
    x[:64] = far[:, None] * torch.tensor([1.0, -0.5, 0.25, 2.0], device=x.device, dtype=dtype)[: x.shape[1]]
    L = _hip.lib()
    try:
        L.jf_gf_bcast_lane_rows(0)
        ref = pdf(x, conditional_input=c)
        L.jf_gf_bcast_lane_rows(1 << 40)
        got = pdf(x, conditional_input=c)
    finally:
        L.jf_gf_bcast_lane_rows(-1)
    assert torch.isfinite(ref[0][64:]).all()
    for g, w in zip(got, ref):
        assert same(g, w)


def test_plan_status_words_live_in_host_memory():
    fx = BY_NAME["c3_e4s2e4"]
    pdf = build_product(fx, torch.float32)
    pdf.check_status = "deferred"
    x, c = rows(fx, 4096, torch.float32)
    pf = pdf.planned_forward(x, conditional_input=c)
    assert not pf.host_status.is_cuda and pf.host_status.is_pinned() and pf.status.data_ptr() == pf.host_status.data_ptr()
    assert not any(name.startswith("jf_plan_add_copy") for name, *_ in pf.plan.calls)
    pf(x, c)
    pf.flush()
    bad = x.clone()
    bad[5, 0] = float("nan")
    pf(bad, c)
    torch.cuda.synchronize()
    assert int(pf.host_status.sum()) > 0
    with pytest.raises(Exception, match="nonfinite"):
        pf.flush()
    pf(x, c)
    pf.flush()


def test_unpluggable_plan_inputs_fall_back_to_the_eager_step():
    """x and conditional_input as column views of ONE tensor (overlapping plan slots) and an empty batch used to raise out of
    the recording with the thread's launch sink still set: every later launch was silently recorded instead of issued"""
    from jammy_flows_amd import _hip
    from jammy_flows_amd.main.default import PlanNotApplicable
    fx = BY_NAME["g_e3_rqs_cond"]
    pdf = build_product(fx, torch.float64)
    pdf.check_status = False
    x, c = rows(fx, 300, torch.float64)
    both = torch.cat([x, c], dim=1).contiguous()
    xv, cv = both[:, : x.shape[1]], both[:, x.shape[1]:]
    want = pdf(x, conditional_input=c)
    with pytest.raises(PlanNotApplicable):
        pdf.planned_forward(xv, conditional_input=cv)
    with pytest.raises(PlanNotApplicable):
        pdf.planned_forward(x[:0], conditional_input=c[:0])
    assert _hip._RECORDING is None
    pdf.use_step_plans = True
    try:
        got = pdf(xv, conditional_input=cv)                       # forward() falls back to the eager step
        empty = pdf(x[:0], conditional_input=c[:0])
        later = pdf(x, conditional_input=c)                       # and later steps still reach the GPU
    finally:
        pdf.use_step_plans = False
    assert empty[0].shape[0] == 0
    for g, w in zip(got, want):
        assert same(g, w)
    for g, w in zip(later, want):
        assert same(g, w)


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_pipelined_forward_equals_eager_forward(depth):
    """consecutive independent steps on alternating streams (pdf.pipelined_forward): every step's outputs are the eager step's, bit for bit,
    also when the inputs change from step to step, and a bad row is reported by drain()"""
    fx = BY_NAME["c3_e4s2e4"]
    pdf = build_product(fx, torch.float32)
    pdf.check_status = "deferred"
    batches = [rows(fx, 3000, torch.float32, seed=s) for s in range(5)]
    want = [pdf(x, conditional_input=c) for x, c in batches]
    pdf.flush_status()
    pipe = pdf.pipelined_forward(batches[0][0], conditional_input=batches[0][1], depth=depth)
    assert len({s.cuda_stream for s in pipe.streams}) == depth
    pending = [pipe.submit(x, c) for x, c in batches]
    pipe.drain()
    for t, w in zip(pending, want):
        for g, ww in zip(t.result(), w):
            assert same(g, ww)
    bad = batches[1][0].clone()
    bad[7, 2] = float("nan")
    pipe.submit(batches[0][0], batches[0][1])
    pipe.submit(bad, batches[1][1])
    with pytest.raises(Exception, match="nonfinite"):
        pipe.drain()
    t = pipe.submit(batches[2][0], batches[2][1])
    pipe.drain()
    assert same(t.result()[0], want[2][0])


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_sampler_on_far_apart_narrow_components(dtype):
    """the approach phase of the samplers (csrc/jf_gf.h: gf_approach) replaces the reference's 25 bisections
    (layers/bisection_n_newton.py:11-60); where its Newton / midpoint rules have not closed the bracket after 40 evaluations it now finishes
    with plain bisection.  Stress: ten narrow components (widths ~0.01) spread over +-60, base points out to |z| = 8 -- flat stretches of the
    cdf between the components.  Every finite row must round-trip: encode(decode(z)) = z, and no row may be left non-finite."""
    import jammy_flows_amd
    from jammy_flows_amd import _hip
    torch.manual_seed(11)
    B, D = 20000, 2
    pdf = jammy_flows_amd.pdf("e%d" % D, "g", options_overwrite={"g": {"inverse_function_type": "isigmoid", "replace_first_sigmoid_with_icdf": 0,
                                                                       "skip_model_offset": 1}}).to(dtype).cuda()
    layer = pdf.layer_list[0][0]
    c = layer.c_struct()
    assert not c.model_offset
    larr = _hip.gf_layer_array([c])
    kd, r0 = c.num_kde * D, c.hh_iter * D
    params = torch.zeros((1, layer.total_param_num), dtype=torch.float64, device="cuda")
    params[0, :r0] = torch.randn(r0, device="cuda")                                                        # Householder vectors
    params[0, r0:r0 + kd] = torch.linspace(-60.0, 60.0, kd, device="cuda")[torch.randperm(kd, device="cuda")]      # means
    params[0, r0 + kd:r0 + 2 * kd] = -12.0                                                                 # log-widths far below zero: the lower bound rules
    params[0, r0 + 2 * kd:r0 + 3 * kd] = torch.randn(kd, device="cuda")                                    # weights
    params = params.to(dtype)
    z = torch.empty((B, D), dtype=torch.float64, device="cuda").uniform_(-8.0, 8.0)
    z[::7] = z[::7].sign() * torch.empty((z[::7].shape[0], D), dtype=torch.float64, device="cuda").uniform_(7.0, 8.0)
    z = z.to(dtype)
    prev = _hip.FWD_TABLE_MIN_ROWS
    _hip.FWD_TABLE_MIN_ROWS = 1 << 62                            # the plain solves: approach phase + Newton stage on every lane
    try:
        status = _hip.new_status(z.device)
        x, ld = _hip.gf_chain("fwd", z, None, params, larr, 1, D, status=status)
    finally:
        _hip.FWD_TABLE_MIN_ROWS = prev
    words = status.cpu().tolist()
    assert torch.isfinite(x).all() and torch.isfinite(ld).all() and words[1] == 0, words
    zb, _ = _hip.gf_chain("inv", x, None, params, larr, 1, D)
    rt = ((zb - z).abs() / (1.0 + z.abs())).max(dim=1).values.sort().values
    # (rows the Newton stage flags as non-converged are exempt, as in the reference: their count is reported.  float64: it must stay small;
    #  float32 cannot meet the reference's 1e-7-class stopping rule on cdf slopes of ~100 per unit -- half the rows of this stress are flagged)
    if dtype == torch.float64:
        assert words[0] <= B // 200, words
    assert rt[B - 1 - words[0]].item() < (1e-6 if dtype == torch.float64 else 5e-3), rt[-5:]


def test_combine_rows_with_more_lists_than_one_launch_takes():
    """a pdf of more than 16 blocks has more per-block sums than one jf_combine_rows launch carries; the sums are then folded
    launch by launch in list order (same summation order), and non-contiguous inputs stay alive until their launch"""
    from jammy_flows_amd import _hip
    torch.manual_seed(1)
    B, n = 3000, 37
    wide = torch.randn(B, 2 * n, device="cuda")
    lds = [wide[:, 2 * i] for i in range(n)]                      # strided views: contiguous copies are made (and kept) by combine_rows
    blps = [torch.randn(B, device="cuda") for _ in range(19)]
    ld, blp, total = _hip.combine_rows(lds, blps)
    torch.cuda.synchronize()
    want_ld = lds[0].clone()
    for t in lds[1:]:
        want_ld = want_ld + t
    want_blp = blps[0].clone()
    for t in blps[1:]:
        want_blp = want_blp + t
    assert torch.equal(ld, want_ld) and torch.equal(blp, want_blp) and torch.equal(total, want_blp + want_ld)


def test_pipelined_forward_follows_parameter_updates():
    """an in-place parameter update (an optimiser step) between two submits: each stream's plan records again by itself"""
    fx = BY_NAME["c3_e4s2e4"]
    pdf = build_product(fx, torch.float32)
    pdf.check_status = "deferred"
    x, c = rows(fx, 2000, torch.float32, seed=4)
    pipe = pdf.pipelined_forward(x, conditional_input=c, depth=2)
    for t in [pipe.submit(x, c) for _ in range(3)]:
        t.result()
    pipe.drain()
    with torch.no_grad():
        for p in pdf.parameters():
            p.mul_(1.01)
    want = pdf(x, conditional_input=c)
    pdf.flush_status()
    got = [pipe.submit(x, c) for _ in range(4)]
    pipe.drain()
    for t in got:
        for g, w in zip(t.result(), want):
            assert same(g, w)
