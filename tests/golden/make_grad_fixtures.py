#!/usr/bin/env python3
"""Gradient fixtures: what the REAL reference's autograd returns for the golden fixtures' inputs and weights.

Runs only in the build container (imports /root/reference).  For each selected fixture (tests/golden/<name>.npz) the reference pdf is
rebuilt from the stored construction arguments + state_dict, and for the rows `rows` (all but the 8 adversarial tail rows, whose
50-sigma inputs make gradients of 1e30 and more)

    loss = -log_prob[rows].mean()          (the training objective of examples/jammy_flows.py:381-412, docs/source/usage/training.rst:24-44)

is back-propagated in float64.  For inverse_function_type = "inormal_full_pade" the rows whose mixture cdf comes within 1e-2 of 0.5 in some
layer are left out as well: there the reference evaluates sqrt(2 (sqrt(F^2 - ln_fac/a) - F)) with ln_fac = log_cdf + log_sf + ln 4 -> 0, a
difference of nearly equal numbers (its own comment at gaussianization_flow.py:636-638 calls the region "computationally unstable"), and the
derivative autograd takes through that expression carries relative errors of 1e-3 .. 1e-1 (row 112 of the fixture: |cdf - 0.5| = 2.3e-5).  Stored in tests/golden/grads/<name>.npz: loss, d loss / d x, d loss / d conditional_input, d loss / d every
parameter (keys = state_dict names), and the losses of 10 Adam steps (lr 1e-3) of the reference on that fixed batch.

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/make_grad_fixtures.py [name-substring ...]
"""
import contextlib
import io
import os
import sys

import numpy
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

with contextlib.redirect_stdout(io.StringIO()):
    import jammy_flows  # noqa: E402

import fixture_io  # noqa: E402

CASES = ["c1_e2_gg", "c2_e4_gggg", "c3_e4s2e4", "g_e1_g", "g_e3_ggg_cond", "g_e2_precise", "g_e2_crude", "g_e2_fullpade", "g_e2_softplusw",
         "g_e2_clampw", "g_e2_nosat", "g_e3_nonorm_hh2", "g_e3_norot_noreg", "g_e1e2e1_cond", "g_e1e2e1_cond_lowrank", "c5_e8s2_ggggv",
         "t_e3_gggt", "t_e3_gt_full_cond", "t_e2_tt_variants", "f_s2", "f_s2_cond_ff", "c4_i1s1_ro", "r_i1_m1p1_rr_cond", "o_s1_cond_oo", "m_s1_cond", "v_s2_cond_vv",
         "g_e2_skew_cond", "g_e3_center_mean", "g_e3_rot_angles", "g_e2_rot_cayley_cond", "g_e3_rot_triangular", "g_e4_all_options",
         # evaluated in EMBEDDING coordinates (force_embedding_coordinates=True: x = (x, y, z) on the sphere): the chart change sits inside the graph
         "f_s2_emb", "mix_s2e2_emb",
         # more than 8 Euclidean dimensions
         "g_e10_gg", "g_e10_ggggg", "g_e12_cond", "t_e10_full", "t_e10_diagonal", "t_e12_full_cond",
         # 'g' with nonlinear_stretch_type = "rq_splines" (gaussianization_flow.py:863-909)
         "g_e3_rqs", "g_e3_rqs_cond", "g_e3_rqs_bins32", "g_e2_rqs_bins24_cond",
         # only_last=True (main/default.py:1018, 1490: only the last layer of every sub-pdf is applied), with gradients
         "c3_e4s2e4@only_last", "g_e3_ggg_cond@only_last", "c4_i1s1_ro@only_last",
         # every remaining forward fixture (all but mix_e2s1i1, see tests/test_gpu_grad.py): option products of 'f', 'v' in both directions and with
         # spline potentials, 'm' / 'o' / 'r' variants with permanent parameters, the remaining 't' covariance types, 20 dimensions
         "c3b_e4s2e4_fsplines", "f_s2_correlated", "f_s2_extra_rot", "f_s2_identity_region", "f_s2_kappa_logb_clamp", "f_s2_rot_angles",
         "f_s2_rot_quat_sq", "f_s2_rot_xyz_mu", "f_s2_splines", "f_s2_splines_cond", "g_e20_g", "m_s1", "m_s1_nat1_rot", "o_s1", "o_s1_nat0_norot",
         "o_s1_nosmooth", "r_i1", "r_i1_fixopts", "r_i1_smooth2", "r_i1_smooth3", "t_e10_diagonal_symmetric", "t_e10_identity", "v_s2",
         "v_s2_nat1_rot", "v_s2_splines_cond", "v_s2_splines_nat1",
         # more than 16 bins
         "r_i1_bins24_cond", "r_i1_bins40", "o_s1_bins20_cond", "f_s2_splines_bins24",
         # more than 32 Euclidean dimensions
         "g_e40_gg", "g_e64_g_cond"]
N_ADV = 8
ADAM_STEPS = 10


def build(fx):
    with contextlib.redirect_stdout(io.StringIO()):
        pdf = jammy_flows.pdf(fx.pdf_defs, fx.flow_defs, **fx.kwargs)
    pdf.double()
    sd = {k: torch.from_numpy(numpy.ascontiguousarray(v)) for k, v in fx.state_dict().items()}
    pdf.load_state_dict(sd, strict=True)
    return pdf


def make(name):
    name, _, flag = name.partition("@")
    extra_kw = {"only_last": True} if flag == "only_last" else {}
    fx = fixture_io.load(name)
    pdf = build(fx)
    emb = bool(fx.meta["embedding"])
    B = fx["x"].shape[0]
    rows = numpy.arange(B - N_ADV)
    if "inormal_full_pade" in str(fx.kwargs):
        from jammy_flows.layers.euclidean import gaussianization_flow as gfm
        near = numpy.zeros(B, dtype=bool)
        orig = gfm.gf_block.sigmoid_inv_error_pass_given_cdf_sf

        def hook(self, lc, ls):
            near[:lc.shape[0]] |= ((torch.exp(lc) - 0.5).abs() < 1e-2).any(dim=1).numpy()
            return orig(self, lc, ls)
        gfm.gf_block.sigmoid_inv_error_pass_given_cdf_sf = hook
        try:
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                pdf(torch.from_numpy(fx["x"]))
        finally:
            gfm.gf_block.sigmoid_inv_error_pass_given_cdf_sf = orig
        rows = rows[~near[rows]]
    x = torch.from_numpy(fx["x"][rows]).clone().requires_grad_(True)
    cond = None
    if fx.get("cond") is not None:
        cond = torch.from_numpy(fx["cond"][rows]).clone().requires_grad_(True)
    with contextlib.redirect_stdout(io.StringIO()):
        logp = pdf(x, conditional_input=cond, force_embedding_coordinates=emb, **extra_kw)[0]
    assert torch.isfinite(logp).all(), name
    loss = -logp.mean()
    loss.backward()
    out = {"rows": rows, "loss": numpy.array(loss.item()), "x_grad": x.grad.numpy(), "logp": logp.detach().numpy()}
    if cond is not None:
        out["cond_grad"] = cond.grad.numpy()
    n_none = 0
    for k, p in pdf.named_parameters():
        if p.grad is None:
            n_none += 1
            continue
        out["pg/" + k] = p.grad.detach().numpy().copy()
    # ---- 10 Adam steps of the reference on the fixed batch
    pdf2 = build(fx)
    opt = torch.optim.Adam(pdf2.parameters(), lr=1e-3)
    xs = torch.from_numpy(fx["x"][rows])
    cs = None if fx.get("cond") is None else torch.from_numpy(fx["cond"][rows])
    traj = []
    for _ in range(ADAM_STEPS):
        opt.zero_grad()
        with contextlib.redirect_stdout(io.StringIO()):
            l2 = -pdf2(xs, conditional_input=cs, force_embedding_coordinates=emb, **extra_kw)[0].mean()
        l2.backward()
        opt.step()
        traj.append(l2.item())
    out["adam_losses"] = numpy.array(traj)
    path = os.path.join(HERE, "grads", name + ("_" + flag if flag else "") + ".npz")
    numpy.savez_compressed(path, **out)
    gmax = max(float(numpy.abs(v).max()) for k, v in out.items() if k.startswith("pg/"))
    print("%-26s rows=%d loss=%.6f max|dL/dparam|=%.3e max|dL/dx|=%.3e params without grad=%d adam: %.5f -> %.5f bytes=%d" % (
        name, len(rows), loss.item(), gmax, float(numpy.abs(out["x_grad"]).max()), n_none, traj[0], traj[-1], os.path.getsize(path)))


if __name__ == "__main__":
    sel = sys.argv[1:]
    for c in CASES:
        if sel and not any(s in c for s in sel):
            continue
        make(c)
