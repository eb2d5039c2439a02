#!/usr/bin/env python3
"""Golden vectors of the analysis reductions from the REAL reference: pdf.approximate_coverage on the fixture inputs and
pdf.entropy with INJECTED standard-normal base samples (torch.randn is patched for the duration of the call, so the product can be fed the
same samples).  Runs only in the build container.  Output: tests/golden/analysis/<name>.npz.

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/make_analysis_fixtures.py
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

CASES = [("c3_e4s2e4", 12, None), ("c4_i1s1_ro", 10, None), ("g_e3_ggg_cond", 16, 3), ("c2_e4_gggg", 32, None), ("f_s2_cond_ff", 8, 2)]


def build(fx):
    with contextlib.redirect_stdout(io.StringIO()):
        pdf = jammy_flows.pdf(fx.pdf_defs, fx.flow_defs, **fx.kwargs)
    pdf.double()
    pdf.load_state_dict({k: torch.from_numpy(numpy.ascontiguousarray(v)) for k, v in fx.state_dict().items()}, strict=True)
    return pdf


def make(name, S, n_cond):
    fx = fixture_io.load(name)
    pdf = build(fx)
    nsub = len(pdf.pdf_defs_list)
    subs = [-1] + list(range(nsub))
    out = {"samplesize": numpy.array(S)}
    B = fx["x"].shape[0] - 8
    x = torch.from_numpy(fx["x"][:B])
    cond = torch.from_numpy(fx["cond"][:B]) if fx.get("cond") is not None else None
    with contextlib.redirect_stdout(io.StringIO()):
        cov = pdf.approximate_coverage(x, conditional_input=cond, num_percentile_points=50, sub_manifolds=subs,
                                       force_embedding_coordinates=bool(fx.meta["embedding"]))
    out["cov_expected"] = cov["expected"]
    for k in cov["true"]:
        out["cov_true/%s" % k] = numpy.asarray(cov["true"][k])
        out["cov_diffs/%s" % k] = numpy.asarray(cov["logprob_diffs"][k])
    # entropy with injected base samples
    ci = None
    batch = 1
    if cond is not None:
        ci = cond[:n_cond].clone()
        batch = n_cond
    g = torch.Generator().manual_seed(17)
    z = torch.randn((S * batch, pdf.total_base_dim), generator=g, dtype=torch.float64)
    orig = torch.randn

    def fake(*a, **kw):
        size = kw.get("size", a[0] if a else None)
        assert tuple(size) == tuple(z.shape), (size, z.shape)
        return z.clone()
    for emb in (True, False):
        torch.randn = fake
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                ent = pdf.entropy(sub_manifolds=subs, conditional_input=ci, samplesize=S, force_embedding_coordinates=emb)
        finally:
            torch.randn = orig
        for k, v in ent.items():
            out["entropy_%s/%s" % ("emb" if emb else "default", k)] = v.detach().numpy()
    out["z"] = z.numpy()
    if ci is not None:
        out["cond"] = ci.numpy()
    path = os.path.join(HERE, "analysis", name + ".npz")
    numpy.savez_compressed(path, **out)
    print("%-20s coverage keys %s  entropy(emb) %s  bytes=%d" % (name, sorted(str(k) for k in cov["true"]),
                                                                 {k: float(numpy.asarray(v.detach()).mean()) for k, v in ent.items()},
                                                                 os.path.getsize(path)))


if __name__ == "__main__":
    for c in CASES:
        if len(sys.argv) > 1 and not any(s in c[0] for s in sys.argv[1:]):
            continue
        make(*c)
