"""GPU tests (run with -m gpu) of the analysis reductions: pdf.approximate_coverage and pdf.entropy against what the REAL
reference returned for the same inputs / the same injected base samples (tests/golden/analysis/*.npz, make_analysis_fixtures.py)."""
import os

import numpy as np
import pytest
import torch

import fixture_io
from helpers import build_product, to_dev

pytestmark = pytest.mark.gpu
DIR = os.path.join(fixture_io.GOLDEN_DIR, "analysis")
CASES = ["c3_e4s2e4", "c4_i1s1_ro", "g_e3_ggg_cond", "c2_e4_gggg", "f_s2_cond_ff"]


def load(name):
    with np.load(os.path.join(DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", CASES)
def test_approximate_coverage_vs_reference(name):
    fx = fixture_io.load(name)
    g = load(name)
    pdf = build_product(fx, torch.float64)
    B = fx["x"].shape[0] - 8
    x = to_dev(fx["x"][:B], torch.float64)
    cond = to_dev(fx["cond"][:B], torch.float64) if fx.get("cond") is not None else None
    nsub = len(pdf.pdf_defs_list)
    cov = pdf.approximate_coverage(x, conditional_input=cond, num_percentile_points=50, sub_manifolds=[-1] + list(range(nsub)),
                                   force_embedding_coordinates=bool(fx.meta["embedding"]))
    assert np.allclose(cov["expected"], g["cov_expected"])
    for key in ["total"] + list(range(nsub)):
        ref_true, ref_d = g["cov_true/%s" % key], g["cov_diffs/%s" % key]
        assert np.abs(cov["logprob_diffs"][key] - ref_d).max() < 1e-7 * (1 + np.abs(ref_d).max())
        # counts may differ only for rows whose 2 dlogp sits within rounding of a chi^2 quantile
        assert np.abs(cov["true"][key] - ref_true).max() <= 1.0 / B + 1e-12, key
        assert cov["chi2_cdf_evals"][key].shape == ref_d.shape


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("emb", [True, False])
def test_entropy_with_injected_base_samples_vs_reference(name, emb):
    fx = fixture_io.load(name)
    g = load(name)
    pdf = build_product(fx, torch.float64)
    S = int(g["samplesize"])
    cond = to_dev(g["cond"], torch.float64) if "cond" in g else None
    nsub = len(pdf.pdf_defs_list)
    ent = pdf.entropy(sub_manifolds=[-1] + list(range(nsub)), conditional_input=cond, samplesize=S, force_embedding_coordinates=emb,
                      predefined_base=to_dev(g["z"], torch.float64))
    tag = "emb" if emb else "default"
    for key in ["total"] + list(range(nsub)):
        ref = g["entropy_%s/%s" % (tag, key)]
        got = ent[key].cpu().numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 1e-7 * (1 + np.abs(ref).max()), (key, got, ref)


def test_entropy_float32_large_sample_stays_on_device():
    fx = fixture_io.load("c3_e4s2e4")
    pdf = build_product(fx, torch.float32)
    ent = pdf.entropy(sub_manifolds=[-1, 0], samplesize=1 << 16)
    ent64 = build_product(fx, torch.float64).entropy(sub_manifolds=[-1, 0], samplesize=1 << 16)
    for k in ("total", 0):
        assert ent[k].is_cuda and ent[k].shape == (1,)
        assert abs(float(ent[k]) - float(ent64[k])) < 0.05      # two independent 65 536-sample Monte-Carlo estimates
