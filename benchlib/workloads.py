"""constants of the roofline accounting, the BASELINE.json workloads and their seeded synthetic inputs"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")          # what the child runs of the benchmark execute

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-input MFMA peak (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md
MFMA_F64_PEAK_TFLOPS = 78.6    # v_mfma_f64_16x16x4_f64: 256 flop/clk/CU x 256 CUs x 2.4 GHz / 2 -- measured: scripts/probe/mfma64.hip

TRANS_PEAK_PER_S = 256 * 32 * 2.4e9   # quarter-rate transcendental issue (v_exp / v_log / v_rcp / v_sqrt _f32): 256 CUs x 128 lanes / 4 per clock x 2.4 GHz = 1.97e13 / s

WORKLOADS = {
    # fixture, pdf/flow strings, dtype of `value`, rows per GPU (weak), total rows (strong), SURVEY 8d algorithmic bytes per eval by dtype
    "c1": dict(fixture="c1_e2_gg", defs=("e2", "gg"), dtype="f64", rows=4096, total=4096, seed=1, bytes_per_eval={"f64": 48, "f32": 24}, flops_per_eval=0,
               metric="log-prob evals/sec (batch 4096), e2 / gg", desc="unconditional, the reference's CPU-runnable plumbing case"),
    "c2": dict(fixture="c2_e4_gggg", defs=("e4", "gggg"), dtype="f32", rows=1 << 20, total=1 << 20, seed=2, bytes_per_eval={"f32": 40, "f64": 80},
               flops_per_eval=0, metric="log-prob evals/sec (batch 2^20), e4 / gggg", desc="Gaussianization flow only, unconditional",
               # transcendental instructions per evaluation of the broadcast g kernel at these options (csrc/gf_kernels.hip gfb_chain_inv_kernel):
               # exp + rcp per (component, coordinate, layer), three logs per (coordinate, layer), ~6 in the inverse-normal stage of layer 0
               trans_per_eval=2 * 10 * 4 * 4 + 3 * 4 * 4 + 6 * 4),
    "c4": dict(fixture="c4_i1s1_ro", defs=("i1+s1", "r+o"), dtype="f32", rows=1 << 20, total=1 << 20, seed=4, bytes_per_eval={"f32": 100, "f64": 200},
               flops_per_eval=2304, metric="log-prob evals/sec (batch 2^20), i1+s1 / r+o", desc="RQ spline on the interval + circular spline on S1"),
    "c3": dict(fixture="c3_e4s2e4", defs=("e4+s2+e4", "gggg+f+gggg"), dtype="f32", rows=1 << 20, total=1 << 20, seed=3,
               bytes_per_eval={"f32": 4612, "f64": 9224}, flops_per_eval=145664,
               metric="log-prob evals/sec (batch 2^20), e4+s2+e4 / gggg+f+gggg",
               desc="unconditional pdf with autoregressive conditioning"),
    # SURVEY 8d's variant of C3: the 'f' layer with the docs-recommended nested spline flows (add_vertical_rq_spline_flow = 1,
    # add_circular_rq_spline_flow = 1; docs/source/usage/suggested_settings.rst:53-70): 46 parameters per row for the s2 block instead of 10
    "c3b": dict(fixture="c3b_e4s2e4_fsplines", defs=("e4+s2+e4", "gggg+f+gggg"), dtype="f32", rows=1 << 20, total=1 << 20, seed=3,
                bytes_per_eval={"f32": 4900, "f64": 9800}, flops_per_eval=154880,
                metric="log-prob evals/sec (batch 2^20), e4+s2+e4 / gggg+f+gggg with vertical + circular splines in f",
                desc="unconditional pdf with autoregressive conditioning, f with vertical + circular rational-quadratic splines"),
    "c5": dict(fixture="c5_e8s2_ggggv", defs=("e8+s2", "gggg+v"), dtype="f64", rows=1 << 19, total=1 << 22, seed=5,
               bytes_per_eval={"f64": 20912}, flops_per_eval=29216,
               metric="log-prob evals/sec (batch 2^19 per GPU = 2^22 over 8), conditional e8+s2 / gggg+v, AmortizableMLP rank 8",
               desc="conditional pdf (16 inputs), AmortizableMLP hidden 128 rank 8"),
}
REFERENCE_8THREAD = {"c1": {"value": 5.95e5, "what": "true reference, float64, batch 4096, 8 threads of the survey container (BASELINE.md section 2)"},
                     "c2": {"value": 3.97e5, "what": "true reference, float32, batch 2^20, 8 threads of the survey container (BASELINE.md section 2)"},
                     "c4": {"value": 1.09e6, "what": "true reference, float64, batch 2^20, 8 threads of the survey container (BASELINE.md section 2)"},
                     "c3": {"value": 3.71e4, "what": "true reference, float64, batch 2^18, 8 threads of the survey container (BASELINE.md section 2)"},
                     "c5": {"value": 2.44e4, "what": "true reference, float64, batch 2^16, 8 threads of the survey container (BASELINE.md section 2)"}}


def make_inputs(workload, n, seed):
    """SURVEY 8d inputs (c1 / c2 / c4: scripts/bench_configs_inputs.py, the same recipe for any pdf definition).  c3: x = [N(0,1.5^2)^4, theta = acos(U(-1,1)) clamped to [1e-3, pi-1e-3], phi = U(0,2pi), N(0,1.5^2)^4];
    c5: c ~ N(0, I_16), x = [N(0,1.5^2)^8, uniform on S2 as (theta, phi)].  Returns (x, cond or None)."""
    if workload in ("c1", "c2", "c4"):
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import fixture_io
        from bench_configs_inputs import inputs
        return inputs(fixture_io.load(WORKLOADS[workload]["fixture"]), n, seed)
    rng = np.random.default_rng(seed)
    if workload in ("c3", "c3b"):
        return np.concatenate([rng.normal(size=(n, 4)) * 1.5,
                               np.arccos(rng.uniform(-1, 1, size=(n, 1))).clip(1e-3, np.pi - 1e-3),
                               rng.uniform(0, 2 * np.pi, size=(n, 1)),
                               rng.normal(size=(n, 4)) * 1.5], axis=1), None
    x = np.concatenate([rng.normal(size=(n, 8)) * 1.5,
                        np.arccos(rng.uniform(-1, 1, size=(n, 1))).clip(1e-3, np.pi - 1e-3),
                        rng.uniform(0, 2 * np.pi, size=(n, 1))], axis=1)
    return x, rng.normal(size=(n, 16))
