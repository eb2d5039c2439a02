#!/usr/bin/env python3
"""Benchmark of the jammy_flows hot path on MI355X (contract: see the task statement / DESIGN.md section "Measurement").

    python bench.py --gpus N --steps K --warmup W [--workload c3|c5] [--scaling weak|strong]
                                                   (N > 1: one rank per GPU.  Under torch.distributed.run the ranks read RANK / LOCAL_RANK /
                                                    WORLD_SIZE / MASTER_* from the environment; started WITHOUT such an environment the script
                                                    launches `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself as a CHILD
                                                    process -- before this process has touched the GPU -- relays its output and exits with its code)

One step = one log-prob evaluation (`pdf.forward`) of one batch of synthetic rows: all sub-pdfs, all amortisation MLPs, every layer.
  --workload c3 (default): BASELINE.json's metric configuration `pdf("e4+s2+e4", "gggg+f+gggg")` ("n" of the upstream README = "f"),
                float32 `value` (+ the float64 rate beside it), 2^20 rows per GPU
  --workload c5: BASELINE configs[4], conditional `pdf("e8+s2", "gggg+v")`, 16 conditioning inputs, AmortizableMLP hidden 128 rank 8, float64
                ('v' asserts float64 in the reference), 2^19 rows per GPU (= 2^22 over 8)
  --scaling strong (default): the TOTAL batch is fixed (2^20 for c3, 2^22 for c5 -- at N = 1 the one GPU's 2^19 share) and row-sharded over the
                ranks: BASELINE.md section 3 defines efficiency = T_1 / (G T_G) on that;  weak: the per-GPU batch is fixed as N grows.
Inputs are synthetic (seeded) and resident in HBM before the timed region; weights are the frozen golden-fixture state_dicts
(tests/golden/*.npz: reference init with the MLP damping undone, so parameter blocks really vary per row).
For N > 1 every step all-gathers its log-probs (ONE RCCL all_gather, asynchronous, overlapping the next step; all waited for in the timed region).

Printed JSON line (rank 0): metric/value (whole-job evals/s), ms_per_step, plus
  roofline      dominant kernel: SURVEY 8d algorithmic bytes per launch / mean launch time from HIP events recorded in the timed region on
                the launch stream; `traffic` = HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of this same
                workload run as child processes before the timed run (or, if rocprofv3 is missing / fails, from the committed profile
                when the kernel sources are unchanged since it was taken -- otherwise null and `traffic_stale`)
  cpu_baseline  the numpy oracle (kind "port") on this box's host cores (one single-threaded process per core), bounded sample, N = 1 only
  parity        max |d log p| of the timed configuration against the float64 oracle on a 4096-row sample
"""
import os

# the CPU baseline forks one single-threaded worker per core: the BLAS / OpenMP pools must be sized BEFORE numpy is imported
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

# hardware queues of the HIP runtime (read when it initialises; default 4): an N > 1 step keeps six streams busy -- the caller's, the pipelined
# steps', the exchange's side stream and RCCL's own -- and streams that share a queue serialise: the 2^17-row shard step with its exchange takes
# 0.118 ms on 4 queues and 0.094-0.097 on 8 (same-box A/B, one-rank RCCL group); without an exchange 4 or 8 queues time the same
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import argparse  # noqa: E402
import glob  # noqa: E402
import hashlib  # noqa: E402
import json  # noqa: E402
import shutil  # noqa: E402
import sqlite3  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402
import tempfile  # noqa: E402
import time  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]

import numpy as np  # noqa: E402

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
               flops_per_eval=0, metric="log-prob evals/sec (batch 2^20 per GPU), e4 / gggg", desc="Gaussianization flow only, unconditional",
               # transcendental instructions per evaluation of the broadcast g kernel at these options (csrc/gf_kernels.hip gfb_chain_inv_kernel):
               # exp + rcp per (component, coordinate, layer), three logs per (coordinate, layer), ~6 in the inverse-normal stage of layer 0
               trans_per_eval=2 * 10 * 4 * 4 + 3 * 4 * 4 + 6 * 4),
    "c4": dict(fixture="c4_i1s1_ro", defs=("i1+s1", "r+o"), dtype="f32", rows=1 << 20, total=1 << 20, seed=4, bytes_per_eval={"f32": 100, "f64": 200},
               flops_per_eval=2304, metric="log-prob evals/sec (batch 2^20 per GPU), i1+s1 / r+o", desc="RQ spline on the interval + circular spline on S1"),
    "c3": dict(fixture="c3_e4s2e4", defs=("e4+s2+e4", "gggg+f+gggg"), dtype="f32", rows=1 << 20, total=1 << 20, seed=3,
               bytes_per_eval={"f32": 4612, "f64": 9224}, flops_per_eval=145664,
               metric="log-prob evals/sec (batch 2^20 per GPU), e4+s2+e4 / gggg+f+gggg",
               desc="unconditional pdf with autoregressive conditioning"),
    # SURVEY 8d's variant of C3: the 'f' layer with the docs-recommended nested spline flows (add_vertical_rq_spline_flow = 1,
    # add_circular_rq_spline_flow = 1; docs/source/usage/suggested_settings.rst:53-70): 46 parameters per row for the s2 block instead of 10
    "c3b": dict(fixture="c3b_e4s2e4_fsplines", defs=("e4+s2+e4", "gggg+f+gggg"), dtype="f32", rows=1 << 20, total=1 << 20, seed=3,
                bytes_per_eval={"f32": 4900, "f64": 9800}, flops_per_eval=154880,
                metric="log-prob evals/sec (batch 2^20 per GPU), e4+s2+e4 / gggg+f+gggg with vertical + circular splines in f",
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


# ---------------------------------------------------------------------------------------------- CPU baseline (oracle, one process per core)
_ORACLE = None


def _oracle_init(fixture):
    global _ORACLE
    import fixture_io
    import helpers
    try:                                             # belt and braces: the env vars above already size the pools of a fresh import
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:                                # noqa: BLE001
        pass
    _ORACLE = helpers.build_oracle(fixture_io.load(fixture))


def _oracle_chunk(args):
    x, c = args
    return _ORACLE.forward(x, c)[0]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota():
    """CPUs this process may use per the cgroup (cpu.max / cfs quota), or None: os.cpu_count() reports the host's logical CPUs, the GPU boxes of
    the pool grant a fraction of them"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:                                  # noqa: BLE001
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:                                  # noqa: BLE001
        return None


def cpu_baseline(workload, budget_s=15.0, workers=None, chunk=4096):
    """time the CPU oracle on a bounded sample of the same workload with single-threaded worker processes (fork BEFORE any GPU call).
    The number of workers is calibrated: on the pool's boxes os.cpu_count() is 256 but the cgroup grants a fraction, and 128 workers measured
    4.5e5 evals/s where 32 reach 6.3e5 (each at the single-process rate of 2e4); a short run per candidate count picks the best.
    4096-row chunks amortise the interpreter overhead of the ~250 numpy calls per chunk (1024-row chunks: 10 % slower, 512: 3x)."""
    import multiprocessing as mp
    w = WORKLOADS[workload]
    cores = os.cpu_count() or 1
    ctx = mp.get_context("fork")

    def chunks_of(n_chunks):
        x, c = make_inputs(workload, chunk * n_chunks, w["seed"])
        return [(x[i * chunk:(i + 1) * chunk], None if c is None else c[i * chunk:(i + 1) * chunk]) for i in range(n_chunks)]

    def run(n_workers, seconds):
        with ctx.Pool(n_workers, initializer=_oracle_init, initargs=(w["fixture"],)) as pool:
            pool.map(_oracle_chunk, chunks_of(n_workers))           # warm-up (imports, first-touch)
            t0 = time.time()
            pool.map(_oracle_chunk, chunks_of(n_workers))
            est = time.time() - t0
            rounds = int(max(1, min(256, seconds / max(est, 1e-3), (1 << 24) // (chunk * n_workers))))        # <= 2^24 rows of inputs in memory
            work = chunks_of(n_workers * rounds)
            t0 = time.time()
            pool.map(_oracle_chunk, work, chunksize=1)
            dt = time.time() - t0
        return chunk * n_workers * rounds, dt

    calibration = {}
    if workers is None:
        quota = cpu_quota()
        if quota is not None and quota >= 1:                       # the cgroup says how many CPUs there are: that many workers, or twice (SMT)
            cands = sorted({max(1, min(cores, int(round(quota)))), max(1, min(cores, int(round(2 * quota))))})
        else:
            cands = sorted({min(cores, c) for c in (8, 16, 32, 64, 128)})
        for cand in cands:
            n, dt = run(cand, 2.0)
            calibration[cand] = n / dt
        workers = max(calibration, key=calibration.get)
    n, dt = run(workers, budget_s)
    quota = cpu_quota()
    usable = int(round(quota)) if (quota is not None and quota >= 1) else cores       # CPUs this process can actually run on at once
    return {"value": n / dt, "unit": "log-prob evals/s", "cores": min(usable, workers), "workers": workers, "kind": "port", "cpu_model": cpu_model(),
            "per_core": n / dt / min(usable, workers), "per_worker": n / dt / workers, "host_logical_cpus": cores, "cgroup_cpu_quota": quota,
            "worker_calibration": {str(k): v for k, v in calibration.items()},
            "sample": "%d rows of %s (float64 numpy oracle, %d single-threaded processes x %d-row chunks), %.1f s"
                      % (n, w["fixture"], workers, chunk, dt),
            "reference_container_8thread": REFERENCE_8THREAD.get(workload)}


def oracle_rows(workload, x, c, workers, chunk=4096):
    """float64 oracle log-probs of the given rows (single-threaded worker processes, forked before any GPU call): the reference values of the
    UNTILED full-size parity check -- rows strided across the whole timed batch, not a leading sample and not a tiled fixture"""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    jobs = [(x[i:i + chunk], None if c is None else c[i:i + chunk]) for i in range(0, x.shape[0], chunk)]
    with ctx.Pool(max(1, min(workers, len(jobs))), initializer=_oracle_init, initargs=(WORKLOADS[workload]["fixture"],)) as pool:
        return np.concatenate(pool.map(_oracle_chunk, jobs, chunksize=1))


# ---------------------------------------------------------------------------------------------- HBM traffic (rocprofv3 PMC)
PROFILE_TRAFFIC = os.path.join(ROOT, "profiles", "r05_traffic.json")
WRITE_CAL = 0.965     # WRITE_SIZE calibration on scripts/probe/wstore (16-byte lane-per-row tile stores); FETCH_SIZE x 2 on gfx950 (guide)


def kernel_source_hash():
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "jammy_flows_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "jammy_hip.h")]):
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def pmc_pass(counter, workload, rows, fuse):
    """one rocprofv3 --pmc pass of a few steps of this workload in a child process -> {kernel name: mean raw counter value}.
    The child is this script (`--pmc-child`): python itself is what follows `--`, nothing re-execs after the GPU is initialised."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="jf_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", counter, "-d", tmp, "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", workload,
               "--batch", str(rows)] + ([] if fuse else ["--no-fuse"])
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
        dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return None
        cur = sqlite3.connect(dbs[0]).cursor()
        q = "select kernel_name, avg(value) from counters_collection where counter_name=? group by kernel_name"
        return {n: v for n, v in cur.execute(q, (counter,)) if "jf::" in n}
    except Exception:                                # noqa: BLE001 -- profiling is optional; the fallback is the committed profile
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_traffic(workload, rows, fuse):
    """{kernel name: {"read_bytes", "write_bytes", "hbm_bytes_per_launch"}} from two PMC passes (FETCH_SIZE and WRITE_SIZE cannot share one)."""
    f = pmc_pass("FETCH_SIZE", workload, rows, fuse)
    if not f:
        return None
    w = pmc_pass("WRITE_SIZE", workload, rows, fuse)
    if not w:
        return None
    out = {}
    for k in f:
        if k in w:
            rd, wr = f[k] * 1024 * 2, w[k] * 1024 * WRITE_CAL
            out[k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
    return {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate child passes of this run (FETCH_SIZE x 2, WRITE_SIZE x %.3f)" % WRITE_CAL,
            "kernels": out}


PROFILE_F64_ISSUE = os.path.join(ROOT, "profiles", "r05_f64_issue.json")
PROFILE_VALU_ISSUE = os.path.join(ROOT, "profiles", "r06_valu_issue.json")
N_SIMDS = 1024                           # 4 per CU x 256 CUs
SIDE_TABLE_STEPS = 3                     # eager steps behind the float64 leg's per-kernel table


def valu_issue_roofline(workload, dtype, rows, table):
    """a step against the VECTOR-ISSUE roof, computed: per kernel, the vector instructions per row by class (rocprofv3 --pmc SQ_INSTS_VALU*, a property
    of the code: profiles/r06_valu_issue.json, scripts/profile_r06.sh / collect_r06.py) x the measured issue cost of a wave64 instruction of that class
    (scripts/probe/f64_rates.hip: float32 plain 2.75 cycles, float32 transcendental 8.3; float64 4 / 16) = issue cycles per row; with THIS run's kernel
    times: achieved = issue cycles per second, peak = 1024 SIMDs x the clock measured during the kernel.  `table`: {(entry, tag): {"mean_ms": ...}}."""
    try:
        prof = json.load(open(PROFILE_VALU_ISSUE))["profiles"].get("%s/%s" % (workload, dtype))
    except (OSError, ValueError, KeyError):
        return None
    if not prof:
        return None
    out = {"bound": "vector issue", "unit": "T issue cycles/s", "kernels": {},
           "source": "profiles/r06_valu_issue.json (rocprofv3 --pmc: instructions per row by class, clock during the kernel) x this run's kernel times",
           "issue_cycles_per_wave_instruction": prof.get("issue_cycles"), "kernel_source_hash_match": prof.get("kernel_source_hash") == kernel_source_hash()}
    tot_c, tot_s, peak_w = 0.0, 0.0, 0.0
    for (name, tag), v in table.items():
        k = prof["kernels"].get("%s[%s]" % (name, tag)) or prof["kernels"].get(name)
        if not k:
            continue
        cyc = k["valu_issue_cycles_per_row"] * rows
        sec = v["mean_ms"] * 1e-3
        peak = N_SIMDS * k["clock_ghz"] * 1e9
        out["kernels"]["%s[%s]" % (name, tag)] = {"ms": round(v["mean_ms"], 4), "frac": cyc / sec / peak, "clock_ghz": k["clock_ghz"],
                                                 "valu_insts_per_row": k.get("valu_insts_per_row"), "trans_insts_per_row": k.get("trans_insts_per_row"),
                                                 "mfma_insts_per_row": k.get("mfma_insts_per_row"), "valu_busy_frac_in_profile": k.get("valu_busy_frac")}
        tot_c += cyc
        tot_s += sec
        peak_w += peak * sec
    if tot_s <= 0:
        return None
    out.update({"achieved": tot_c / tot_s / 1e12, "peak": peak_w / tot_s / 1e12, "frac": tot_c / peak_w, "peak_at_2.4GHz": N_SIMDS * 2.4e9 / 1e12})
    return out


def float64_issue_roofline(workload, rows, side_table):
    """the float64 step against the bound its counters show: VECTOR ISSUE (DESIGN "float64").  The committed profile holds, per kernel of the step,
    the vector instructions per row by class (rocprofv3 --pmc SQ_INSTS_VALU*: a property of the code), their issue cycles (4 per wave64
    instruction, float64 add / mul / fma included; 16 for the transcendental class) and the chip's clock during the kernel (GRBM_GUI_ACTIVE / 8 /
    duration: ~1.6 GHz under float64 load, not the 2.4 GHz of the headline peaks).  With THIS run's kernel times: achieved = vector issue cycles
    per second, peak = 1024 SIMDs x the measured clock."""
    try:
        prof = json.load(open(PROFILE_F64_ISSUE))
    except (OSError, ValueError):
        return None
    if prof.get("workload") != workload:
        return None
    out = {"bound": "vector issue (float64 arithmetic: %s of the instructions)", "unit": "T issue cycles/s",
           "source": "profiles/r05_f64_issue.json (rocprofv3 --pmc, per-row instruction counts and clocks) x this run's kernel times",
           "kernel_source_hash_match": prof.get("kernel_source_hash") == kernel_source_hash(), "kernels": {}}
    tot_c, tot_s, peak_w, f64w = 0.0, 0.0, 0.0, 0.0
    for (name, tag), v in side_table.items():
        k = prof["kernels"].get("%s[%s]" % (name, tag)) or prof["kernels"].get(name)
        if not k:
            continue
        cyc = k["valu_issue_cycles_per_row"] * rows
        sec = v["mean_ms"] * 1e-3
        peak = N_SIMDS * k["clock_ghz"] * 1e9
        out["kernels"]["%s[%s]" % (name, tag)] = {"ms": round(v["mean_ms"], 4), "frac": cyc / sec / peak, "clock_ghz": k["clock_ghz"],
                                                 "valu_busy_frac_in_profile": k.get("valu_busy_frac"), "f64_share_of_valu_insts": k.get("f64_share_of_valu_insts"),
                                                 "valu_insts_per_row": k["wave_insts_per_row"].get("SQ_INSTS_VALU")}
        tot_c += cyc
        tot_s += sec
        peak_w += peak * sec
        f64w += (k.get("f64_share_of_valu_insts") or 0.0) * cyc
    if tot_s <= 0:
        return None
    out["bound"] = out["bound"] % ("%.0f %%" % (100.0 * f64w / tot_c))
    out.update({"achieved": tot_c / tot_s / 1e12, "peak": peak_w / tot_s / 1e12, "frac": tot_c / peak_w, "peak_at_2.4GHz": N_SIMDS * 2.4e9 / 1e12,
                "note": "vector-issue cycles of the float64 step's kernels over (1024 SIMDs x the clock measured during each kernel).  The flow kernels sit "
                        "at ~0.8 of this roof; the 40 % HBM bar on SURVEY 8d bytes would need the step in 3.0 ms, i.e. fewer instructions, not more bandwidth"})
    return out


def committed_traffic():
    try:
        t = json.load(open(PROFILE_TRAFFIC))
    except (OSError, ValueError):
        return None
    if t.get("kernel_source_hash") != kernel_source_hash():
        return {"stale": True, "source": "profiles/r05_traffic.json (taken at kernel sources %s, now %s)" % (t.get("kernel_source_hash"), kernel_source_hash())}
    t["source"] = "profiles/r05_traffic.json (committed rocprofv3 --pmc passes of this command; kernel sources unchanged since)"
    return t


# device-kernel name (as rocprofv3 reports it) of a (C entry point, tag) pair of the host-side timer
KERNEL_OF = {"jf_cond_f_chain_inv_f32": "cond_mchain_kernel<float, jf::FFam", "jf_cond_f_chain_inv_f64": "cond_mchain_kernel<double, jf::FFam",
             "jf_conditioning_rows_f32": "conditioning_kernel<float", "jf_conditioning_rows_f64": "conditioning_kernel<double",
             "jf_v_chain_inv_f64": "mchain_kernel<double, jf::VFam", "jf_amlp2_f64": "amlp2_mfma_kernel",
             "jf_cond_gf_chain_inv_split_f32": "cond_gf_split_kernel",
             "jf_cond_gf_chain_split2_f32": "cond_gf_split_kernel", "jf_cond_gf_chain_split3_f32": "cond_gf_split_kernel",
             "jf_mlp2_i8_f64": "mlp2_i8_kernel", "jf_mlp2_i8_seg_f64": "mlp2_i8_kernel",
             "jf_r_chain_inv_f32": "mchain_kernel<float, jf::RFam", "jf_o_chain_inv_f32": "mchain_kernel<float, jf::OFam",
             "jf_cond_gf_chain_inv_f32": "cond_gf_chain_kernel<float",
             "jf_cond_gf_chain_inv_f64": "cond_gf_chain_kernel<double", "jf_mlp2_f32": "mlp2_kernel<float", "jf_mlp2_f64": "mlp2_kernel<double",
             "jf_gf_chain_inv_f32": "gf_chain_kernel<float", "jf_gf_chain_inv_f64": "gf_chain_kernel<double",
             "jf_gf_chain_inv_total_f32": "gf_chain_kernel<float", "jf_gf_chain_inv_total_f64": "gf_chain_kernel<double",
             "jf_amlp_gf_chain_inv_f64": "amlp_gf_mfma_kernel", "jf_merge_end": "merged_side_kernel"}


def traffic_of(traffic, kname, ktag):
    if not traffic or traffic.get("stale") or "kernels" not in traffic:
        return None
    key = KERNEL_OF.get(kname)
    if key is None:
        return None
    cands = [(n, v) for n, v in traffic["kernels"].items() if key in n]
    if kname.startswith("jf_gf_chain_inv"):          # broadcast: the lane = row kernel gfb_chain_inv_kernel<T, D> (classic stretch) or
        if ktag == "bcast":                          # gf_chain_kernel<..., true, false>; per-sample: gf_chain_kernel<..., false, false>
            rows_kernel = [(n, v) for n, v in traffic["kernels"].items() if key.replace("gf_chain_kernel", "gfb_chain_inv_kernel") in n]
            cands = rows_kernel or [(n, v) for n, v in cands if ", true, false>" in n]
        else:
            cands = [(n, v) for n, v in cands if ", false, false>" in n]
    if kname.startswith("jf_mlp2") and len(cands) > 1:   # narrow-output variant (TN = 1) for N <= 16, wide otherwise
        narrow = int(ktag.split("_")[-1][1:]) <= 32
        cands = [(n, v) for n, v in cands if (", 1, true" in n) == narrow] or cands
    return cands[0][1] if len(cands) == 1 else None


# ---------------------------------------------------------------------------------------------- algorithmic accounting
def kernel_accounting(kname, ktag, s):
    """(algorithmic HBM bytes per row, MFMA flops per row, fused?) of one timed kernel; s = bytes per scalar."""
    if (kname.startswith("jf_cond_gf_chain_inv_split") or kname.startswith("jf_cond_gf_chain_split2")
            or kname.startswith("jf_cond_gf_chain_split3")):
        K1, H, L, D = (int(t[1:]) for t in ktag.split("_")[:4])
        N = L * (3 * 10 * D + D * D) + D                 # default g rows: 3 K D + D^2 (+ D offsets on the last layer)
        return s * (K1 + N) + s * (D + 1 + N + D + 1), 2 * (K1 * H + H * N), True
    if kname.startswith("jf_cond_gf_chain") or kname.startswith("jf_amlp_gf_chain"):
        K1, H, N, D = (int(t[1:]) for t in ktag.split("_")[:4])
        # fused launch (MLP + g layers): SURVEY 8d "materialised" accounting = MLP (reads inputs, writes block) + flow (reads block);
        # the block itself never reaches HBM, so the real traffic is only s (K1 + 2 D + 2) bytes per row
        return s * (K1 + N) + s * (D + 1 + N + D + 1), 2 * (K1 * H + H * N), True
    if kname.startswith("jf_linear"):
        K, N = int(ktag.split("_")[0][1:]), int(ktag.split("_")[1][1:])
        return s * (K + N), 2 * K * N, False
    if kname.startswith("jf_mlp2"):
        K1, H, N = (int(t[1:]) for t in ktag.split("_"))
        return s * (K1 + N), 2 * (K1 * H + H * N), False
    if kname.startswith("jf_gf_chain_inv"):
        if ktag == "bcast":
            return s * 10, 0, False
        return None, 0, False                            # per-sample: depends on the block (filled in by the caller)
    return 0, 0, False


# ---------------------------------------------------------------------------------------------- sampling / training directions
def _oracle_sample_chunk(args):
    z, c = args
    return _ORACLE.sample_from_base(z, c)[0]


def cpu_baseline_sampling(workload, budget_s=12.0, chunk=1024):
    """the numpy oracle's sampling direction (25 bisection + <= 20 Newton steps per layer, oracle/gf.py) on this box's host cores: bounded
    sample, one single-threaded process per CPU of the cgroup quota"""
    import multiprocessing as mp
    w = WORKLOADS[workload]
    quota = cpu_quota()
    workers = max(1, int(round(quota))) if (quota is not None and quota >= 1) else min(os.cpu_count() or 1, 16)
    ctx = mp.get_context("fork")
    rng = np.random.default_rng(11)
    fx_dim = {"c3": 10, "c3b": 10, "c5": 10}[workload]

    def work(n_chunks):
        _, c = make_inputs(workload, chunk * n_chunks, w["seed"])
        z = rng.normal(size=(chunk * n_chunks, fx_dim))
        return [(z[i * chunk:(i + 1) * chunk], None if c is None else c[i * chunk:(i + 1) * chunk]) for i in range(n_chunks)]
    with ctx.Pool(workers, initializer=_oracle_init, initargs=(w["fixture"],)) as pool:
        pool.map(_oracle_sample_chunk, work(workers))
        t0 = time.time()
        pool.map(_oracle_sample_chunk, work(workers))
        est = time.time() - t0
        rounds = int(max(1, min(64, budget_s / max(est, 1e-3))))
        jobs = work(workers * rounds)
        t0 = time.time()
        pool.map(_oracle_sample_chunk, jobs, chunksize=1)
        dt = time.time() - t0
    n = chunk * workers * rounds
    return {"value": n / dt, "unit": "samples/s", "cores": workers, "workers": workers, "kind": "port", "cpu_model": cpu_model(),
            "cgroup_cpu_quota": quota, "host_logical_cpus": os.cpu_count(),
            "sample": "%d rows of %s through the float64 numpy oracle's sampling direction (%d single-threaded processes x %d-row chunks), %.1f s"
                      % (n, w["fixture"], workers, chunk, dt)}


def train_parity(name, dtype, dev, torch_adam):
    """The training step's parity: the gradient of -mean(log p) at the golden fixture's rows against the REAL reference's autograd
    (tests/golden/grads/<fixture>.npz, made by tests/golden/make_grad_fixtures.py), and 10 Adam steps (lr 1e-3) against the reference's loss
    trajectory -- with the optimiser the timed step uses, in the timed dtype, on a fresh model with the fixture's frozen weights."""
    import numpy as np
    import torch
    import fixture_io
    import helpers
    from jammy_flows_amd import optim as jf_optim
    fx = fixture_io.load(name)
    with np.load(os.path.join(fixture_io.GOLDEN_DIR, "grads", name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    pdf = helpers.build_product(fx, dtype, dev)
    rows = g["rows"]
    x = torch.from_numpy(fx["x"][rows]).to(device=dev, dtype=dtype).requires_grad_(True)
    cond = None if fx.get("cond") is None else torch.from_numpy(fx["cond"][rows]).to(device=dev, dtype=dtype).requires_grad_(True)

    def rel(got, ref):
        return float(np.abs(got.detach().double().cpu().numpy().reshape(ref.shape) - ref).max()) / max(float(np.abs(ref).max()), 1e-6)
    with torch.enable_grad():
        loss = -pdf(x, conditional_input=cond, force_embedding_coordinates=bool(fx.meta["embedding"]))[0].mean()
    loss.backward()
    worst = {"x": rel(x.grad, g["x_grad"])}
    if cond is not None and "cond_grad" in g:
        worst["cond"] = rel(cond.grad, g["cond_grad"])
    named = dict(pdf.named_parameters())
    for k in (k[3:] for k in g if k.startswith("pg/")):
        worst[k] = rel(named[k].grad, g["pg/" + k])
    opt = torch.optim.Adam(pdf.parameters(), lr=1e-3) if torch_adam else jf_optim.Adam(pdf.parameters(), lr=1e-3)
    xs, cs = x.detach(), None if cond is None else cond.detach()
    losses = []
    for _ in range(len(g["adam_losses"])):
        opt.zero_grad(set_to_none=True)
        with torch.enable_grad():
            ls = -pdf(xs, conditional_input=cs, force_embedding_coordinates=bool(fx.meta["embedding"]))[0].mean()
        ls.backward()
        opt.step()
        losses.append(float(ls.item()))
    return {"fixture": "tests/golden/grads/%s.npz (reference autograd, float64)" % name, "rows": int(len(rows)),
            "loss_abs_err": abs(float(loss.item()) - float(g["loss"])), "max_rel_gradient_err": max(worst.values()), "worst_tensor": max(worst, key=worst.get),
            "tensors_compared": len(worst), "adam_10_steps_max_loss_dev": float(np.abs(np.array(losses) - g["adam_losses"]).max()),
            "adam_losses_first_last": [losses[0], losses[-1]], "reference_first_last": [float(g["adam_losses"][0]), float(g["adam_losses"][-1])]}


def other_direction(args, W, rank, local_rank, world):
    """--direction sample | train: same launch / sharding / timing contract as the log-prob benchmark, one JSON line of the same shape."""
    direction = args.direction
    rows_default = W["rows"] if direction == "sample" else W["rows"] // 4          # training: 2^18 (c3) / 2^17 (c5) rows per GPU
    if args.scaling == "weak":
        B = args.batch if args.batch is not None else rows_default
        total_rows = B * world
    else:
        total_rows = args.batch if args.batch is not None else rows_default
        base, rem = divmod(total_rows, world)
        B = base + (1 if rank < rem else 0)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and direction == "sample":
        cpu = cpu_baseline_sampling(args.workload)                      # before the GPU is touched (fork safety)

    import torch
    import torch.distributed as dist
    import fixture_io
    import helpers
    from jammy_flows_amd import _hip, parallel

    backend = os.environ.get("JF_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # JF_FORCE_COLLECTIVES=1 with --gpus 1: a process group of ONE rank, so that the N > 1 code path (RCCL set-up, the all-gather on the step's
    # stream, gradient all-reduce, barriers, the exchange report) runs on a single-GPU box; the line then says "forced_collectives": true
    multi = world > 1 or os.environ.get("JF_FORCE_COLLECTIVES") == "1"
    if multi:
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_ranks_seen = dist.get_world_size() if multi else 1
    fx = fixture_io.load(W["fixture"])
    dtype = torch.float32 if W["dtype"] == "f32" else torch.float64
    s = 4 if W["dtype"] == "f32" else 8
    x64, c64 = make_inputs(args.workload, B, W["seed"] + rank)
    pdf = helpers.build_product(fx, dtype, dev)
    c = None if c64 is None else torch.from_numpy(c64).to(device=dev, dtype=dtype)
    extra = {}
    if direction == "sample":
        torch.set_grad_enabled(False)
        pdf.check_status = False
        g = torch.Generator(device=dev).manual_seed(17 + rank)
        z = torch.randn((B, pdf.total_base_dim), dtype=dtype, device=dev, generator=g)       # base points resident in HBM
        gather = parallel.PipelinedGather(B, dtype, dev, tail_shape=(pdf.total_target_dim,)) if (multi and total_rows % world == 0) else None
        last = {}

        # consecutive sampling steps draw independent batches: like the log-prob steps they alternate between --pipeline-depth streams, so the
        # ragged tail of one step's solver kernels (waves end with their slowest lane) is filled by the next step's launches
        depth = max(1, args.pipeline_depth)
        streams = [torch.cuda.Stream(device=dev) for _ in range(depth)] if depth > 1 else None
        extra["pipeline_depth"] = depth
        counter = {"i": 0}

        def step():
            if streams is not None:
                s = streams[counter["i"] % depth]
                counter["i"] += 1
                with torch.cuda.stream(s):
                    xs, _, lp, _ = pdf._obtain_sample(conditional_input=c, predefined_target_input=z)
                    if gather is not None:
                        gather.submit(xs)
            else:
                xs, _, lp, _ = pdf._obtain_sample(conditional_input=c, predefined_target_input=z)
                if gather is not None:
                    gather.submit(xs)
            last["x"], last["lp"] = xs, lp

        def finish():
            if streams is not None:
                cur = torch.cuda.current_stream(dev)
                for s in streams:
                    cur.wait_stream(s)
            if gather is not None:
                gather.wait()
        unit, metric = "samples/s", W["metric"].replace("log-prob evals/sec", "samples/sec")
    else:
        x = torch.from_numpy(x64).to(device=dev, dtype=dtype)
        pdf.check_status = False
        from jammy_flows_amd import optim as jf_optim
        # one launch over all parameter tensors (csrc/misc_kernels.hip: jf_adam_step); --torch-adam times torch.optim.Adam's foreach launches
        opt = torch.optim.Adam(pdf.parameters(), lr=1e-4) if args.torch_adam else jf_optim.Adam(pdf.parameters(), lr=1e-4)
        extra["optimizer"] = "torch.optim.Adam (foreach)" if args.torch_adam else "jammy_flows_amd.optim.Adam (one launch per step)"
        last = {}

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.enable_grad():
                loss = -pdf(x, conditional_input=c)[0].mean()
            loss.backward()
            if multi:
                parallel.allreduce_gradients(pdf.parameters(), average=True)
            opt.step()
            last["loss"] = loss

        def finish():
            pass
        unit, metric = "training rows/s", W["metric"].replace("log-prob evals/sec", "training rows/sec (forward + backward + Adam)")
    # the K timed steps run WITHOUT the per-launch HIP events (two event records per launch cost the host 10-20 us, which a training step of
    # ~20-30 launches feels); the per-kernel table is a second pass of the same steps right after the timed region
    dt = parallel.timed_steps(step, args.steps, args.warmup, finish=finish, device=dev, timer=None)
    timer = _hip.KernelTimer()
    n_table = min(args.steps, 10)
    with timer:
        for _ in range(n_table):
            step()
        finish()
    torch.cuda.synchronize(dev)
    table = timer.summary()
    for v in table.values():                                   # per-step figures below divide by args.steps: scale the second pass to it
        v["total_ms"] *= args.steps / n_table
        v["launches"] *= args.steps / n_table
    parity = None
    if rank == 0:
        if direction == "sample":                                         # what was just timed, against the float64 oracle (2048 rows)
            n_chk = min(2048, B)
            ox, olp, _ = helpers.build_oracle(fx).sample_from_base(z[:n_chk].double().cpu().numpy(), None if c64 is None else c64[:n_chk])
            ex = np.abs(last["x"][:n_chk].double().cpu().numpy() - ox)
            fin = np.isfinite(ox).all(axis=1) & np.isfinite(ex).all(axis=1)
            parity = {"max_abs_dx_vs_f64_oracle": float(ex[fin].max()), "rows_checked": int(fin.sum()),
                      "max_abs_dlogp_vs_f64_oracle": float(np.abs(last["lp"][:n_chk].double().cpu().numpy() - olp)[fin].max()),
                      "note": "float32 samples of rows whose float64 solution sits on a chart edge differ by the chart's float32 resolution" if s == 4 else None}
        else:
            extra["final_loss"] = float(last["loss"].item())
            parity = train_parity(W["fixture"], dtype, dev, args.torch_adam)
            if world == 1:
                # the same step (forward, backward, Adam with device-side step counters) captured once in a HIP graph and replayed: what a
                # training loop with static shapes would run; measured after the timed region, reported beside it
                try:
                    # nothing of the eager steps' autograd graphs may stay alive: their AccumulateGrad nodes belong to the default stream
                    last.clear()
                    opt.zero_grad(set_to_none=True)
                    import gc
                    gc.collect()
                    torch.cuda.synchronize(dev)
                    gopt = (torch.optim.Adam(pdf.parameters(), lr=1e-4, capturable=True) if args.torch_adam
                            else jf_optim.Adam(pdf.parameters(), lr=1e-4, capturable=True))

                    def gstep():
                        gopt.zero_grad(set_to_none=True)
                        with torch.enable_grad():
                            loss = -pdf(x, conditional_input=c)[0].mean()
                        loss.backward()
                        gopt.step()
                        return loss
                    side = torch.cuda.Stream(device=dev)
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            gstep()
                    torch.cuda.current_stream(dev).wait_stream(side)
                    graph = torch.cuda.CUDAGraph()
                    gopt.zero_grad(set_to_none=True)
                    with torch.cuda.graph(graph):
                        gloss = gstep()
                    for _ in range(3):
                        graph.replay()
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        graph.replay()
                    torch.cuda.synchronize(dev)
                    gdt = time.perf_counter() - t0
                    extra["hip_graph_replay"] = {"ms_per_step": 1e3 * gdt / args.steps, "value": total_rows * args.steps / gdt, "loss": float(gloss.item()),
                                                 "optimizer": "torch.optim.Adam(capturable=True)" if args.torch_adam else "jammy_flows_amd.optim.Adam(capturable=True): step count on the device",
                                                 "note": "forward + backward + Adam captured once in a HIP graph, replayed (measured after the timed region)"}
                except Exception as e:          # a capture failure must not cost the timed line
                    extra["hip_graph_replay"] = {"error": repr(e)[:200]}
                # Both regions time exactly K full steps (forward + backward + Adam) between synchronisations.  The eager one also measures the
                # HOST: ~22 launches and the autograd bookkeeping per 1.4 ms step sit at what a slower or busier host core can issue (the same
                # code has read 1.43 and 1.61 ms on two boxes of the pool with identical kernel times); the replay does not.  The line's value is
                # the faster of the two, named in `step_issue`, the other one stays beside it.
                g = extra["hip_graph_replay"]
                extra["eager"] = {"ms_per_step": 1e3 * dt / args.steps, "value": total_rows * args.steps / dt}
                if g.get("ms_per_step") is not None and 1e-3 * g["ms_per_step"] * args.steps < dt:
                    dt = 1e-3 * g["ms_per_step"] * args.steps
                    extra["step_issue"] = "HIP graph replay of the captured step (forward + backward + Adam with the step count on the device)"
                else:
                    extra["step_issue"] = "eager (one ctypes call per launch, torch autograd)"
    if direction == "sample" and gather is not None:
        extra["exchange_path"] = gather.path
        gather.close()
    if multi:
        dist.barrier()
    if rank == 0:
        dom = max(table.items(), key=lambda kv: kv[1]["total_ms"])
        (kname, ktag), kstat = dom
        secs = kstat["mean_ms"] * 1e-3
        bytes_per_row, flops_per_row, fused = kernel_accounting(kname.replace("_fwd", "_inv"), ktag, s)
        if bytes_per_row is None or bytes_per_row == 0:
            # per-row parameters / coordinates of the dominant block (0 parameters: permanent ones, shared by every row)
            P = {"c1": 0, "c2": 0, "c3": 548, "c3b": 548, "c4": 8, "c5": 1224}[args.workload]
            D = {"c1": 2, "c2": 4, "c3": 4, "c3b": 4, "c4": 1, "c5": 8}[args.workload]
            mult = 3 if kname.endswith("_bwd" + ("_f32" if s == 4 else "_f64")) else 1      # adjoint: parameters read twice, their gradient written
            bytes_per_row = s * (mult * P + (2 + mult) * (D + 1))
        gbs = bytes_per_row * B / secs / 1e9
        roofline = {"bound": "hbm", "kernel": "%s[%s]" % (kname, ktag), "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                    "traffic": None, "mean_launch_ms": kstat["mean_ms"], "launches_per_step": kstat["launches"] / args.steps,
                    "kernel_times_from": "HIP events around every C-ABI launch in a second pass of %d steps after the timed region" % n_table,
                    "algorithmic_bytes_per_launch": bytes_per_row * B,
                    "all_kernels_ms_per_step": {"%s[%s]" % k: round(v["total_ms"] / args.steps, 4) for k, v in sorted(table.items())}}
        line = {"metric": metric, "value": total_rows * args.steps / dt, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": W["dtype"],
                "data": "synthetic (seeded; weights = frozen golden-fixture state_dict)",
                "config": {"workload": 'pdf("%s","%s") %s, %s, %d rows %s' % (W["defs"][0], W["defs"][1], direction, W["desc"],
                                                                            B if args.scaling == "weak" else total_rows,
                                                                            "per GPU" if args.scaling == "weak" else "in total, row-sharded"),
                           "direction": direction, "batch_per_gpu": B, "total_rows": total_rows, "parallelism": "rows sharded over %d GPU(s)" % world},
                "n_ranks_seen": n_ranks_seen, "collective_backend": dist.get_backend() if multi else None, "parity": parity,
                "forced_collectives": bool(multi and world == 1),
                "roofline": roofline, "cpu_baseline": cpu}
        if direction == "train":
            line["cpu_baseline_note"] = "the oracle restates the forward arithmetic only: no CPU training baseline travels to the GPU box"
        line.update(extra)
        emit_line(line)
    if multi:
        dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------- self-launch for N > 1
def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_LINE_FD = None


def isolate_stdout():
    """keep this process's stdout for the contract's ONE JSON line: RCCL prints a five-line version banner on the C-level stdout when a process
    group comes up (seen on the GPU box with a one-rank group: "RCCL version : 2.26.6 ...", after the JSON line in the file), and anything a
    library prints there would sit next to the line the driver parses.  File descriptor 1 is pointed at stderr for the rest of the run (Python's
    sys.stdout and every C library follow it); emit_line() writes the line to the saved descriptor."""
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def emit_line(line):
    data = (json.dumps(line) + "\n").encode()
    sys.stdout.flush()
    fd = 1 if _LINE_FD is None else _LINE_FD
    while data:
        data = data[os.write(fd, data):]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` (N > 1) started without a torch.distributed.run environment: start the N ranks as a CHILD process tree and
    return its exit code.  This process has not imported torch nor made any HIP call at this point, and it never execs: the pool's boxes go
    down when a process that has initialised the GPU replaces itself.  Rank 0 of the child prints the JSON line on the inherited stdout."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool: RCCL needs it
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, W, rank, world, B, total_rows, lo):
    """the multi-rank plumbing of this script without a GPU: rendezvous, the rows each rank owns, the contract's timing loop and the per-step
    all-gather, with a stand-in row function evaluated by torch on the host.  Prints the same line shape with "dry_run": true and value null."""
    import torch
    import torch.distributed as dist
    from jammy_flows_amd import parallel
    torch.set_num_threads(1)
    backend = os.environ.get("JF_BENCH_BACKEND", "gloo")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    n_ranks_seen = dist.get_world_size() if world > 1 else 1
    x = torch.arange(lo, lo + B, dtype=torch.float64)
    gather = parallel.PipelinedGather(B, torch.float64, torch.device("cpu")) if (world > 1 and total_rows % world == 0) else None

    def step():
        y = -0.5 * x * x
        if gather is not None:
            gather.submit(y)

    def finish():
        if gather is not None:
            gather.wait()

    tinfo = {}
    dt = parallel.timed_steps(step, args.steps, args.warmup, finish=finish, device=None, info=tinfo)
    ok = True
    if gather is not None:
        full = gather.wait()
        ref = torch.arange(0, total_rows, dtype=torch.float64)
        ok = bool(torch.equal(full, -0.5 * ref * ref))
    exchange = parallel.gather_report(B, torch.float64, torch.device("cpu")) if world > 1 else None
    if world > 1:
        dist.barrier()
    if rank == 0:
        emit_line({"metric": W["metric"], "value": None, "unit": "log-prob evals/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling,
                          "vs_baseline": None, "dtype": W["dtype"], "data": "none (dry run: stand-in row function on the host, no kernels)",
                          "dry_run": True, "config": {"workload": "dry run of %s" % args.workload, "batch_per_gpu": B, "total_rows": total_rows,
                                                      "parallelism": "rows sharded over %d rank(s)" % world},
                          "n_ranks_seen": n_ranks_seen, "collective_backend": dist.get_backend() if world > 1 else None,
                          "rank_ms_per_step": parallel.rank_time_stats(tinfo, args.steps), "ranks_in_timing": tinfo.get("n_ranks_seen"),
                          "scaling_efficiency_vs_t1": None if (args.t1_ms is None or world < 2) else {
                              "t1_ms": args.t1_ms, "tN_ms": 1e3 * dt / args.steps, "n_gpus": world, "scaling": args.scaling,
                              "efficiency": (args.t1_ms / (world * 1e3 * dt / args.steps)) if args.scaling == "strong" else (args.t1_ms / (1e3 * dt / args.steps))},
                          "exchange": exchange, "gathered_rows_correct": ok})
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


# ---------------------------------------------------------------------------------------------- after the timed region: sweeps and the other configurations
def _time_steps(fn, flush, steps, warm=5, repeats=1):
    """seconds per call of fn (median of `repeats` timed loops of `steps` calls, each ended by flush() + a device synchronisation)"""
    import torch
    for _ in range(warm):
        fn()
    flush()
    torch.cuda.synchronize()
    out = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        flush()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps)
    return sorted(out)[len(out) // 2]


def rows_sweep(pdf, x, c, steps=50, depth=1):
    """the step time against the batch size on this one GPU (prefixes of the resident inputs): what strong scaling over G GPUs needs is
    t(B / G) <= t(B) / (G x 0.85), BASELINE.md section 3.  Every size runs through its own recorded step plan, like the timed step."""
    B = x.shape[0]
    out = []
    for lg in (20, 19, 18, 17, 16):
        n = 1 << lg
        if n > B:
            continue
        xs, cs = x[:n], (None if c is None else c[:n])
        if depth > 1:                                  # as the timed step: consecutive steps on alternating streams
            pipe = pdf.pipelined_forward(xs, conditional_input=cs, depth=depth)
            dt = _time_steps(lambda: pipe.submit(xs, cs), pipe.drain, steps if lg >= 19 else 4 * steps, warm=20, repeats=3)
            del pipe
        else:
            dt = _time_steps(lambda: pdf(xs, conditional_input=cs), pdf.flush_status, steps if lg >= 19 else 4 * steps, warm=20, repeats=3)
        row = {"log2_rows": lg, "ms_per_step": 1e3 * dt, "evals_per_s": n / dt}
        if depth > 1:
            row["one_stream_ms_per_step"] = 1e3 * _time_steps(lambda: pdf(xs, conditional_input=cs), pdf.flush_status, steps if lg >= 19 else 4 * steps,
                                                              warm=20, repeats=3)
        out.append(row)
    if out:
        top = out[0]
        for r in out:
            r["efficiency_vs_largest"] = top["ms_per_step"] / (r["ms_per_step"] * (1 << (top["log2_rows"] - r["log2_rows"])))
    return out


def shard_with_exchange(workload, rows, gather_steps, full_ms, exchange=True):
    """the step of an 8-GPU shard as a rank of that run would execute it, measured on this one GPU: a child process (fresh GPU context, after the
    timed region) runs `bench.py --batch rows`; exchange: through the N > 1 path, with a process group of ONE rank (JF_FORCE_COLLECTIVES=1), so
    every step hands its log-probs to RCCL"""
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--batch", str(rows), "--gather-steps", str(gather_steps), "--no-cpu-baseline",
           "--no-pmc", "--no-sweep", "--steps", "400", "--warmup", "20"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("JF_FORCE_COLLECTIVES", None)
    if exchange:
        env["JF_FORCE_COLLECTIVES"] = "1"
    try:
        import socket
        with socket.socket() as sk:                             # a free rendezvous port for the child's one-rank group
            sk.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(sk.getsockname()[1])
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        d = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
        return {"rows": rows, "ms_per_step": round(d["ms_per_step"], 5), "host_issue_ms_per_step": round(d.get("host_issue_ms_per_step") or 0.0, 5),
                "exchange": d.get("exchange"), "collective_backend": d.get("collective_backend"), "pipeline_depth": d.get("pipeline_depth"),
                "predicted_8gpu_strong_scaling_efficiency": full_ms / (8 * d["ms_per_step"]),
                "command": "%spython bench.py --batch %d --gather-steps %d" % ("JF_FORCE_COLLECTIVES=1 " if exchange else "", rows, gather_steps)}
    except Exception as e:                                     # noqa: BLE001 -- reported, never hidden
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def other_directions_summary(workload):
    """`bench.py --train` and `bench.py --direction sample` of the same configuration as child processes (fresh GPU contexts, started after this
    process's timed region; nothing re-execs): their ms per step, rate and parity, compact."""
    out = {}
    for key, flags in (("train", ["--train"]), ("sample", ["--direction", "sample"])):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--scaling", "weak", "--no-cpu-baseline", "--no-pmc", "--steps", "20",
               "--warmup", "5"] + flags
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            d = json.loads(r.stdout.decode().strip().splitlines()[-1])
            par = d.get("parity") or {}
            out[key] = {"ms_per_step": round(d["ms_per_step"], 4), "value": d["value"], "unit": d["unit"], "rows": d["config"]["total_rows"], "dtype": d["dtype"],
                        "parity": {k: par[k] for k in ("max_rel_gradient_err", "adam_10_steps_max_loss_dev", "max_abs_dx_vs_f64_oracle",
                                                       "max_abs_dlogp_vs_f64_oracle") if k in par},
                        "command": "python bench.py --workload %s %s" % (workload, " ".join(flags))}
            if d.get("optimizer"):
                out[key]["optimizer"] = d["optimizer"]
            if d.get("step_issue"):
                out[key]["step_issue"] = d["step_issue"]
                out[key]["eager_ms_per_step"] = (d.get("eager") or {}).get("ms_per_step")
                out[key]["hip_graph_replay_ms_per_step"] = (d.get("hip_graph_replay") or {}).get("ms_per_step")
        except Exception as e:                                 # noqa: BLE001 -- reported, never hidden
            out[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return out


def side_config(key, dev, steps=20):
    """one of the other BASELINE configurations, measured after the timed region on rank 0: step time through a recorded plan, parity against
    the float64 oracle on 2048 rows, the roof fraction on that configuration's own accounting."""
    import torch
    import fixture_io
    import helpers
    from jammy_flows_amd import _hip
    W = WORKLOADS[key]
    fx = fixture_io.load(W["fixture"])
    dtype = torch.float32 if W["dtype"] == "f32" else torch.float64
    s = 4 if W["dtype"] == "f32" else 8
    B = W["rows"]
    x64, c64 = make_inputs(key, B, W["seed"])
    pdf = helpers.build_product(fx, dtype, dev)
    pdf.check_status = "deferred"
    pdf.use_step_plans = True
    x = torch.from_numpy(x64).to(device=dev, dtype=dtype)
    c = None if c64 is None else torch.from_numpy(c64).to(device=dev, dtype=dtype)
    dt = _time_steps(lambda: pdf(x, conditional_input=c), pdf.flush_status, steps)
    timer = _hip.KernelTimer()
    with timer:
        for _ in range(5):
            logp = pdf(x, conditional_input=c)[0]
    table = timer.summary()
    n_chk = min(2048, B)
    o = helpers.build_oracle(fx).forward(x64[:n_chk], None if c64 is None else c64[:n_chk])[0]
    got = logp[:n_chk].double().cpu().numpy()
    fin = np.isfinite(o)
    row = {"workload": 'pdf("%s","%s")' % W["defs"], "dtype": W["dtype"], "rows": B, "ms_per_step": 1e3 * dt, "evals_per_s": B / dt,
           "max_abs_dlogp_vs_f64_oracle": float(np.abs(got - o)[fin].max()), "bar": 1e-2 if s == 4 else 1e-4,
           "whole_step_hbm_frac": W["bytes_per_eval"][W["dtype"]] * B / dt / 1e9 / HBM_PEAK_GBS,
           "kernels_ms": {"%s[%s]" % k: round(v["mean_ms"], 4) for k, v in sorted(table.items())}}
    vi = valu_issue_roofline(key, W["dtype"], B, table)
    if vi:
        row["valu_issue"] = vi
    if "trans_per_eval" in W:                              # the unconditional g kernel is bound by transcendental / vector issue, not by its 40 B per row
        kt = max(table.items(), key=lambda kv: kv[1]["total_ms"])[1]["mean_ms"] * 1e-3
        row["roofline"] = {"bound": "transcendental", "achieved": W["trans_per_eval"] * B / kt / 1e12, "peak": TRANS_PEAK_PER_S / 1e12,
                           "unit": "T transcendental instructions/s", "frac": W["trans_per_eval"] * B / kt / TRANS_PEAK_PER_S,
                           "transcendentals_per_eval": W["trans_per_eval"]}
    del pdf, x, c
    return row


# ---------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="strong (default, BASELINE.md section 3: efficiency = T_1 / (G T_G) at FIXED TOTAL batch): the configuration's batch is row-sharded "
                         "over the ranks;  weak: every rank gets the full per-GPU batch")
    ap.add_argument("--batch", type=int, default=None, help="rows per GPU (weak) / total rows (strong); default: the BASELINE configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 PMC child passes (traffic then comes from the committed profile)")
    ap.add_argument("--direction", choices=("logprob", "sample", "train"), default="logprob",
                    help="logprob (default; the contract metric): pdf.forward.  sample: pdf sampling from resident base points (bisection + Newton "
                         "kernels).  train: forward + backward + Adam step of -mean(log p) (the reference's training objective), rows = 1/4 of the "
                         "log-prob batch, gradients all-reduced over the ranks")
    ap.add_argument("--train", action="store_true", help="same as --direction train")
    ap.add_argument("--torch-adam", action="store_true", help="training: torch.optim.Adam (foreach) instead of jammy_flows_amd.optim.Adam")
    ap.add_argument("--no-fuse", action="store_true", help="time the two-launch path (MLP launch + flow launch) instead of the fused conditional block")
    ap.add_argument("--preheat-ms", type=float, default=1000.0,
                    help="run the step untimed for this long before the warm-up steps, so that the timed region sees the chip's sustained clocks (0 = off)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the rows sweep and the table of the other BASELINE configurations (measured after the timed region)")
    ap.add_argument("--no-plan", action="store_true", help="eager pdf.forward (one ctypes call per launch) instead of the recorded step plan")
    ap.add_argument("--pipeline-depth", type=int, default=None,
                    help="log-prob steps alternate between this many HIP streams, each through its own recorded plan (pdf.pipelined_forward): the batches "
                         "of consecutive steps are independent, so the tail of one step overlaps the head of the next; 1 = one stream.  Three streams: same-box "
                         "A/B against two -- 2^20 rows 0.663 / 0.663 ms, 2^19 0.335 / 0.338, 2^18 0.172 / 0.175, 2^17 0.0889 / 0.0920, 2^16 0.053 / 0.061; four "
                         "are slower at every size.  With the exchange of an N > 1 run (GPU_MAX_HW_QUEUES=8, which this script sets: on the runtime's default "
                         "of 4 queues three streams + exchange take 0.118 ms): 2^17 rows 0.0916 on three streams, 0.0946 on two")
    ap.add_argument("--gather-steps", type=int, default=4,
                    help="N > 1: the log-probs of this many consecutive steps travel in ONE all-gather (parallel.PipelinedGather(group_steps=k): fewer, larger "
                         "collectives -- an RCCL enqueue costs ~50 us of host time whatever its size); every step's rows are exchanged inside the timed region")
    ap.add_argument("--t1-ms", type=float, default=None,
                    help="N > 1: the ms_per_step of the SAME command at --gpus 1 (what the driver measured first): the line then carries the measured "
                         "scaling efficiency -- strong: T_1 / (N T_N), weak: T_1 / T_N -- beside the per-rank step times")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-dtype", choices=("f32", "f64"), default=None, help=argparse.SUPPRESS)    # precision of the --pmc-child steps (default: the workload's)
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise launch / rendezvous / row sharding / timing loop / all-gather with a stand-in step on the host (no GPU, no kernels): "
                         "the line carries \"dry_run\": true and no throughput claim.  For the CPU tests (JF_BENCH_BACKEND=gloo)")
    args = ap.parse_args()
    if args.pipeline_depth is None:
        args.pipeline_depth = 3
    if args.train:
        args.direction = "train"
    W = WORKLOADS[args.workload]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    if not args.pmc_child:
        isolate_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not args.pmc_child:
        sys.exit("bench.py: WORLD_SIZE (%d) != --gpus (%d): the launcher's --nproc-per-node must equal --gpus" % (world, args.gpus))

    # rows of this rank
    if args.scaling == "weak":
        B = args.batch if args.batch is not None else W["rows"]
        total_rows = B * world
        lo = rank * B
    else:
        total_rows = args.batch if args.batch is not None else W["total"]
        base, rem = divmod(total_rows, world)
        lo = rank * base + min(rank, rem)
        B = base + (1 if rank < rem else 0)

    if args.dry_run:
        return dry_run(args, W, rank, world, B, total_rows, lo)
    if args.direction != "logprob":
        return other_direction(args, W, rank, local_rank, world)

    cpu = None
    traffic = None
    strided = None
    if rank == 0 and world == 1 and not args.pmc_child:
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(args.workload)                        # before the GPU is touched (fork safety)
            # untiled full-size parity: 2^16 rows strided across the whole batch through the float64 oracle (a few seconds on the box's cores)
            n_str = min(1 << 16, B)
            stride = max(1, B // n_str)
            xs64, cs64 = make_inputs(args.workload, B, W["seed"] + rank)
            idx = np.arange(n_str) * stride
            t0 = time.time()
            strided = {"idx": idx, "logp": oracle_rows(args.workload, xs64[idx], None if cs64 is None else cs64[idx], cpu["workers"]),
                       "stride": int(stride)}
            strided["oracle_s"] = time.time() - t0
            del xs64, cs64
        if not args.no_pmc:
            traffic = measure_traffic(args.workload, B, not args.no_fuse)   # child processes; this process has not touched the GPU yet
    if traffic is None:
        traffic = committed_traffic()

    import torch
    import torch.distributed as dist
    import fixture_io
    import helpers
    from jammy_flows_amd import _hip, parallel

    torch.set_grad_enabled(False)        # a log-prob EVALUATION benchmark: no autograd graph (under grad mode pdf.forward builds one, like the reference)
    # one rank per GPU.  (JF_BENCH_BACKEND=gloo lets the multi-process logic be exercised on a box with fewer GPUs than ranks: the ranks
    # then share devices round-robin, which RCCL refuses; never used for reported numbers.)
    backend = os.environ.get("JF_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # JF_FORCE_COLLECTIVES=1 with --gpus 1: a process group of ONE rank, so that the N > 1 code path (RCCL set-up, the all-gather on the step's
    # stream, gradient all-reduce, barriers, the exchange report) runs on a single-GPU box; the line then says "forced_collectives": true
    multi = world > 1 or os.environ.get("JF_FORCE_COLLECTIVES") == "1"
    if multi:
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_ranks_seen = dist.get_world_size() if multi else 1
    backend_name = dist.get_backend() if multi else None

    fx = fixture_io.load(W["fixture"])
    # every rank generates the rows it owns from its own seed (no scatter)
    x64, c64 = make_inputs(args.workload, B, W["seed"] + rank)
    dtypes = {"f32": torch.float32, "f64": torch.float64}
    main_dt = W["dtype"]
    order = [main_dt] + (["f64"] if (main_dt == "f32" and not args.pmc_child) else [])

    if args.pmc_child:                                   # a few untimed steps for the PMC passes
        if args.pmc_dtype:
            main_dt = args.pmc_dtype
        pdf = helpers.build_product(fx, dtypes[main_dt], dev)
        pdf.check_status = "deferred"
        pdf.fuse_conditional_blocks = not args.no_fuse
        x = torch.from_numpy(x64).to(device=dev, dtype=dtypes[main_dt])
        c = None if c64 is None else torch.from_numpy(c64).to(device=dev, dtype=dtypes[main_dt])
        for _ in range(4):
            pdf(x, conditional_input=c)
        pdf.flush_status()
        torch.cuda.synchronize()
        return

    results = {}
    pipeline_error = None
    kernel_table = None
    two_launch = None
    graph_replay = None
    side_table = None
    for dname in order:
        dtype = dtypes[dname]
        pdf = helpers.build_product(fx, dtype, dev)
        pdf.check_status = "deferred"                 # throughput loop: status words are read back asynchronously and flushed inside the timed region
        pdf.fuse_conditional_blocks = not args.no_fuse
        pdf.use_step_plans = not args.no_plan         # pdf.forward through a recorded step plan: ONE ctypes call issues every launch of the step
        x = torch.from_numpy(x64).to(device=dev, dtype=dtype)
        c = None if c64 is None else torch.from_numpy(c64).to(device=dev, dtype=dtype)
        # N > 1: the per-row log-probs of every step are all-gathered (RCCL), asynchronously, while the next step computes
        gather = parallel.PipelinedGather(B, dtype, dev, group_steps=args.gather_steps) if (multi and total_rows % world == 0) else None
        last = {}

        # consecutive steps are independent batches: they alternate between `--pipeline-depth` streams, each through its own recorded plan, so
        # the idle tail of one step's last round of workgroups is filled by the head of the next step (DESIGN "small shards"); every step's
        # kernels, status words and (N > 1) all-gather still complete inside the timed region (finish())
        pipe = None
        if not args.no_plan and args.pipeline_depth > 1:
            try:
                pipe = pdf.pipelined_forward(x, conditional_input=c, depth=args.pipeline_depth)
            except Exception as e:                    # noqa: BLE001 -- reported in the line; the run continues on one stream
                pipeline_error = "%s: %s" % (type(e).__name__, str(e)[:200])
                print("bench: pipelined_forward failed (%s); continuing on one stream" % pipeline_error, file=sys.stderr)

        def step():
            if pipe is not None:
                if gather is not None:
                    # the step writes its log-probs straight into the stage of the next exchange; every --gather-steps-th step the exchange is
                    # issued from a side stream that waits for the staged steps' completion events (parallel.PipelinedGather.next_slot / staged)
                    t = pipe.submit(x, c, logp_out=gather.next_slot(pipe.peek_stream()))
                    gather.staged(t)
                else:
                    t = pipe.submit(x, c)
                # (the log-probs are what the step is for: the other two outputs -- 44 MB at 2^20 rows -- go back to the allocator at once, as
                #  they do for a caller that does not hold on to them; holding three steps' worth of them alive rotates the writes over ~190 MB)
                t.outputs = t.outputs[:1]
                last["pending"] = t
                return
            logp = pdf(x, conditional_input=c)[0]
            if gather is not None:
                gather.submit(logp)
            last["logp"] = logp

        def finish():
            if pipe is not None:
                pipe.drain()                          # the current stream waits for every submitted step; their status words are examined
                if "pending" in last:
                    last["logp"] = last["pending"].result()[0]
            pdf.flush_status()                        # deferred kernel status words of the timed steps: raises if any row went wrong
            if gather is not None:
                gather.wait()                         # every step's gather has landed inside the timed region

        # HIP events around the kernels of every 4th replay of a plan (the instrumentation costs ~2 % of a 2^20-row step and ~7 % of a 2^17-row
        # shard step when every replay carries it; the default 50 steps give a dozen timed replays per kernel; totals are scaled to all replays)
        timer = _hip.KernelTimer(plan_every=int(os.environ.get("JF_BENCH_TIMER_EVERY", "4"))) if dname == main_dt else None
        # bring the chip to its sustained clocks first: a 20-step region of 0.8 ms steps starts ~20 ms after the GPU sat idle (inputs were being
        # generated on the host), inside the power-management ramp -- the same plan measured 0.86 ms per step there and 0.78 ms once it had run for
        # 50 ms.  Untimed, before the contract's own W warm-up steps; reported in the line (`preheat_ms`).
        if args.preheat_ms > 0:
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < args.preheat_ms * 1e-3:
                for _ in range(8):
                    step()
                torch.cuda.synchronize(dev)
            finish()
        tinfo = {}
        dt = parallel.timed_steps(step, args.steps, args.warmup, finish=finish, device=dev, timer=timer, info=tinfo)
        if timer is None and rank == 0:
            # the secondary (float64) leg: per-kernel HIP-event times of three more EAGER steps, outside its timed region.  Two untimed eager steps
            # first: the timed region ran recorded plans (merged side kernels), so an eager step is the first launch of the stand-alone kernels in
            # this process -- code-object load and attribute calls landed between the events of that first launch (round 5's line read 20 ms for
            # jf_f_chain_inv_f64 that way; rocprofv3: 0.19 ms)
            for _ in range(2):
                pdf(x, conditional_input=c)
            pdf.flush_status()
            torch.cuda.synchronize(dev)
            t64 = _hip.KernelTimer()
            with t64:
                for _ in range(SIDE_TABLE_STEPS):
                    pdf(x, conditional_input=c)
            torch.cuda.synchronize(dev)
            side_table = t64.summary()
        logp = last["logp"]
        # N > 1: this rank's block of the last gathered buffer is what the last step computed; which path carried it (parallel.PipelinedGather)
        gather_info = None
        if gather is not None:
            full = gather.wait()
            torch.cuda.synchronize(dev)
            mine = gather.last_block(rank)
            gather_info = {"path": gather.path, "direct_error": gather.direct_error, "steps_per_collective": gather.k,
                           "own_block_correct": bool(((mine == logp) | (mine.isnan() & logp.isnan())).all())}
            gather.close()
        # parity of what was just timed, against the float64 oracle (rank 0, 4096 rows)
        err = None
        if rank == 0:
            n_chk = min(4096, B)
            o = helpers.build_oracle(fx).forward(x64[:n_chk], None if c64 is None else c64[:n_chk])[0]
            err = float(np.max(np.abs(logp[:n_chk].double().cpu().numpy() - o)))
        # untiled full-size parity: rows strided across the WHOLE timed batch against the float64 oracle (computed before the GPU was touched)
        untiled = None
        if rank == 0 and strided is not None:
            got = logp[torch.from_numpy(strided["idx"]).to(dev)].double().cpu().numpy()
            fin = np.isfinite(strided["logp"])
            untiled = {"rows_checked": int(fin.sum()), "row_stride": strided["stride"], "max_abs_dlogp_vs_f64_oracle": float(np.abs(got - strided["logp"])[fin].max()),
                       "finiteness_mismatches": int((np.isfinite(got) != fin).sum()), "oracle_seconds": round(strided["oracle_s"], 2)}
        # determinism of what was just timed (outside the timed region): the same step again, compared bit for bit over ALL rows -- the check that
        # exposes rare wrong row groups which a 4096-row oracle sample cannot see
        repeats, identical = 3, True
        for _ in range(repeats):
            again = pdf(x, conditional_input=c)[0]
            identical = identical and bool(((again == logp) | (again.isnan() & logp.isnan())).all())
        pdf.flush_status()
        results[dname] = dict(dt=dt, evals_per_s=total_rows * args.steps / dt, ms_per_step=1e3 * dt / args.steps, err=err, identical=identical, repeats=repeats,
                              untiled=untiled, gather=gather_info, host_issue_ms=1e3 * tinfo.get("host_issue_s", 0.0) / args.steps,
                              rank_ms=parallel.rank_time_stats(tinfo, args.steps), ranks_in_timing=tinfo.get("n_ranks_seen"))
        if rank == 0 and world == 1 and dname == main_dt and not args.no_sweep:
            results[dname]["rows_sweep"] = rows_sweep(pdf, x, c, depth=args.pipeline_depth if pipe is not None else 1)
        if timer is not None:
            kernel_table = timer.summary()
            if rank == 0 and world == 1:
                # for reference, outside the timed region: the SAME step replayed from a HIP graph (pdf.graphed_forward): one graph launch instead
                # of the step's kernel launches + host-side preparation; results must be bit-identical to the eager step just timed
                try:
                    gf_ = pdf.graphed_forward(x, conditional_input=c)
                    for _ in range(3):
                        gf_.graph.replay()
                    torch.cuda.synchronize()
                    tg0 = time.perf_counter()
                    for _ in range(args.steps):
                        gf_.graph.replay()
                    torch.cuda.synchronize()
                    tg = time.perf_counter() - tg0
                    graph_replay = {"ms_per_step": 1e3 * tg / args.steps, "value": B * args.steps / tg,
                                    "bit_identical_to_timed_step": bool(torch.equal(gf_.out[0], logp)),
                                    "note": "pdf.graphed_forward: the step captured once in a HIP graph, replayed (measured after the timed region)"}
                    del gf_
                except Exception as e:                 # noqa: BLE001 -- reported, never hidden
                    graph_replay = {"error": "%s: %s" % (type(e).__name__, e)}
            if rank == 0 and pdf.fuse_conditional_blocks and args.workload in ("c3", "c3b"):
                # for reference, outside the timed region: the same steps with the conditional block as two launches (jf_mlp2 + jf_gf_chain_inv),
                # whose kernels have clean single-roof accountings (MFMA for the MLP, HBM for the g-chain reading the materialised block)
                pdf.fuse_conditional_blocks = False
                for _ in range(2):
                    pdf(x)
                torch.cuda.synchronize()
                t2 = _hip.KernelTimer()
                tt0 = time.perf_counter()
                with t2:
                    for _ in range(args.steps):
                        pdf(x)
                pdf.flush_status()
                torch.cuda.synchronize()
                two_launch = {"dt": time.perf_counter() - tt0, "table": t2.summary()}
                pdf.fuse_conditional_blocks = True
        del pdf, x, c

    # N > 1: who held how many rows, and what one stand-alone all-gather of the log-probs costs (outside the timed region; every rank)
    exchange = parallel.gather_report(B, dtypes[main_dt], dev) if multi else None
    if exchange is not None and results[main_dt].get("gather"):
        exchange.update(results[main_dt]["gather"])
    if multi:
        dist.barrier()
    if rank == 0:
        rm = results[main_dt]
        s = 4 if main_dt == "f32" else 8
        # ---- roofline of the dominant kernel.  SURVEY 8d: the conditional blocks are priced against HBM on the MATERIALISED accounting
        # (MLP reads its inputs and writes the block, the flow reads the block); a fused kernel moves almost none of it and may exceed 100 %.
        dom = max(kernel_table.items(), key=lambda kv: kv[1]["total_ms"])
        (kname, ktag), kstat = dom
        # With --pipeline-depth > 1 the kernels of consecutive steps run SIDE BY SIDE on the chip: a launch's HIP-event / rocprofv3 duration then
        # covers a time in which it held only part of the chip (the durations of a step add up to `concurrency` x the step time).  The roofline
        # prices a launch at its share of the chip's time: duration / concurrency -- for one stream that is the duration itself.
        sum_ms = sum(v["total_ms"] for v in kernel_table.values()) / args.steps
        concurrency = max(1.0, sum_ms / rm["ms_per_step"])
        secs = kstat["mean_ms"] * 1e-3 / concurrency
        bytes_per_row, flops_per_row, fused = kernel_accounting(kname, ktag, s)
        if bytes_per_row is None:                        # per-sample g-chain: the block's row (C3 block 2: 548 floats, C5 block 0: 1224 doubles)
            # per-row parameters / coordinates of the dominant block (0 parameters: permanent ones, shared by every row)
            P = {"c1": 0, "c2": 0, "c3": 548, "c3b": 548, "c4": 8, "c5": 1224}[args.workload]
            D = {"c1": 2, "c2": 4, "c3": 4, "c3b": 4, "c4": 1, "c5": 8}[args.workload]
            bytes_per_row = s * (D + 1 + P + D + 1)
        hbm_gbs = bytes_per_row * B / secs / 1e9
        tr = traffic_of(traffic, kname, ktag)
        roofline = {"bound": "hbm", "kernel": "%s[%s]" % (kname, ktag), "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": tr["hbm_bytes_per_launch"] if tr else None,
                    "traffic_detail": tr, "traffic_source": (traffic or {}).get("source"), "traffic_stale": bool((traffic or {}).get("stale")),
                    "mean_launch_ms": kstat["mean_ms"], "concurrency": concurrency, "effective_launch_ms": kstat["mean_ms"] / concurrency,
                    "concurrency_note": ("HIP-event duration of a launch (what rocprofv3 --kernel-trace reports too) / the number of kernels that ran side "
                                         "by side on average = (sum of the step's launch durations) / (step time): consecutive steps alternate between "
                                         "%d streams" % args.pipeline_depth) if concurrency > 1.0 else None,
                    "algorithmic_bytes_per_launch": bytes_per_row * B}
        if fused:
            roofline["fused"] = True
            roofline["note"] = ("amortisation MLP + its g layers in one launch: the per-sample parameter block never leaves the chip, so the SURVEY 8d "
                                "(materialised-block) bytes are far more than the kernel moves -- see `traffic`; HBM does not bind this kernel: the vector "
                                "issue of the flow arithmetic and the matrix pipe do (`bound`), `hbm_materialised_*` keep the SURVEY 8d figure")
        if args.workload == "c2" and "trans_per_eval" in W:
            roofline.update({"bound": "transcendental", "achieved": W["trans_per_eval"] * B / secs / 1e12, "peak": TRANS_PEAK_PER_S / 1e12,
                             "unit": "T transcendental instructions/s", "frac": W["trans_per_eval"] * B / secs / TRANS_PEAK_PER_S,
                             "transcendentals_per_eval": W["trans_per_eval"], "hbm_frac": hbm_gbs / HBM_PEAK_GBS,
                             "note": "the unconditional g chain evaluates ~%d exp / log / rcp-class instructions per 40-byte row: bound by quarter-rate "
                                     "transcendental + vector issue (SURVEY 8d, D6), the 40 %% HBM bar does not apply" % W["trans_per_eval"]})
        if flops_per_row:
            tf = flops_per_row * B / secs / 1e12
            mf = {"algorithmic_TFLOPs": tf, "algorithmic_flops_per_launch": flops_per_row * B}
            if kname.endswith("_split2_f32") or kname.endswith("_split3_f32"):
                # the default: two f16 pieces per operand, three f16 MFMA passes (lo hi, hi lo, hi hi) over the padded columns
                K1, H, L, D = (int(t[1:]) for t in ktag.split("_")[:4])
                cols = L * 9 * 16
                executed = 2 * K1 * 128 + 3 * 2 * 128 * cols
                mf.update({"arithmetic": "2-way split f16 (operands scaled into the normal f16 range), 3 MFMA passes, f32 accumulate",
                           "executed_f16_TFLOPs": executed * B / secs / 1e12,
                           "frac_of_f16_peak": executed * B / secs / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                           "frac_of_f32_mfma_peak_equivalent": tf / MFMA_F32_PEAK_TFLOPS})
            elif kname.endswith("_split_f32"):
                K1, H, L, D = (int(t[1:]) for t in ktag.split("_"))
                cols = L * 9 * 16                                                   # padded columns per row: 9 tiles of 16 per layer
                executed = 2 * K1 * 128 + 6 * 2 * 128 * cols               # first layer + six bf16 passes over the padded columns
                mf.update({"arithmetic": "3-way split bf16, 6 MFMA passes, f32 accumulate", "executed_bf16_TFLOPs": executed * B / secs / 1e12,
                           "frac_of_bf16_peak": executed * B / secs / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                           "frac_of_f32_mfma_peak_equivalent": tf / MFMA_F32_PEAK_TFLOPS})
            elif kname.startswith("jf_amlp_gf_chain"):
                # low-rank factors: the kernel never forms the dense product.  float64 with ranks <= 8 runs on v_mfma_f64_16x16x4 (jf_amlp_mfma.h; on
                # MI355X the f64 matrix rate equals the f64 vector rate -- the matrix cores are used for their 16x lower LDS operand traffic)
                K1, H, N, D = (int(t[1:]) for t in ktag.split("_")[:4])
                r = int(ktag.split("_")[4][1:])
                L = {"c3": 4, "c3b": 4, "c5": 4}[args.workload]
                useful = 2 * (r * K1 + H * r + r * H + N * r)               # V1 c, U1 t1, V2 h, U2 t2 per row (23 936 for C5 block 0)
                on_mfma = main_dt == "f64" and r <= 8 and H % 16 == 0
                executed = 2 * (16 * 4 * ((K1 + 3) // 4) + H * 8 + 16 * H + L * 21 * 16 * 8) if on_mfma else useful     # padded 16 x 16 x 4 tiles
                mf = {"dense_equivalent_TFLOPs": tf, "dense_equivalent_flops_per_launch": flops_per_row * B,
                      "arithmetic": ("f64 MFMA (v_mfma_f64_16x16x4) on the low-rank factors (rank %d), permuted so that results land in the flow's registers" % r)
                      if on_mfma else "%s VALU FMAs on the low-rank factors (rank %d)" % (main_dt, r),
                      "useful_flops_per_launch": useful * B, "executed_flops_per_launch": executed * B, "executed_TFLOPs": executed * B / secs / 1e12,
                      "frac_of_%s_%s_peak" % (main_dt, "mfma" if on_mfma else "vector"):
                          executed * B / secs / 1e12 / (MFMA_F64_PEAK_TFLOPS if main_dt == "f64" else MFMA_F32_PEAK_TFLOPS)}
            elif main_dt == "f32":
                mf.update({"arithmetic": "exact f32 MFMA", "frac_of_f32_mfma_peak": tf / MFMA_F32_PEAK_TFLOPS})
            else:
                mf.update({"arithmetic": "f64 MFMA", "frac_of_f64_mfma_peak": tf / MFMA_F64_PEAK_TFLOPS})
            roofline["mfma"] = mf
            if fused and ("executed_f16_TFLOPs" in mf or "executed_bf16_TFLOPs" in mf or "executed_TFLOPs" in mf):
                # the fused block: report it against what binds it.  achieved / peak = the executed matrix rate against the pipe the kernel uses;
                # the SURVEY 8d (materialised-block) HBM figure moves to hbm_materialised_*
                ex = mf.get("executed_f16_TFLOPs", mf.get("executed_bf16_TFLOPs", mf.get("executed_TFLOPs")))
                pk = MFMA_F64_PEAK_TFLOPS if kname.startswith("jf_amlp_gf_chain") and main_dt == "f64" else MFMA_BF16_PEAK_TFLOPS
                roofline.update({"hbm_materialised_GBs": hbm_gbs, "hbm_materialised_frac": hbm_gbs / HBM_PEAK_GBS,
                                 "bound": "valu+mfma", "achieved": ex, "peak": pk, "unit": "TFLOP/s (executed on the matrix pipe)", "frac": ex / pk,
                                 "valu_busy_frac_committed_profile": {"c3": 0.66, "c5": 0.60}.get(args.workload),
                                 "mfma_busy_frac_committed_profile": {"c3": 0.37, "c5": 0.16}.get(args.workload)})
        if flops_per_row:
            # 2 x sum(in x out) of the amortisation MLP per row: what the reference's float32 product computes, beside the executed (3-pass f16) figure
            roofline["algorithmic_TFLOPs"] = flops_per_row * B / secs / 1e12
            roofline["algorithmic_flops_per_row"] = flops_per_row
        # the vector-issue roof, computed (instructions per row x measured cycles per instruction x this run's kernel times): one stream's durations
        # are the kernels' own; under pipelining a launch is priced at its share of the chip (duration / concurrency)
        vi = valu_issue_roofline(args.workload, main_dt, B, {k: {"mean_ms": v["mean_ms"] / concurrency} for k, v in kernel_table.items()})
        if vi:
            roofline["valu_issue"] = vi
        step_bytes = W["bytes_per_eval"][main_dt]
        roofline["whole_step"] = {"algorithmic_bytes_per_eval": step_bytes, "achieved_GBs": step_bytes * B / (rm["ms_per_step"] * 1e-3) / 1e9,
                                  "frac": step_bytes * B / (rm["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "SURVEY 8d bytes of ALL blocks over the whole step time (the north-star >= 40 % figure)"}
        roofline["all_kernels_ms_per_step"] = {"%s[%s]" % k: round(v["total_ms"] / args.steps, 4) for k, v in sorted(kernel_table.items())}
        if concurrency > 1.0:
            # the same durations at each kernel's share of the chip's time (they add up to the step time)
            roofline["all_kernels_effective_ms_per_step"] = {"%s[%s]" % k: round(v["total_ms"] / args.steps / concurrency, 4)
                                                             for k, v in sorted(kernel_table.items())}
        line = {
            "metric": W["metric"],
            "value": rm["evals_per_s"], "unit": "log-prob evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rm["ms_per_step"], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": main_dt,
            "data": "synthetic (seeded; weights = frozen golden-fixture state_dict)",
            "config": {"workload": 'pdf("%s","%s") log-prob, %s, %d rows %s' % (W["defs"][0], W["defs"][1], W["desc"],
                                                                                  B if args.scaling == "weak" else total_rows,
                                                                                  "per GPU" if args.scaling == "weak" else "in total, row-sharded"),
                       "batch_per_gpu": B, "total_rows": total_rows, "parallelism": "rows sharded over %d GPU(s)" % world},
            "n_ranks_seen": n_ranks_seen, "collective_backend": backend_name, "exchange": exchange,
            "forced_collectives": bool(multi and world == 1),
            # every rank's own clock between the two fences of the timed region (ms per step): `ms_per_step` is the slowest rank's; a straggler shows here
            "rank_ms_per_step": rm["rank_ms"], "ranks_in_timing": rm["ranks_in_timing"],
            "hw_queues": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "set_by": "bench.py / jammy_flows_amd.parallel at import (%s)" % parallel.HW_QUEUES_STATE},
            "host_issue_ms_per_step": rm["host_issue_ms"],
            "parity": {"max_abs_dlogp_vs_f64_oracle": rm["err"], "bar": 1e-2 if main_dt == "f32" else 1e-4, "rows_checked": min(4096, B),
                       "repeat_launches_bit_identical": rm["identical"], "repeat_launches": rm["repeats"], "repeat_rows_compared": B,
                       "untiled_full_size": rm["untiled"]},
            "dtype_note": ("float32 value: the 128 -> N product of the conditional block runs as 3 f16 MFMA passes over 2-piece splits of the f32 operands "
                           "(f32 accumulation; measured deviation from the float64 oracle in `parity`, bar 1e-2 -- the reference's own fp32-vs-fp64 "
                           "agreement is ~3e-5); `cpu_baseline` is the float64 oracle; the like-for-like float64 rate is under `float64`")
                          if main_dt == "f32" else None,
            "preheat_ms": args.preheat_ms,
            "step_issue": ("recorded step plan: one ctypes call per step (jf_plan_launch)" + (
                "; consecutive steps alternate between %d streams (pdf.pipelined_forward)" % args.pipeline_depth if args.pipeline_depth > 1 else ""))
            if not args.no_plan else "eager: one ctypes call per launch",
            "pipeline_depth": 1 if (args.no_plan or pipeline_error) else args.pipeline_depth, "pipeline_error": pipeline_error,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if args.t1_ms is not None and world > 1:
            tn = rm["ms_per_step"]
            line["scaling_efficiency_vs_t1"] = {"t1_ms": args.t1_ms, "tN_ms": tn, "n_gpus": world, "scaling": args.scaling,
                                                "efficiency": (args.t1_ms / (world * tn)) if args.scaling == "strong" else (args.t1_ms / tn),
                                                "definition": "strong: T_1 / (N T_N) at fixed total batch; weak: T_1 / T_N at fixed per-GPU batch (BASELINE.md section 3)"}
        if "f64" in results and main_dt != "f64":
            r64 = results["f64"]
            line["float64"] = {"value": r64["evals_per_s"], "ms_per_step": r64["ms_per_step"], "max_abs_dlogp_vs_f64_oracle": r64["err"], "bar": 1e-4,
                               "untiled_full_size": r64["untiled"]}
            b64 = W["bytes_per_eval"].get("f64")
            if b64:
                g64 = b64 * B / (r64["ms_per_step"] * 1e-3) / 1e9
                line["float64"]["roofline"] = {"bound": "hbm", "achieved": g64, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g64 / HBM_PEAK_GBS,
                                               "algorithmic_bytes_per_eval": b64,
                                               "note": "whole float64 step on the SURVEY 8d bytes (the north-star >= 40 % figure, like-for-like with the float64 CPU baseline)"}
            if side_table:
                line["float64"]["all_kernels_ms_per_step"] = {"%s[%s]" % k: round(v["mean_ms"] * v["launches"] / SIDE_TABLE_STEPS, 4)
                                                              for k, v in sorted(side_table.items())}
                i8 = [(k, v) for k, v in side_table.items() if k[0] in ("jf_mlp2_i8_f64", "jf_mlp2_i8_seg_f64")]
                if i8:
                    # the wide amortisation MLP of the float64 step on the int8 matrix cores (csrc/mlp_i8_kernels.hip): executed integer
                    # multiply-adds = slice pairs x 2 x rows x 128 hidden units x output columns padded to 16-column tiles
                    (k, v), = i8[:1]
                    n_out = int(k[1].split("_N")[1].split("_")[0]); slices = int(k[1].split("_x")[1])
                    pairs = slices * (slices + 1) // 2
                    ops = pairs * 2.0 * B * 128 * ((n_out + 15) // 16 * 16)
                    sec = v["mean_ms"] * 1e-3
                    line["float64"]["mlp_i8"] = {"bound": "mfma", "achieved": ops / sec / 1e12, "peak": 5000.0, "unit": "TOP/s (int8, executed)",
                                                 "frac": ops / sec / 1e12 / 5000.0, "mean_launch_ms": v["mean_ms"],
                                                 "arithmetic": "%d int8 digit slices per operand, %d slice-pair products, exact int32 accumulation" % (slices, pairs),
                                                 "float64_equivalent_TFLOPs": 2.0 * B * 128 * n_out / sec / 1e12,
                                                 "f64_mfma_peak_TFLOPs": 78.6}
        if "float64" in line and side_table:
            f64_issue = valu_issue_roofline(args.workload, "f64", B, side_table) or float64_issue_roofline(args.workload, B, side_table)
            if f64_issue:
                line["float64"]["roofline_f64_issue"] = f64_issue
        if rm.get("rows_sweep") and "one_stream_ms_per_step" in rm["rows_sweep"][0] and (1 << rm["rows_sweep"][0]["log2_rows"]) == B:
            # the drop-in call -- pdf(x) step after step on the caller's ONE stream, through its recorded plan -- next to the headline, which needs
            # the non-reference pipelined_forward().submit() API (consecutive steps on alternating streams)
            os_ms = rm["rows_sweep"][0]["one_stream_ms_per_step"]
            line["one_stream"] = {"ms_per_step": os_ms, "value": B / (os_ms * 1e-3), "unit": "log-prob evals/s",
                                  "what": "pdf(x) on one stream (the reference's call pattern), measured after the timed region (rows sweep, 3 x 50 steps)",
                                  "headline_api": "pdf.pipelined_forward(x).submit(x): %d streams" % args.pipeline_depth}
        if rm.get("rows_sweep"):
            sw = rm["rows_sweep"]
            line["rows_sweep"] = {"what": "step time of this workload against the batch size on this ONE GPU (prefixes of the resident inputs, each size "
                                          "through its own recorded plan), measured after the timed region",
                                  "sizes": [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()} for r in sw]}
            by = {r["log2_rows"]: r for r in sw}
            top = sw[0]["log2_rows"]
            if top - 3 in by:
                # T_1 = the TIMED step of this line where the sweep's own full-size entry is slower than it (the sweep runs 3 x 50 steps per size
                # after the other post-timing work and has read 0.73 ms against a timed 0.66): the smaller T_1 gives the smaller, honest efficiency
                t1 = min(by[top]["ms_per_step"], rm["ms_per_step"]) if (1 << top) == B else by[top]["ms_per_step"]
                line["rows_sweep"]["predicted_8gpu_strong_scaling_efficiency"] = t1 / (8 * by[top - 3]["ms_per_step"])
                line["rows_sweep"]["predicted_8gpu_strong_scaling_efficiency_T1_ms"] = t1
                if "one_stream_ms_per_step" in by[top]:
                    line["rows_sweep"]["predicted_8gpu_strong_scaling_efficiency_one_stream"] = (by[top]["one_stream_ms_per_step"] /
                                                                                                 (8 * by[top - 3]["one_stream_ms_per_step"]))
                line["rows_sweep"]["note"] = ("T_1 / (8 T_8) with T_8 = this GPU's time on an eighth of the batch: compute only, the per-step log-prob "
                                              "all-gather (512 KiB per rank) comes on top; `with_exchange`: the same shard step through the N > 1 path "
                                              "of this script (a one-rank RCCL process group in a child process, JF_FORCE_COLLECTIVES=1: the exchange's "
                                              "host and device cost without the cross-GPU wait)")
                if world == 1 and not multi:
                    # the shard step as a rank of the 8-GPU run executes it -- a fresh process that runs nothing else (the in-process sweep above
                    # reads ~10 % more at 2^17 rows: it follows every other size's plans and streams in this process) --, with and without the exchange
                    rs = line["rows_sweep"]
                    rs["in_process_sweep_predicted_8gpu_strong_scaling_efficiency"] = rs["predicted_8gpu_strong_scaling_efficiency"]
                    rs["with_exchange"] = shard_with_exchange(args.workload, (1 << top) // 8, args.gather_steps, t1)
                    rs["compute_only_fresh_process"] = shard_with_exchange(args.workload, (1 << top) // 8, args.gather_steps, t1, exchange=False)
                    if "predicted_8gpu_strong_scaling_efficiency" in rs["with_exchange"]:
                        rs["predicted_8gpu_strong_scaling_efficiency"] = rs["with_exchange"]["predicted_8gpu_strong_scaling_efficiency"]
                        rs["predicted_8gpu_strong_scaling_efficiency_source"] = "with_exchange (T_1 / 8 T(2^17 rows through the N > 1 path))"
        if rank == 0 and world == 1 and not args.no_sweep:
            # the other BASELINE configurations, after the timed region (C3 float64 is the `float64` object above)
            table = {}
            for key in ("c1", "c2", "c3", "c3b", "c4", "c5"):
                if key == args.workload:
                    table[key] = {"workload": 'pdf("%s","%s")' % W["defs"], "dtype": main_dt, "rows": B, "ms_per_step": rm["ms_per_step"], "evals_per_s": rm["evals_per_s"],
                                  "max_abs_dlogp_vs_f64_oracle": rm["err"], "bar": 1e-2 if main_dt == "f32" else 1e-4,
                                  "whole_step_hbm_frac": roofline["whole_step"]["frac"], "note": "the timed run of this line"}
                    continue
                try:
                    table[key] = side_config(key, dev)
                except Exception as e:                     # noqa: BLE001 -- reported, never hidden
                    table[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            line["configs"] = table
        if rank == 0 and world == 1 and not args.no_sweep and args.workload in ("c3", "c3b", "c5"):
            # the other two directions of this configuration (training step, sampling step): child runs of this script after the timed region,
            # so that the driver's record of the default command carries them too (their own lines: --train / --direction sample)
            line["other_directions"] = other_directions_summary(args.workload)
        if graph_replay is not None:
            line["hip_graph_replay"] = graph_replay
        if two_launch is not None:
            tb = two_launch["table"]
            ml = tb.get(("jf_mlp2_f32", "K7_H128_N548"))
            gf = tb.get(("jf_gf_chain_inv_f32", "per-sample"))
            blk = {"ms_per_step": 1e3 * two_launch["dt"] / args.steps, "value": B * args.steps / two_launch["dt"],
                   "note": "same steps with the conditional block as jf_mlp2 + jf_gf_chain_inv (measured after the timed region, this rank only)"}
            if ml is not None:
                tf = 2 * (7 * 128 + 128 * 548) * B / (ml["mean_ms"] * 1e-3) / 1e12
                blk["mlp2"] = {"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                               "mean_launch_ms": ml["mean_ms"]}
            sp = tb.get(("jf_linear_split_f32", "K128_N548"))
            if sp is not None:                         # the wide second layer on split-bf16 MFMA (first layer: a streaming jf_linear launch)
                tf = 6 * 2 * 128 * 548 * B / (sp["mean_ms"] * 1e-3) / 1e12
                blk["linear_split"] = {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0,
                                       "mean_launch_ms": sp["mean_ms"], "arithmetic": "3-way split bf16, 6 MFMA passes, f32 accumulate (executed bf16 flop)"}
                blk["note"] = "same steps with the conditional block as jf_linear + jf_linear_split + jf_gf_chain_inv (measured after the timed region, this rank only)"
            if gf is not None:
                gb = 4 * 558 * B / (gf["mean_ms"] * 1e-3) / 1e9
                blk["gf_chain_per_sample"] = {"bound": "hbm", "achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS,
                                              "mean_launch_ms": gf["mean_ms"], "algorithmic_bytes_per_launch": 4 * 558 * B}
            line["two_launch_path"] = blk
        emit_line(line)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
