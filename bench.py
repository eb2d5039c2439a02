#!/usr/bin/env python3
"""Benchmark of the jammy_flows hot path on MI355X (contract: see the task statement / DESIGN.md section "Measurement").

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

One step = one log-prob evaluation (`pdf.forward`) of a batch of 2^20 rows per GPU of BASELINE.json's metric configuration
`pdf("e4+s2+e4", "gggg+f+gggg")` ("n" of the upstream README = "f"): all three sub-pdfs, both amortisation MLPs, every layer.
Inputs are synthetic (seeded) and resident in HBM before the timed region; weights are the frozen golden-fixture state_dict
(tests/golden/c3_e4s2e4.npz, reference init with the MLP damping undone so parameter blocks really vary per row).
For N > 1 every rank evaluates its own 2^20 rows (weak scaling) and every step all-gathers its log-probs (ONE RCCL all_gather,
issued asynchronously so that it overlaps the next step; all of them are waited for inside the timed region).

Printed JSON line (rank 0): metric/value (whole-job evals/s), ms_per_step, plus
  roofline      dominant kernel: algorithmic bytes per launch / mean launch time from HIP events recorded in the timed region
  cpu_baseline  the numpy oracle (kind "port") on this box's host cores, bounded sample, rank 0 at N = 1 only
  parity        max |d log p| of the timed configuration against the float64 oracle on a 4096-row sample
  float64       the same workload evaluated in float64 (evals/s), for reference
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]

import numpy as np  # noqa: E402

WORKLOAD = "c3_e4s2e4"
PDF_DEFS, FLOW_DEFS = "e4+s2+e4", "gggg+f+gggg"
BATCH = 1 << 20
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-input MFMA peak (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md


def make_inputs(n, seed):
    """x = [N(0,1.5^2)^4, theta = acos(U(-1,1)) clamped to [1e-3, pi-1e-3], phi = U(0,2pi), N(0,1.5^2)^4]."""
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.normal(size=(n, 4)) * 1.5,
                           np.arccos(rng.uniform(-1, 1, size=(n, 1))).clip(1e-3, np.pi - 1e-3),
                           rng.uniform(0, 2 * np.pi, size=(n, 1)),
                           rng.normal(size=(n, 4)) * 1.5], axis=1)


# ---------------------------------------------------------------------------------------------- CPU baseline (oracle, multi-process)
_ORACLE = None


def _oracle_init():
    global _ORACLE
    import fixture_io
    import helpers
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    _ORACLE = helpers.build_oracle(fixture_io.load(WORKLOAD))


def _oracle_chunk(x):
    return _ORACLE.forward(x)[0]


def cpu_baseline(budget_s=20.0):
    """time the CPU oracle on a bounded sample of the same workload using every host core (process pool, forked BEFORE any GPU call)."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    workers = max(1, min(cores, 64))
    chunk = 2048
    ctx = mp.get_context("fork")
    with ctx.Pool(workers, initializer=_oracle_init) as pool:
        x = make_inputs(chunk * workers, 3)
        chunks = [x[i * chunk:(i + 1) * chunk] for i in range(workers)]
        t0 = time.time()
        pool.map(_oracle_chunk, chunks)            # warm-up + rate estimate
        est = time.time() - t0
        rounds = int(max(1, min(64, budget_s / max(est, 1e-3))))
        x = make_inputs(chunk * workers * rounds, 3)
        chunks = [x[i * chunk:(i + 1) * chunk] for i in range(workers * rounds)]
        t0 = time.time()
        pool.map(_oracle_chunk, chunks)
        dt = time.time() - t0
    n = chunk * workers * rounds
    return {"value": n / dt, "unit": "log-prob evals/s", "cores": workers, "kind": "port",
            "sample": "%d rows of %s (float64 numpy oracle, %d processes x %d-row chunks), %.1f s" % (n, WORKLOAD, workers, chunk, dt)}


# ---------------------------------------------------------------------------------------------- HBM traffic from the committed PMC passes
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_g_traffic.json")
TRAFFIC_KEYS = {("jf_cond_gf_chain_inv_f32", "K7_H128_N548_D4"): "jf::cond_gf_chain_kernel<float, 8>",
                ("jf_mlp2_f32", "K7_H128_N548"): "jf::mlp2_kernel<float, 4, 2, true, true>",
                ("jf_mlp2_f32", "K4_H128_N10"): "jf::mlp2_kernel<float, 4, 1, true, true>",
                ("jf_gf_chain_inv_f32", "per-sample"): "jf::gf_chain_kernel<float, 4, false, false>",
                ("jf_gf_chain_inv_f32", "bcast"): "jf::gf_chain_kernel<float, 4, true, false>"}


SQ_FILE = os.path.join(ROOT, "profiles", "r01_g_sq_counters.json")


def pmc_issue(kname, ktag, B):
    """share of the kernel's cycles in which the matrix pipe / the VALU were busy, from the committed SQ counter pass of this command
    (SQ_VALU_MFMA_BUSY_CYCLES, SQ_ACTIVE_INST_VALU [quad-cycles], GRBM_GUI_ACTIVE; 1024 SIMDs, 8 XCDs).  On CDNA4 f32 MFMA and VALU work
    do not co-issue on a SIMD (scripts/probe/coexec.hip), so their sum is the fraction of the kernel's issue floor that is reached."""
    prefix = TRAFFIC_KEYS.get((kname, ktag))
    try:
        table = json.load(open(SQ_FILE))
    except OSError:
        return None
    if prefix is None or B != BATCH:
        return None
    for key, v in table.items():
        if key.startswith(prefix) and v.get("GRBM_GUI_ACTIVE"):
            cyc = v["GRBM_GUI_ACTIVE"] / 8.0
            m = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc
            a = v.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / 1024.0 / cyc
            return {"mfma_busy_frac": m, "valu_busy_frac": a, "issue_frac": m + a, "source": "profiles/r01_g_sq_counters.json"}
    return None


def pmc_traffic(kname, ktag, B):
    """HBM bytes per launch of one kernel from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
    (profiles/r01_g_*; PMC collection needs its own rocprofv3 runs, so the figure is read from the committed summary, not measured live).
    Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE x 2 on gfx950 for wide coalesced reads; WRITE_SIZE x 0.965, calibrated on
    scripts/probe/wstore for the 16-byte lane-per-row tile stores these kernels use.  None when the profile was taken at another batch."""
    try:
        table = json.load(open(TRAFFIC_FILE))
    except OSError:
        return None
    prefix = TRAFFIC_KEYS.get((kname, ktag))
    if prefix is None or B != BATCH:
        return None
    for key, v in table.items():
        if key.startswith(prefix) and v.get("FETCH_SIZE_raw_KB") is not None and v.get("WRITE_SIZE_raw_KB") is not None:
            return {"hbm_bytes_per_launch": v["FETCH_SIZE_raw_KB"] * 1024 * 2 + v["WRITE_SIZE_raw_KB"] * 1024 * 0.965,
                    "read_bytes": v["FETCH_SIZE_raw_KB"] * 1024 * 2, "write_bytes": v["WRITE_SIZE_raw_KB"] * 1024 * 0.965,
                    "source": "profiles/r01_g_traffic.json (rocprofv3 --pmc, separate passes)"}
    return None


# ---------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="rows per GPU (default 2^20 = the BASELINE configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fuse", action="store_true", help="time the two-launch path (jf_mlp2 + jf_gf_chain_inv) instead of the fused conditional block")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run --nproc-per-node N" % (world, args.gpus)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()                      # before the GPU is touched (fork safety)

    import torch
    import torch.distributed as dist
    import fixture_io
    import helpers
    from jammy_flows_amd import _hip, parallel

    # one rank per GPU.  (JF_BENCH_BACKEND=gloo lets the multi-process logic be exercised on a box with fewer GPUs than ranks: the ranks
    # then share devices round-robin, which RCCL refuses; never used for reported numbers.)
    backend = os.environ.get("JF_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    fx = fixture_io.load(WORKLOAD)
    B = args.batch
    x64 = make_inputs(B, 3 + rank)
    results = {}
    kernel_table = None
    two_launch = None
    for dtype in (torch.float32, torch.float64):
        pdf = helpers.build_product(fx, dtype, dev)
        pdf.fuse_conditional_blocks = not args.no_fuse
        x = torch.from_numpy(x64).to(device=dev, dtype=dtype)
        # N > 1: the per-row log-probs of every step are all-gathered (RCCL), asynchronously, while the next step computes
        gather = parallel.PipelinedGather(B, dtype, dev) if world > 1 else None

        def step():
            logp = pdf(x)[0]
            if gather is not None:
                gather.submit(logp)
            return logp

        for _ in range(args.warmup):
            step()
        if gather is not None:
            gather.wait()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        timer = _hip.KernelTimer() if dtype == torch.float32 else None
        t0 = time.perf_counter()
        if timer is not None:
            with timer:
                for _ in range(args.steps):
                    logp = step()
        else:
            for _ in range(args.steps):
                logp = step()
        pdf.flush_status()                            # deferred kernel status words of the timed steps: raises if any row went wrong
        if gather is not None:
            gather.wait()                             # every step's gather has landed inside the timed region
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # parity of what was just timed, against the float64 oracle (rank 0, 4096 rows)
        err = None
        if rank == 0:
            n_chk = min(4096, B)
            o = helpers.build_oracle(fx).forward(x64[:n_chk])[0]
            err = float(np.max(np.abs(logp[:n_chk].double().cpu().numpy() - o)))
        results[dtype] = dict(dt=dt, evals_per_s=world * B * args.steps / dt, ms_per_step=1e3 * dt / args.steps, err=err)
        if timer is not None:
            kernel_table = timer.summary()
            if rank == 0 and pdf.fuse_conditional_blocks:
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
        del pdf, x

    if world > 1:
        dist.barrier()
    if rank == 0:
        r32, r64 = results[torch.float32], results[torch.float64]
        # ---- roofline of the dominant kernel (float32 run).  Algorithmic bytes per row (SURVEY 8d / DESIGN.md), float32:
        #   conditional g-chain (block 2):  x 4 + log_det 1 + params 548 + base 4 + log_det 1            = 558 scalars
        #   dense layers: inputs + outputs of each launch (K + N scalars)
        alg = {("jf_gf_chain_inv_f32", "per-sample"): 4 * 558, ("jf_gf_chain_inv_f32", "bcast"): 4 * 10,
               ("jf_f_chain_inv_f32", "per-sample"): 4 * (2 + 1 + 10 + 2 + 1)}
        dom = max(kernel_table.items(), key=lambda kv: kv[1]["total_ms"])
        (kname, ktag), kstat = dom
        secs = kstat["mean_ms"] * 1e-3
        flops_per_row = 0
        fused = False
        if kname.startswith("jf_linear"):
            K = int(ktag.split("_")[0][1:]); N = int(ktag.split("_")[1][1:])
            bytes_per_row = 4 * (K + N)
            flops_per_row = 2 * K * N
        elif kname.startswith("jf_cond_gf_chain"):
            K1, H, N, D = (int(t[1:]) for t in ktag.split("_"))
            # fused launch (MLP + g layers): SURVEY 8d "materialised" accounting = MLP (reads inputs, writes block) + flow (reads block);
            # the block itself never reaches HBM, so the real traffic is only 4 (K1 + 2 D + 2) bytes per row
            bytes_per_row = 4 * (K1 + N) + 4 * (D + 1 + N + D + 1)
            flops_per_row = 2 * (K1 * H + H * N)
            fused = True
        elif kname.startswith("jf_mlp2"):
            K1, H, N = (int(t[1:]) for t in ktag.split("_"))
            bytes_per_row = 4 * (K1 + N)                       # MLP reads its inputs, writes the parameter block
            flops_per_row = 2 * (K1 * H + H * N)
        else:
            bytes_per_row = alg.get((kname, ktag), 0)
        hbm_gbs = bytes_per_row * B / secs / 1e9
        # which roofline binds this kernel: time at the HBM peak vs time at the dense f32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md)
        t_hbm = bytes_per_row * B / (HBM_PEAK_GBS * 1e9)
        t_mfma = flops_per_row * B / (MFMA_F32_PEAK_TFLOPS * 1e12)
        if t_mfma > t_hbm:
            achieved = flops_per_row * B / secs / 1e12
            roofline = {"bound": "mfma", "kernel": "%s[%s]" % (kname, ktag), "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": None, "hbm_algorithmic_GBs": hbm_gbs,
                        "hbm_frac": hbm_gbs / HBM_PEAK_GBS, "algorithmic_flops_per_launch": flops_per_row * B}
        else:
            roofline = {"bound": "hbm", "kernel": "%s[%s]" % (kname, ktag), "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": None}
        if fused:
            roofline["fused"] = True
            roofline["note"] = ("amortisation MLP + its 4 g layers in one launch; the parameter block stays in LDS.  f32 MFMA and VALU work "
                                "do not co-issue on a CDNA4 SIMD (scripts/probe/coexec.hip), so this kernel's floor is MFMA time + VALU time")
        roofline["issue"] = pmc_issue(kname, ktag, B)
        tr = pmc_traffic(kname, ktag, B)
        roofline["traffic"] = tr["hbm_bytes_per_launch"] if tr else None
        roofline["traffic_detail"] = tr
        roofline.update({"mean_launch_ms": kstat["mean_ms"], "algorithmic_bytes_per_launch": bytes_per_row * B,
                         "all_kernels_ms_per_step": {"%s[%s]" % k: round(v["total_ms"] / args.steps, 4) for k, v in sorted(kernel_table.items())}})
        # the HBM-bound flow kernel the north star names (per-sample parameter blocks), reported alongside
        gfk = kernel_table.get(("jf_gf_chain_inv_f32", "per-sample"))      # only with --no-fuse
        if gfk is not None:
            g = 4 * 558 * B / (gfk["mean_ms"] * 1e-3) / 1e9
            roofline["gf_chain_per_sample"] = {"bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS,
                                               "mean_launch_ms": gfk["mean_ms"], "algorithmic_bytes_per_launch": 4 * 558 * B,
                                               "traffic": (pmc_traffic("jf_gf_chain_inv_f32", "per-sample", B) or {}).get("hbm_bytes_per_launch")}
        line = {
            "metric": "log-prob evals/sec (batch 2^20 per GPU), e4+s2+e4 / gggg+f+gggg",
            "value": r32["evals_per_s"], "unit": "log-prob evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r32["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (seeded; weights = frozen golden-fixture state_dict)",
            "config": {"workload": 'pdf("%s","%s") log-prob, batch %d rows per GPU, unconditional pdf with autoregressive conditioning'
                                   % (PDF_DEFS, FLOW_DEFS, B), "batch_per_gpu": B, "parallelism": "rows sharded over %d GPU(s)" % world},
            "parity": {"max_abs_dlogp_vs_f64_oracle": r32["err"], "bar": 1e-2, "rows_checked": min(4096, B)},
            "float64": {"value": r64["evals_per_s"], "ms_per_step": r64["ms_per_step"], "max_abs_dlogp_vs_f64_oracle": r64["err"], "bar": 1e-4},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if two_launch is not None:
            tb = two_launch["table"]
            ml = tb.get(("jf_mlp2_f32", "K7_H128_N548"))
            gf = tb.get(("jf_gf_chain_inv_f32", "per-sample"))
            blk = {"ms_per_step": 1e3 * two_launch["dt"] / args.steps, "value": B * args.steps / two_launch["dt"],
                   "note": "same steps with the conditional block as jf_mlp2 + jf_gf_chain_inv (measured after the timed region, this rank only)"}
            if ml is not None:
                tf = 2 * (7 * 128 + 128 * 548) * B / (ml["mean_ms"] * 1e-3) / 1e12
                blk["mlp2"] = {"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                               "mean_launch_ms": ml["mean_ms"], "traffic": (pmc_traffic("jf_mlp2_f32", "K7_H128_N548", B) or {}).get("hbm_bytes_per_launch")}
            if gf is not None:
                gb = 4 * 558 * B / (gf["mean_ms"] * 1e-3) / 1e9
                blk["gf_chain_per_sample"] = {"bound": "hbm", "achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS,
                                              "mean_launch_ms": gf["mean_ms"], "algorithmic_bytes_per_launch": 4 * 558 * B,
                                              "traffic": (pmc_traffic("jf_gf_chain_inv_f32", "per-sample", B) or {}).get("hbm_bytes_per_launch")}
            line["two_launch_path"] = blk
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
