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

if os.environ.get("JF_BENCH_HANG_TRACE"):        # debugging aid: the Python stacks of all threads after this many seconds, then exit
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ["JF_BENCH_HANG_TRACE"]), exit=True)

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

from benchlib.workloads import *          # noqa: E402,F401,F403 -- peaks, WORKLOADS, REFERENCE_8THREAD, make_inputs
from benchlib.workloads import WORKLOADS, make_inputs  # noqa: E402
from benchlib.cpu import cpu_baseline, cpu_model, cpu_quota, oracle_rows  # noqa: E402,F401
from benchlib.pmc import (KERNEL_OF, N_SIMDS, SIDE_TABLE_STEPS, WRITE_CAL, committed_traffic, float64_issue_roofline, kernel_accounting,  # noqa: E402,F401
                          kernel_source_hash, measure_traffic, traffic_of, valu_issue_roofline)
from benchlib.directions import other_direction, train_parity  # noqa: E402,F401
from benchlib.launcher import dry_run, emit_line, isolate_stdout, launch_ranks  # noqa: E402,F401
from benchlib.sweeps import other_directions_summary, rows_sweep, shard_with_exchange, side_config  # noqa: E402,F401


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
                                 # (the contract's enumeration: "hbm" | "mfma".  `achieved` counts the flops EXECUTED on the matrix pipe -- three f16
                                 #  passes per product --, `algorithmic_TFLOPs` below the reference's product; what binds the launch beside the matrix
                                 #  pipe is the vector issue of the flow arithmetic: `valu_issue`)
                                 "bound": "mfma", "bound_detail": "vector issue of the flow arithmetic + the matrix pipe, mostly one after the other "
                                                                  "(profiles/r06_experiments.md section 1): see valu_issue",
                                 "achieved": ex, "peak": pk, "unit": "TFLOP/s", "flops_counted": "executed on the matrix pipe", "frac": ex / pk,
                                 "valu_busy_frac_committed_profile": {"c3": 0.76, "c3b": 0.77, "c5": 0.60}.get(args.workload),
                                 "mfma_busy_frac_committed_profile": {"c3": 0.42, "c3b": 0.41, "c5": 0.16}.get(args.workload)})
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
