"""self-launch of the ranks for N > 1, stdout isolation for the one JSON line, the GPU-less dry run of the multi-rank plumbing"""
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

from .workloads import ROOT, BENCH_PY, WORKLOADS

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
           "--master-port", str(free_port()), BENCH_PY] + list(argv)
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
