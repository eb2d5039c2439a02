"""world_size-2 gloo tests (CPU) of the row-sharding helpers used by the multi-GPU path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jammy_flows_amd import parallel


def test_shard_bounds_cover_all_rows():
    for n in (0, 1, 7, 8, 1 << 20, (1 << 20) + 3):
        for w in (1, 2, 4, 8):
            b = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, q):
    torch.set_num_threads(1)                 # the test host runs several of these process pairs at once (pytest -n): no thread oversubscription
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        x = torch.randn(n_rows, 3, dtype=torch.float64)
        cond = torch.randn(n_rows, 2, dtype=torch.float64)

        def evaluate(xs, cs):       # stand-in for the HIP log-prob (no GPU here): any row-wise function of the shard
            return (xs ** 2).sum(dim=1) + cs[:, 0]

        full = parallel.sharded_log_prob(None, x, cond, gather=True, evaluate=evaluate)
        expect = (x ** 2).sum(dim=1) + cond[:, 0]
        ok = bool(torch.equal(full, expect))
        local = parallel.sharded_log_prob(None, x, cond, gather=False, evaluate=evaluate)
        lo, hi = parallel.shard_bounds(n_rows, rank, world)
        ok = ok and local.shape[0] == hi - lo and bool(torch.equal(local, expect[lo:hi]))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [64, 65, 1])
def test_sharded_log_prob_gloo_world2(n_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == [0, 1]
    assert all(ok for _, ok in res)


def _pipe_worker(rank, world, port, q):
    torch.set_num_threads(1)                 # the test host runs several of these process pairs at once (pytest -n): no thread oversubscription
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 33
        pg = parallel.PipelinedGather(n, torch.float64, torch.device("cpu"), depth=2)
        ok = True
        bufs = []
        for step in range(5):                               # more steps than buffers: slots are reused only after their gather finished
            local = torch.full((n,), float(10 * step + rank), dtype=torch.float64)
            bufs.append((step, pg.submit(local)))
        last = pg.wait()
        expect = torch.cat([torch.full((n,), float(10 * 4 + r), dtype=torch.float64) for r in range(world)])
        ok = ok and bool(torch.equal(last, expect))
        # slot of step 3 (the other buffer) holds step 3's gather
        expect3 = torch.cat([torch.full((n,), float(10 * 3 + r), dtype=torch.float64) for r in range(world)])
        ok = ok and bool(torch.equal(bufs[3][1], expect3))
        # fewer, larger collectives: three steps per all-gather, 11 steps = three full exchanges + a partly filled stage that wait() exchanges
        pk = parallel.PipelinedGather(n, torch.float64, torch.device("cpu"), depth=2, group_steps=3)
        for step in range(11):
            pk.submit(torch.full((n,), float(100 * step + rank), dtype=torch.float64))
        full = pk.wait()
        ok = ok and tuple(full.shape) == (world, 3, n)
        for r in range(world):                               # the last exchange holds steps 9, 10 (and, in its third slot, step 5 of the stage's previous use)
            ok = ok and bool((full[r, 0] == 100 * 9 + r).all()) and bool((full[r, 1] == 100 * 10 + r).all()) and bool((full[r, 2] == 100 * 5 + r).all())
            ok = ok and bool(torch.equal(pk.last_block(r), torch.full((n,), float(100 * 10 + r), dtype=torch.float64)))
        prev = pk.out[pk.i % 2]                              # the other output buffer: steps 6, 7, 8
        for r in range(world):
            ok = ok and all(bool((prev[r, t] == 100 * (6 + t) + r).all()) for t in range(3))
        # zero-copy staging (what bench.py's N > 1 steps do): the producer writes into next_slot(), staged() counts it; same layout across ranks

        class _Step:
            event = None
        for k in (1, 4):
            pz = parallel.PipelinedGather(n, torch.float64, torch.device("cpu"), depth=2, group_steps=k)
            for step in range(9):
                pz.next_slot().fill_(float(1000 * step + rank))
                pz.staged(_Step())
            full = pz.wait()
            for r in range(world):
                ok = ok and bool((pz.last_block(r) == 1000 * 8 + r).all())
            if k == 4:                                        # 9 steps = two full exchanges + step 8 alone in the third (slots 1..3: steps 1..3 of the stage's first use)
                ok = ok and tuple(full.shape) == (world, 4, n)
                ok = ok and all(bool((full[r, 0] == 8000 + r).all()) and bool((full[r, 1] == 1000 + r).all()) for r in range(world))
                ok = ok and all(bool((pz.out[pz.i % 2][r, t] == 1000 * (4 + t) + r).all()) for r in range(world) for t in range(4))
        # wait() in the middle of a stage, then more submissions (last_block indexed the stage with n_submitted % k, which is wrong
        # once a partly filled stage has been flushed): k = 4, 2 submits, wait(), 4 more -> the last rows sit in stage row 3
        pw = parallel.PipelinedGather(n, torch.float64, torch.device("cpu"), depth=2, group_steps=4)
        for step in range(2):
            pw.submit(torch.full((n,), float(7000 + 10 * step + rank), dtype=torch.float64))
        pw.wait()
        for r in range(world):
            ok = ok and bool((pw.last_block(r) == 7000 + 10 * 1 + r).all())
        for step in range(2, 6):
            pw.submit(torch.full((n,), float(7000 + 10 * step + rank), dtype=torch.float64))
        pw.wait()
        for r in range(world):
            ok = ok and bool((pw.last_block(r) == 7000 + 10 * 5 + r).all())
        # ... and the same through the zero-copy staging calls, with a partly filled stage flushed at the end
        pw = parallel.PipelinedGather(n, torch.float64, torch.device("cpu"), depth=2, group_steps=4)
        for step in range(7):
            pw.next_slot().fill_(float(9000 + 10 * step + rank))
            pw.staged(_Step())
            if step == 1:
                pw.wait()
        pw.wait()
        for r in range(world):
            ok = ok and bool((pw.last_block(r) == 9000 + 10 * 6 + r).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == [0, 1]
    assert all(ok for _, ok in res)


def test_pipelined_gather_single_process():
    pg = parallel.PipelinedGather(4, torch.float32, torch.device("cpu"))
    a = pg.submit(torch.arange(4, dtype=torch.float32))
    assert torch.equal(pg.wait(), torch.arange(4, dtype=torch.float32)) and a.shape[0] == 4


def test_hw_queue_setting_is_made_by_the_library(monkeypatch):
    """parallel sets GPU_MAX_HW_QUEUES while the runtime has not started (here: never, no GPU); an explicit larger value is kept"""
    assert parallel.HW_QUEUES_STATE in ("set", "kept") and int(os.environ["GPU_MAX_HW_QUEUES"]) >= parallel.HW_QUEUES_WANTED
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "16")
    assert parallel.ensure_hw_queues() == "kept" and os.environ["GPU_MAX_HW_QUEUES"] == "16"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "2")
    assert parallel.ensure_hw_queues() == "set" and os.environ["GPU_MAX_HW_QUEUES"] == str(parallel.HW_QUEUES_WANTED)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "2")
    with pytest.warns(RuntimeWarning, match="GPU_MAX_HW_QUEUES"):
        assert parallel.ensure_hw_queues() == "late"


def test_rank_time_stats_single_process():
    info = {}
    parallel.timed_steps(lambda: None, steps=3, warmup=1, info=info)
    st = parallel.rank_time_stats(info, 3)
    assert info["n_ranks_seen"] == 1 and st["slowest_rank"] == 0 and st["min"] == st["max"] == st["mean"]


def _loop_worker(rank, world, port, q):
    torch.set_num_threads(1)                 # the test host runs several of these process pairs at once (pytest -n): no thread oversubscription
    """the bench.py step loop (parallel.timed_steps + PipelinedGather, weak and strong row sharding) with a stand-in evaluate, world 2 on gloo"""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for scaling, total in (("weak", None), ("strong", 130)):
            if scaling == "weak":
                n = 64
            else:
                lo, hi = parallel.shard_bounds(total, rank, world)
                n = hi - lo
            x = torch.arange(n, dtype=torch.float64) + 1000 * rank
            gather = parallel.PipelinedGather(n, torch.float64, torch.device("cpu"))
            calls = {"step": 0, "finish": 0}

            def step():
                calls["step"] += 1
                if rank == 1:
                    time.sleep(0.01)                      # the slow rank sets the job's time (MAX over ranks)
                gather.submit(x * 2.0)

            def finish():
                calls["finish"] += 1
                gather.wait()

            info = {}
            dt = parallel.timed_steps(step, steps=5, warmup=2, finish=finish, device=None, info=info)
            ok = ok and calls["step"] == 7 and calls["finish"] == 2 and dt >= 0.05
            # the per-rank clocks of the N > 1 line: both ranks seen, the job's time is the slowest rank's, the same numbers on every rank
            st = parallel.rank_time_stats(info, 5)
            ok = ok and info["n_ranks_seen"] == world and len(st["per_rank"]) == world and abs(st["max"] * 5e-3 - dt) < 1e-12
            ok = ok and st["min"] <= st["mean"] <= st["max"] and st["per_rank"][st["slowest_rank"]] == st["max"] and st["max"] >= 10.0
            full = gather.wait()
            expect = torch.cat([(torch.arange(n, dtype=torch.float64) + 1000 * r) * 2.0 for r in range(world)])
            ok = ok and bool(torch.equal(full, expect))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_bench_step_loop_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == [0, 1]
    assert all(ok for _, ok in res)


def _grad_worker(rank, world, port, q):
    torch.set_num_threads(1)                 # the test host runs several of these process pairs at once (pytest -n): no thread oversubscription
    """row-sharded training step: local backward on the shard + ONE gradient all-reduce == the gradient of the global-batch mean loss"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(3, 8), torch.nn.Tanh(), torch.nn.Linear(8, 1)).double()
        extra = torch.nn.Parameter(torch.ones(2, dtype=torch.float64))       # touched by rank 0 only
        unused = torch.nn.Parameter(torch.ones(2, dtype=torch.float64))      # touched by nobody
        x = torch.randn(64, 3, dtype=torch.float64)
        ref = torch.nn.Sequential(torch.nn.Linear(3, 8), torch.nn.Tanh(), torch.nn.Linear(8, 1)).double()
        ref.load_state_dict(net.state_dict())
        ref(x).mean().backward()
        lo, hi = parallel.shard_bounds(64, rank, world)
        loss = net(x[lo:hi]).mean()
        if rank == 0:
            loss = loss + extra.sum()
        loss.backward()
        params = list(net.parameters()) + [extra, unused]
        n = parallel.allreduce_gradients(params, average=True)
        ok = n == sum(p.numel() for p in params)
        for p, r in zip(net.parameters(), ref.parameters()):
            ok = ok and bool(torch.allclose(p.grad, r.grad, rtol=1e-12, atol=1e-14))
        ok = ok and bool(torch.allclose(extra.grad, torch.full((2,), 0.5, dtype=torch.float64))) and unused.grad is None
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == [0, 1]
    assert all(ok for _, ok in res)


def _run_bench(args, env_extra=None, drop=()):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=300)


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` without a torch.distributed.run environment starts the ranks itself (child process), rank 0 prints ONE JSON line"""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--batch", "1000", "--scaling", "weak"], {"JF_BENCH_BACKEND": "gloo"},
                   drop=("WORLD_SIZE", "RANK", "LOCAL_RANK"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert [l for l in r.stdout.splitlines() if l.strip()] == lines, "stdout carries the JSON line and nothing else (bench.isolate_stdout)"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["collective_backend"] == "gloo"
    assert line["dry_run"] is True and line["gathered_rows_correct"] is True
    assert line["config"]["total_rows"] == 2000 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    # the line says who held how many rows and what one stand-alone all-gather of the log-probs takes
    assert line["exchange"]["rows_per_rank"] == [1000, 1000] and line["exchange"]["gather_us"] > 0 and line["exchange"]["gather_bytes_per_rank"] == 8000
    # the default is STRONG scaling (efficiency at fixed total batch): --batch is then the total, split over the ranks
    # every rank's own step time (a straggler would show), counted from the ranks the timing loop really saw
    rk = line["rank_ms_per_step"]
    assert line["ranks_in_timing"] == 2 and len(rk["per_rank"]) == 2 and rk["min"] <= rk["mean"] <= rk["max"] and rk["slowest_rank"] in (0, 1)
    assert abs(rk["max"] - line["ms_per_step"]) < 1e-9 and line["scaling_efficiency_vs_t1"] is None
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run", "--batch", "1000", "--t1-ms", "3.0"], {"JF_BENCH_BACKEND": "gloo"},
                   drop=("WORLD_SIZE", "RANK", "LOCAL_RANK"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["scaling"] == "strong" and line["config"]["total_rows"] == 1000 and line["config"]["batch_per_gpu"] == 500
    # --t1-ms: the measured efficiency of this run against the driver's N = 1 time (strong: T_1 / (N T_N))
    eff = line["scaling_efficiency_vs_t1"]
    assert eff["n_gpus"] == 2 and eff["t1_ms"] == 3.0 and abs(eff["efficiency"] - 3.0 / (2 * line["ms_per_step"])) < 1e-9


def test_bench_c5_strong_scaling_row_split_dry_run():
    """BASELINE configs[4]: 2^22 rows in total; --gpus 8 weak = 2^19 per rank, strong = the same total split over the ranks (here 2 ranks)"""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run", "--workload", "c5", "--scaling", "strong", "--batch", "4097"],
                   {"JF_BENCH_BACKEND": "gloo"}, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["config"]["total_rows"] == 4097 and line["config"]["batch_per_gpu"] == 2049 and line["scaling"] == "strong"
    assert line["exchange"]["rows_per_rank"] == [2049, 2048] and line["exchange"]["gather_us"] is None      # unequal shards: nothing is gathered


def test_bench_refuses_a_rank_count_mismatch():
    r = _run_bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
