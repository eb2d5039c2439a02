"""RCCL on the GPU box (run with -m gpu).  The box has ONE GPU, and RCCL refuses two ranks on one device, so the collectives of the row-sharded
path (jammy_flows_amd/parallel.py) are exercised here through a process group of a single rank with
JF_FORCE_COLLECTIVES=1: group set-up on the device, all_gather_into_tensor submitted on a pipelined step's own stream while the next step
runs, the flat gradient all-reduce, the barrier of the contract's timing loop and the exchange report.  What more than one rank adds -- the row
arithmetic of the shards and the ordering of the gathered blocks -- is covered by the world-size-2 gloo tests (tests/test_parallel_cpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

import fixture_io
from helpers import build_product, to_dev

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.fixture()
def one_rank_rccl(monkeypatch):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    monkeypatch.setenv("JF_FORCE_COLLECTIVES", "1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        yield dev
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("direct", ["1", "0"], ids=["rccl-direct", "torch.distributed"])
def test_pipelined_steps_with_gather_on_the_step_stream(one_rank_rccl, monkeypatch, direct):
    """both carriers of the exchange: ncclAllGather through ctypes on a communication stream (jammy_flows_amd/rccl.py, the default) and
    torch.distributed.all_gather_into_tensor (JF_RCCL_DIRECT=0, also the fallback)"""
    from jammy_flows_amd import parallel
    monkeypatch.setenv("JF_RCCL_DIRECT", direct)
    dev = one_rank_rccl
    assert dist.get_backend() == "nccl" and parallel.collectives_active()
    fx = fixture_io.load("c3_e4s2e4")
    pdf = build_product(fx, torch.float32)
    reps = (1 << 15) // fx["x"].shape[0]
    x = to_dev(np.tile(fx["x"], (reps, 1)), torch.float32)
    B = x.shape[0]
    with torch.no_grad():
        want = pdf(x)[0].clone()
        pipe = pdf.pipelined_forward(x, depth=2)
        gather = parallel.PipelinedGather(B, torch.float32, dev)
        assert gather.collective and gather.path == ("rccl-direct" if direct == "1" else "torch.distributed"), (gather.path, gather.direct_error)
        outs = []
        for _ in range(6):
            t = pipe.submit(x)
            with torch.cuda.stream(t.stream):
                outs.append(gather.submit(t.outputs[0]))
        pipe.drain()
        full = gather.wait()
        torch.cuda.synchronize()
    assert torch.equal(full, want)
    assert all(torch.equal(o, want) for o in outs[-2:])
    gather.close()
    # fewer, larger collectives: three pipelined steps (two streams) per all-gather, seven steps = two exchanges + a partly filled stage
    with torch.no_grad():
        gk = parallel.PipelinedGather(B, torch.float32, dev, group_steps=3)
        xs = [x, torch.flip(x, dims=(0,)).contiguous()]
        wants = [want, pdf(xs[1])[0].clone()]
        for i in range(7):
            t = pipe.submit(xs[i % 2])
            with torch.cuda.stream(t.stream):
                gk.submit(t.outputs[0])
        pipe.drain()
        full = gk.wait()
        torch.cuda.synchronize()
    assert tuple(full.shape) == (1, 3, B) and torch.equal(gk.last_block(0), wants[6 % 2]) and torch.equal(full[0, 0], wants[6 % 2])
    other = gk.out[gk.i % 2]
    assert all(torch.equal(other[0, t], wants[(3 + t) % 2]) for t in range(3))
    gk.close()
    # zero-copy staging (what bench.py does for N > 1): the step writes its log-probs into the stage slot, the exchange waits for the steps' events
    for k in (1, 3):
        with torch.no_grad():
            gz = parallel.PipelinedGather(B, torch.float32, dev, group_steps=k)
            for i in range(16 if k == 3 else 7):            # (k = 3: more than two stage cycles, so that the re-use guard on the step streams runs)
                t = pipe.submit(xs[i % 2], logp_out=gz.next_slot(pipe.peek_stream() if k == 3 else None))
                gz.staged(t)
            pipe.drain()
            full = gz.wait()
            torch.cuda.synchronize()
        n = 16 if k == 3 else 7
        assert torch.equal(gz.last_block(0), wants[(n - 1) % 2]), k
        prev = gz.out[gz.i % 2]
        if k == 3:                                           # 16 steps = 5 full exchanges + one step: the previous buffer holds steps 12, 13, 14
            assert all(torch.equal(prev[0, t], wants[(12 + t) % 2]) for t in range(3))
        else:
            assert torch.equal(prev, wants[5 % 2])
        gz.close()
    rep = parallel.gather_report(B, torch.float32, dev)
    assert rep["rows_per_rank"] == [B] and rep["gather_us"] is not None and rep["gather_us"] > 0.0


def test_all_gather_rows_allreduce_and_timing_loop(one_rank_rccl):
    from jammy_flows_amd import parallel
    dev = one_rank_rccl
    fx = fixture_io.load("g_e3_ggg_cond")
    pdf = build_product(fx, torch.float64)
    x, c = to_dev(fx["x"][:128], torch.float64), to_dev(fx["cond"][:128], torch.float64)
    with torch.no_grad():
        local = pdf(x, conditional_input=c)[0]
    got = parallel.sharded_log_prob(pdf, x, conditional_input=c)
    assert torch.equal(got, local)
    with torch.enable_grad():
        (-pdf(x, conditional_input=c)[0].mean()).backward()
    before = [p.grad.clone() for p in pdf.parameters() if p.grad is not None]
    n = parallel.allreduce_gradients(pdf.parameters(), average=True)
    after = [p.grad for p in pdf.parameters() if p.grad is not None]
    assert n == sum(b.numel() for b in before)
    assert all(torch.allclose(a, b, rtol=0, atol=0) for a, b in zip(after, before))
    calls = []
    dt = parallel.timed_steps(lambda: calls.append(pdf(x, conditional_input=c)[0]), steps=3, warmup=1, device=dev)
    assert len(calls) == 4 and dt > 0.0
