"""what the per-step exchange costs a 2^17-row shard step (C3 float32, two alternating streams), one-rank RCCL group (JF_FORCE_COLLECTIVES=1):
python3 scripts/probe/gather_cost.py [rows]
modes: none | copy (a device copy on the step's stream) | direct (PipelinedGather on ncclAllGather through ctypes, the default) | torch
(PipelinedGather on all_gather_into_tensor, async: JF_RCCL_DIRECT=0) | torch_sync (async_op=False) |
every4 (torch, every 4th step gathers 4 steps' rows at once)"""
import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
os.environ["JF_FORCE_COLLECTIVES"] = "1"
import numpy as np
import torch
import torch.distributed as dist
import fixture_io, helpers
from jammy_flows_amd import parallel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32, dev)
pdf.check_status = "deferred"
reps = B // (fx["x"].shape[0] - 8) + 1
x = torch.from_numpy(np.tile(fx["x"][:-8], (reps, 1))[:B]).to(device=dev, dtype=torch.float32)
torch.set_grad_enabled(False)
pipe = pdf.pipelined_forward(x, depth=2)


def run(mode, steps=400):
    os.environ["JF_RCCL_DIRECT"] = "1" if mode == "direct" else "0"
    gather = parallel.PipelinedGather(B, torch.float32, dev)
    big = parallel.PipelinedGather(4 * B, torch.float32, dev)
    outs = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(2)]
    stash = torch.empty(4 * B, dtype=torch.float32, device=dev)
    full = torch.empty(B, dtype=torch.float32, device=dev)

    def step(i):
        t = pipe.submit(x)
        if mode == "none":
            return
        with torch.cuda.stream(t.stream):
            if mode == "copy":
                outs[i % 2].copy_(t.outputs[0])
            elif mode in ("torch", "direct"):
                gather.submit(t.outputs[0])
            elif mode == "torch_sync":
                dist.all_gather_into_tensor(full, t.outputs[0])
            elif mode == "every4":
                stash[(i % 4) * B:(i % 4 + 1) * B].copy_(t.outputs[0])
                if i % 4 == 3:
                    big.submit(stash)
    for i in range(40):
        step(i)
    pipe.drain(); gather.wait(); big.wait(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    t_host = time.perf_counter() - t0
    pipe.drain(); gather.wait(); big.wait(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gather.close(); big.close()
    print("%-10s %.4f ms per step (host issue %.4f ms per step)" % (mode, 1e3 * dt / steps, 1e3 * t_host / steps), flush=True)


for _ in range(2):
    for m in ("none", "copy", "direct", "torch", "torch_sync", "every4"):
        run(m)
if os.environ.get("JF_PROBE_PROFILE"):
    import cProfile, pstats, io
    for m in ("none", "direct"):
        pr = cProfile.Profile()
        pr.enable()
        run(m, steps=2000)
        pr.disable()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(28)
        print(buf.getvalue()[:6000])
pdf.flush_status()
dist.destroy_process_group()
