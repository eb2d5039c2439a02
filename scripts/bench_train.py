#!/usr/bin/env python3
"""Training-step timing: forward + backward (+ Adam) of -mean(log p) through the backward kernels, per-kernel breakdown.
    python scripts/bench_train.py [--rows N] [--workload c3|c5]
Not the contract benchmark (bench.py is); this feeds the backward table of DESIGN.md."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import numpy as np
import torch
import fixture_io
import helpers
from jammy_flows_amd import _hip

def inputs(fx, n, seed):
    rng = np.random.default_rng(seed)
    cols = []
    for part in fx.pdf_defs.split("+"):
        kind, dim = part[0], int(part[1:].split("_")[0])
        if kind == "e":
            cols.append(rng.normal(size=(n, dim)) * 1.5)
        elif kind == "i":
            cols.append(rng.uniform(1e-6, 1 - 1e-6, size=(n, 1)))
        elif dim == 1:
            cols.append(rng.uniform(0, 2 * np.pi, size=(n, 1)))
        else:
            cols.append(np.arccos(rng.uniform(-1, 1, size=(n, 1))).clip(1e-3, np.pi - 1e-3))
            cols.append(rng.uniform(0, 2 * np.pi, size=(n, 1)))
    x = np.concatenate(cols, axis=1)
    c = fx.get("cond")
    cond = rng.normal(size=(n, c.shape[1])) if c is not None else None
    return x, cond


ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1 << 18)
ap.add_argument("--workload", default="c3")
ap.add_argument("--torch-profile", action="store_true", help="also print the device-time table of torch.profiler for one step (library GEMMs, elementwise glue)")
ap.add_argument("--graph", action="store_true", help="capture the whole step (forward, backward, Adam) in one HIP graph and time its replays")
ap.add_argument("--fused-adam", action="store_true", help="torch.optim.Adam(fused=True): one optimizer launch instead of ~8")
ap.add_argument("--pmc-child", action="store_true", help="a few steps only (under rocprofv3 --pmc: every launch is serialised and slow)")
args = ap.parse_args()
name, dtype = ("c3_e4s2e4", torch.float32) if args.workload == "c3" else ("c5_e8s2_ggggv", torch.float64)
fx = fixture_io.load(name)
pdf = helpers.build_product(fx, dtype)
x, cond = inputs(fx, args.rows, 7)
x = torch.from_numpy(x).to(device="cuda", dtype=dtype)
cond = torch.from_numpy(cond).to(device="cuda", dtype=dtype) if cond is not None else None
opt = torch.optim.Adam(pdf.parameters(), lr=1e-4, capturable=args.graph, fused=args.fused_adam or None)
if args.graph:
    pdf.check_status = False                          # reading the status words is a host synchronisation: not capturable


def step():
    opt.zero_grad(set_to_none=True)
    logp, _, _ = pdf(x, conditional_input=cond)
    loss = -logp.mean()
    loss.backward()
    opt.step()
    return loss


if args.graph:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(5):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        static_loss = step()
    eager_step = step

    def step():
        graph.replay()
        return static_loss
for _ in range(5):
    step()
if args.pmc_child:
    torch.cuda.synchronize()
    sys.exit(0)
n = 20
reps = []
for _ in range(5):                                   # the step issues ~100 launches: host jitter shows, so 5 repetitions of 20 steps, median reported
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    reps.append((time.perf_counter() - t0) / n)
dt = sorted(reps)[len(reps) // 2]
with torch.no_grad():
    for _ in range(3):
        pdf(x, conditional_input=cond)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        pdf(x, conditional_input=cond)
    torch.cuda.synchronize()
    dt_fwd = (time.perf_counter() - t0) / n
timer = _hip.KernelTimer()
with timer:
    step()
print("workload %s dtype %s rows %d: training step %.3f ms (%.3g rows/s; median of 5 x 20 steps, best %.3f, worst %.3f), no-grad forward %.3f ms, loss %.4f" % (
    args.workload, dtype, args.rows, 1e3 * dt, args.rows / dt, 1e3 * min(reps), 1e3 * max(reps), 1e3 * dt_fwd, float(loss)))
for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
    print("  %-44s x%d  %.3f ms" % (k[0] + "[" + k[1] + "]", v["launches"], v["total_ms"]))
if args.torch_profile:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
