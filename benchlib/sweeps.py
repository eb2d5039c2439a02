"""what the default command measures AFTER its timed region: the rows sweep, the shard step through the N > 1 path, the other directions and
the other BASELINE configurations"""
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
from .pmc import valu_issue_roofline

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
    cmd = [sys.executable, BENCH_PY, "--workload", workload, "--batch", str(rows), "--gather-steps", str(gather_steps), "--no-cpu-baseline",
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
        cmd = [sys.executable, BENCH_PY, "--workload", workload, "--scaling", "weak", "--no-cpu-baseline", "--no-pmc", "--steps", "20",
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
