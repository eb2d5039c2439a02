"""the CPU baseline of the contract line: the numpy oracle (kind "port") on the box's host cores, one single-threaded process per core -- log-prob
and sampling directions.  Only bench.py's cpu_baseline / parity legs come here (the oracle is test infrastructure, never the product path)."""
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

from .workloads import ROOT, WORKLOADS, REFERENCE_8THREAD, make_inputs

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
