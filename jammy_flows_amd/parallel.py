"""Row sharding of a batch over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

The hot path is embarrassingly parallel over rows: every rank evaluates its own contiguous slice of rows with
replicated weights (< 1 MB); no data-path collective is needed.  The only exchange is the optional all-gather of the per-row
log-probabilities (4 MiB per rank at 2^20 float32 rows) -- one all_gather_into_tensor, never an all-reduce.
"""
import os

import torch
import torch.distributed as dist


def collectives_active(group=None):
    """True when the exchanges below really call the backend: an initialised process group of more than one rank -- or of ONE rank with
    JF_FORCE_COLLECTIVES=1, which is how the RCCL calls (group set-up on the device, all_gather_into_tensor on a step's own stream, the flat
    gradient all-reduce, the barrier of the timing loop) are exercised on a box that has a single GPU (tests/test_gpu_rccl.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("JF_FORCE_COLLECTIVES") == "1"


def shard_bounds(n_rows, rank, world_size):
    """[lo, hi) of the rows owned by `rank`: contiguous, sizes differ by at most one, empty shards allowed (ragged inputs)."""
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rows(t, rank=None, world_size=None):
    if t is None:
        return None
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    lo, hi = shard_bounds(t.shape[0], rank, world_size)
    return t[lo:hi]


def all_gather_rows(local, n_rows_total=None, group=None):
    """gather the per-rank row blocks (possibly of different length) into the full tensor, on every rank."""
    if not collectives_active(group):
        return local
    world = dist.get_world_size(group)
    if n_rows_total is None:
        n = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
        dist.all_reduce(n, group=group)
        n_rows_total = int(n.item())
    sizes = [shard_bounds(n_rows_total, r, world)[1] - shard_bounds(n_rows_total, r, world)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        out = torch.empty((n_rows_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged: pad to the largest shard, gather, trim
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)], dim=0)


def sharded_log_prob(pdf, x, conditional_input=None, gather=True, evaluate=None, **kwargs):
    """log-prob of the FULL batch x (same tensor on every rank): each rank evaluates its row shard, then (optionally) one all-gather.
    `evaluate(x_shard, cond_shard) -> (B_shard,)` defaults to pdf.log_prob."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_bounds(x.shape[0], rank, world)
    xs = x[lo:hi]
    cs = None if conditional_input is None else conditional_input[lo:hi]
    if evaluate is None:
        local = pdf.log_prob(xs, conditional_input=cs, **kwargs)
    else:
        local = evaluate(xs, cs)
    return all_gather_rows(local, x.shape[0]) if gather else local


class PipelinedGather:
    """all-gather of equal-sized per-rank row blocks that overlaps with the NEXT step's kernels: submit() enqueues the collective
    asynchronously (RCCL runs it on its own stream once the producer kernels of `local` are done) into one of `depth` rotating output
    buffers and returns immediately; the buffer of a submission is complete after the next submit() into the same slot or after wait().
    A step of the row-sharded hot path therefore never stalls on the 4 MiB log-prob exchange."""

    def __init__(self, n_rows_local, dtype, device, tail_shape=(), depth=2, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.collective = collectives_active(group)
        self.out = [torch.empty((self.world * n_rows_local,) + tuple(tail_shape), dtype=dtype, device=device) for _ in range(depth)]
        self.work = [None] * depth
        self.keep = [None] * depth          # the submitted tensors must outlive their collectives
        self.i = 0

    def submit(self, local):
        j = self.i % len(self.out)
        self.i += 1
        if self.work[j] is not None:
            self.work[j].wait()
        if not self.collective:
            self.out[j].copy_(local)
            return self.out[j]
        self.keep[j] = local.contiguous()
        self.work[j] = dist.all_gather_into_tensor(self.out[j], self.keep[j], group=self.group, async_op=True)
        return self.out[j]

    def wait(self):
        """block the current stream (not the host) until every outstanding gather has landed; returns the most recent buffer."""
        for j, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[j] = None
        return self.out[(self.i - 1) % len(self.out)] if self.i else None


def gather_report(n_rows_local, dtype, device, reps=10, group=None):
    """what the N > 1 bench line says about its only exchange (collective: every rank calls it, outside the timed region): the rows every rank
    holds (all_gather_object), and the duration of ONE stand-alone all_gather_into_tensor of the per-row log-probs -- mean of `reps`, HIP events
    on a GPU, host clock otherwise, MAX over ranks.  In the timed steps the same collective runs asynchronously behind the next step's kernels
    (PipelinedGather); this is its exposed cost if nothing hid it."""
    import time
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if not collectives_active(group):
        return {"rows_per_rank": [int(n_rows_local)], "gather_us": 0.0, "gather_bytes_per_rank": 0}
    rows = [None] * world
    dist.all_gather_object(rows, int(n_rows_local), group=group)
    on_gpu = torch.device(device).type == "cuda"
    out = {"rows_per_rank": [int(r) for r in rows], "gather_us": None, "gather_bytes_per_rank": int(n_rows_local) * torch.empty((), dtype=dtype).element_size()}
    if len(set(rows)) != 1:
        return out                                     # unequal shards: the bench does not gather (all_gather_into_tensor wants equal blocks)
    local = torch.zeros((n_rows_local,), dtype=dtype, device=device)
    full = torch.empty((world * n_rows_local,), dtype=dtype, device=device)
    for _ in range(2):
        dist.all_gather_into_tensor(full, local, group=group)
    if on_gpu:
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dist.all_gather_into_tensor(full, local, group=group)
        e1.record()
        torch.cuda.synchronize(device)
        us = e0.elapsed_time(e1) * 1e3 / reps
    else:
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_gather_into_tensor(full, local, group=group)
        us = (time.perf_counter() - t0) * 1e6 / reps
    t = torch.tensor([us], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    out["gather_us"] = float(t.item())
    return out


def timed_steps(step, steps, warmup, finish=None, device=None, timer=None):
    """the benchmark contract's timing loop: `warmup` untimed calls of step(), then EXACTLY `steps` calls bracketed by a barrier + device
    synchronisation on both sides; returns the wall time in seconds, MAX over ranks.  `finish()` (optional) runs inside the timed region after
    the last step (flush deferred status words, wait for outstanding gathers).  `device`: the rank's torch device (None / cpu: no device sync,
    which is how the world-size-2 gloo tests drive this loop without a GPU).  `timer`: optional context manager active during the timed steps
    (per-kernel HIP events)."""
    import contextlib
    import time
    multi = collectives_active()
    on_gpu = device is not None and torch.device(device).type == "cuda"

    def fence():
        if multi:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize(device)

    for _ in range(warmup):
        step()
    if finish is not None:
        finish()
    fence()
    t0 = time.perf_counter()
    with (timer if timer is not None else contextlib.nullcontext()):
        for _ in range(steps):
            step()
    if finish is not None:
        finish()
    fence()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def allreduce_gradients(parameters, average=True, group=None):
    """data-parallel training step glue: every rank back-propagates the loss of ITS row shard, then ONE all-reduce (RCCL ncclAllReduce over
    xGMI) of all parameter gradients packed into a single flat bucket per dtype -- the whole model is < 1 MB (largest BASELINE configuration:
    16 138 MLP scalars), so one bucket, one collective, no overlap machinery.  With `average` the sum is divided by the world size (loss =
    mean over the GLOBAL batch when every shard has the same number of rows).  Parameters without a gradient on this rank contribute zeros
    (they keep grad None only if no rank produced one).  Returns the number of scalars reduced."""
    params = [p for p in parameters if p.requires_grad]
    if not collectives_active(group):
        return sum(p.grad.numel() for p in params if p.grad is not None)
    world = dist.get_world_size(group)
    total = 0
    by_dtype = {}
    for p in params:
        by_dtype.setdefault((p.dtype, p.device), []).append(p)
    for (dtype, device), ps in by_dtype.items():
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=dtype, device=device)
        has = torch.zeros(len(ps), dtype=dtype, device=device)
        o = 0
        for i, p in enumerate(ps):
            if p.grad is not None:
                flat[o:o + p.numel()] = p.grad.reshape(-1)
                has[i] = 1
            o += p.numel()
        bucket = torch.cat([flat, has])
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
        flat, has = bucket[:flat.numel()], bucket[flat.numel():]
        if average:
            flat = flat / world
        o = 0
        for i, p in enumerate(ps):
            if has[i] > 0:
                g = flat[o:o + p.numel()].reshape(p.shape)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
            o += p.numel()
        total += flat.numel()
    return total
