"""Row sharding of a batch over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

The hot path is embarrassingly parallel over rows: every rank evaluates its own contiguous slice of rows with
replicated weights (< 1 MB); no data-path collective is needed.  The only exchange is the optional all-gather of the per-row
log-probabilities (4 MiB per rank at 2^20 float32 rows) -- one all_gather_into_tensor, never an all-reduce.
"""
import os
import warnings

import torch
import torch.distributed as dist

# An N > 1 step keeps six streams busy (the caller's, up to three step streams, the exchange's side stream, RCCL's own) and the HIP runtime maps
# streams onto GPU_MAX_HW_QUEUES hardware queues -- 4 by default; streams that share a queue serialise (a 2^17-row shard step 0.118 instead of
# 0.091 ms).  The variable is read when the runtime initialises, so it is set HERE, at import, while that has not happened yet;
# later it can only be reported.
HW_QUEUES_WANTED = 8


def ensure_hw_queues(n=HW_QUEUES_WANTED):
    """set GPU_MAX_HW_QUEUES >= n for this process if the HIP runtime has not started; returns what is in force ("set", "kept", or "late": the
    runtime was already initialised with fewer -- a warning says so once)"""
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    if cur is not None and cur.isdigit() and int(cur) >= n:
        return "kept"
    started = torch.cuda.is_initialized() if hasattr(torch.cuda, "is_initialized") else False
    if not started:
        os.environ["GPU_MAX_HW_QUEUES"] = str(n)
        return "set"
    warnings.warn("jammy_flows_amd.parallel: the HIP runtime was initialised before this import with GPU_MAX_HW_QUEUES=%s (< %d): pipelined steps and "
                  "their exchange will share hardware queues and serialise.  Import jammy_flows_amd.parallel (or export GPU_MAX_HW_QUEUES=%d) before "
                  "the first GPU call." % (cur or "4 (runtime default)", n, n), RuntimeWarning, stacklevel=2)
    return "late"


HW_QUEUES_STATE = ensure_hw_queues()


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


_SIDE_STREAMS = {}


class PipelinedGather:
    """all-gather of equal-sized per-rank row blocks that overlaps with the NEXT step's kernels: submit() enqueues the collective
    asynchronously (on a communication stream, once the producer kernels of `local` on the CURRENT stream are done) into one of `depth` rotating
    output buffers and returns immediately; the buffer of a submission is complete after the next submit() into the same slot or after wait().
    A step of the row-sharded hot path therefore never stalls on the 4 MiB log-prob exchange.

    group_steps = k > 1: FEWER, LARGER collectives -- k consecutive submissions are staged (one device copy each, on the submitting stream) and
    exchanged in ONE all-gather of (k, rows) per rank; the output buffer is then (world, k, rows, ...) and a step's rows arrive up to k - 1
    submissions later (flush() / wait() exchange a partly filled stage).  NOTE: with k > 1 the buffer submit() / staged() return is the one the
    stage WILL be exchanged into -- its exchange has not been issued until the k-th submission (or flush() / wait()); read rows through
    wait() + last_block().  An enqueue of RCCL costs ~50 us of host time and a few us of the device
    whatever its size (scripts/probe/gather_cost.py: a 2^17-row shard step 0.103 -> 0.110 ms with one gather per step, 0.103 with one per four).

    `self.path` says what carries the exchange: "torch.distributed" (default), "rccl-direct" (JF_RCCL_DIRECT=1: ncclAllGather through ctypes,
    jammy_flows_amd/rccl.py; every rank falls back to torch unless the direct communicator came up on ALL of them) or "copy" (one rank)."""

    def __init__(self, n_rows_local, dtype, device, tail_shape=(), depth=2, group=None, group_steps=1):
        self.group = group
        self.k = max(1, int(group_steps))
        self.n_rows_local = int(n_rows_local)
        self.device = torch.device(device)
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.collective = collectives_active(group)
        if self.k == 1:
            self.out = [torch.empty((self.world * n_rows_local,) + tuple(tail_shape), dtype=dtype, device=device) for _ in range(depth)]
            self.stage = None
        else:
            self.out = [torch.empty((self.world, self.k, n_rows_local) + tuple(tail_shape), dtype=dtype, device=device) for _ in range(depth)]
            self.stage = [torch.zeros((self.k, n_rows_local) + tuple(tail_shape), dtype=dtype, device=device) for _ in range(depth)]
        self.n_staged = 0
        self.n_submitted = 0
        self._last_row = 0                  # stage row of the most recent submission (n_staged restarts after flush() / wait(), n_submitted does not)
        self._last_out = None               # ... and the buffer its exchange lands in
        self.staged_ev = []
        self._guarded = set()
        self._zero_copy = False
        self.side_stream = None
        self.work = [None] * depth
        self.keep = [None] * depth          # the submitted tensors must outlive their collectives
        self.i = 0
        self.comm, self.comm_stream, self.direct_error = None, None, None
        self.path = "torch.distributed" if self.collective else "copy"
        if self.collective and self.device.type == "cuda":
            from . import rccl
            # whether the direct path is tried at all is decided by ALL ranks (a rank without librccl.so or without JF_RCCL_DIRECT must not skip
            # collectives the others enter): MIN of the local availability first, then -- on every rank of a group that tries -- the communicator
            # (whose set-up always takes part in its broadcast, rccl.Communicator) and the MIN of the outcome
            want = torch.tensor([1 if rccl.available() else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(want, op=dist.ReduceOp.MIN, group=group)
            if int(want.item()) == 1:
                ok = 1
                try:
                    self.comm = rccl.Communicator(self.device, group)
                except Exception as e:           # noqa: BLE001 -- reported (direct_error); the torch path takes over on every rank
                    ok, self.direct_error = 0, "%s: %s" % (type(e).__name__, e)
                flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                if int(flag.item()) == 0 and self.comm is not None:
                    self.comm.destroy()
                    self.comm = None
                if self.comm is not None:
                    self.comm_stream = torch.cuda.Stream(device=self.device)
                    self.path = "rccl-direct"

    def submit(self, local):
        self.n_submitted += 1
        if self.k > 1:
            # consecutive submissions may come from DIFFERENT streams (pipelined steps): every copy into the stage first waits, on its own stream,
            # for the exchange that last read this stage buffer, and leaves an event the exchange of the full stage waits for
            j = self.i % len(self.out)
            self._wait_on_current(self.work[j])
            self.stage[j][self.n_staged].copy_(local)
            self._last_row, self._last_out = self.n_staged, j
            if self.device.type == "cuda":
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self.staged_ev.append(ev)
            self.n_staged += 1
            if self.n_staged < self.k:
                return self.out[j]
            return self._exchange_stage(j)
        return self._exchange(local)           # (submit() copies, next_slot() / staged() do not: one gather object uses one of the two)

    # ---- zero-copy staging for pipelined steps: the step writes its rows straight into the stage of the next exchange
    def next_slot(self, stream=None):
        """the (rows, ...) tensor the NEXT step should write its per-row result into -- pass it as `logp_out` to PipelinedForward.submit(), then
        hand the PendingStep to staged().  No copy, no per-step event: the steps' own completion events order the exchange behind them.  The
        exchange that last read this stage must be over before it is written again: `stream` (the stream the step will run on,
        PipelinedForward.peek_stream()) waits for it, once per stage and stream; without it the caller's current stream (which every pipelined
        step waits for) does, once per stage."""
        j = self.i % len(self.out)
        self._zero_copy = True
        if self.stage is None:                              # group_steps = 1: one send buffer per slot
            self.stage = [torch.zeros((1, self.n_rows_local) + tuple(self.out[0].shape[1:]), dtype=self.out[0].dtype, device=self.device)
                          for _ in range(len(self.out))]
        w = self.work[j]
        if w is not None:
            if stream is None:
                if self.n_staged == 0:
                    self._wait_on_current(w)
            elif stream.cuda_stream not in self._guarded:
                self._guarded.add(stream.cuda_stream)
                if self.comm is not None:
                    stream.wait_event(w)
                else:
                    cur = torch.cuda.current_stream(self.device)
                    torch.cuda.set_stream(stream)
                    try:
                        w.wait()
                    finally:
                        torch.cuda.set_stream(cur)
        return self.stage[j][self.n_staged]

    def staged(self, pending):
        """the step submitted with next_slot() as its output (a PendingStep: .event fires when its rows are in the slot).  Every k-th call issues
        the exchange, from a side stream that waits for the k steps."""
        self.n_submitted += 1
        self.staged_ev.append(pending.event)
        self._last_row, self._last_out = self.n_staged, self.i % len(self.out)
        self.n_staged += 1
        if self.n_staged >= self.k:
            self._exchange_stage(self.i % len(self.out), side=True)
        return self.out[self.i % len(self.out)]

    def _exchange_stage(self, j, side=False):
        """exchange stage j behind the events of the submissions that filled it; side: from the side stream (zero-copy staging: the caller's
        stream, which every pipelined step waits for, must not wait for the steps), else from the current stream"""
        self.n_staged = 0
        self._guarded = set()
        evs, self.staged_ev = self.staged_ev, []
        stage = self.stage[j] if self.k > 1 else self.stage[j][0]
        if self.device.type != "cuda":
            return self._exchange(stage)
        cur = torch.cuda.current_stream(self.device)
        if not side:
            for ev in evs:
                cur.wait_event(ev)
            return self._exchange(stage)
        if self.side_stream is None:                       # one exchange stream per device for the whole process (streams are mapped onto a few
            key = (self.device.type, self.device.index)    # hardware queues: every extra one can land on a step stream's queue and serialise with it)
            if key not in _SIDE_STREAMS:
                _SIDE_STREAMS[key] = torch.cuda.Stream(device=self.device)
            self.side_stream = _SIDE_STREAMS[key]
        torch.cuda.set_stream(self.side_stream)
        try:
            for ev in evs:
                self.side_stream.wait_event(ev)
            return self._exchange(stage)
        finally:
            torch.cuda.set_stream(cur)

    def flush(self):
        """exchange a partly filled stage now (the rows of the missing steps keep what the stage held before)"""
        if self.n_staged > 0 and self.stage is not None:
            self._exchange_stage(self.i % len(self.out), side=self._zero_copy)

    def _wait_on_current(self, w):
        if w is None:
            return
        if self.comm is not None:
            torch.cuda.current_stream(self.device).wait_event(w)
        else:
            w.wait()

    def _wait_one(self, j):
        w, self.work[j] = self.work[j], None
        self._wait_on_current(w)

    def _exchange(self, local):
        j = self.i % len(self.out)
        self.i += 1
        if self.comm is not None:
            cur = torch.cuda.current_stream(self.device)
            send = local if local.is_contiguous() else local.contiguous()
            ready = torch.cuda.Event()
            ready.record(cur)
            self.comm_stream.wait_event(ready)                 # the collective waits for the producer of `local`, nobody waits for the collective
            self.comm.all_gather(self.out[j], send, self.comm_stream)     # (one stream for all gathers: a slot's reuse is ordered behind its last use)
            done = torch.cuda.Event()
            done.record(self.comm_stream)
            send.record_stream(self.comm_stream)
            self.keep[j], self.work[j] = send, done
            return self.out[j]
        if self.work[j] is not None:
            self.work[j].wait()
        if not self.collective:
            self.out[j].copy_(local)
            return self.out[j]
        self.keep[j] = local.contiguous()
        recv = self.out[j] if self.k == 1 else self.out[j].view((self.world * self.k,) + tuple(self.out[j].shape[2:]))   # (gloo wants the dim-0 concatenation)
        self.work[j] = dist.all_gather_into_tensor(recv, self.keep[j], group=self.group, async_op=True)
        return self.out[j]

    def wait(self):
        """block the current stream (not the host) until every outstanding gather has landed (a partly filled stage is exchanged first);
        returns the most recent buffer."""
        self.flush()
        for j in range(len(self.work)):
            self._wait_one(j)
        return self.out[(self.i - 1) % len(self.out)] if self.i else None

    def last_block(self, rank):
        """after wait(): the rows rank `rank` handed to the most recent submit() / staged()"""
        if self.k > 1:
            return self.out[self._last_out][rank, self._last_row]
        full = self.out[(self.i - 1) % len(self.out)]
        return full[rank * self.n_rows_local:(rank + 1) * self.n_rows_local]

    def close(self):
        """wait for the outstanding gathers on the host and give the direct communicator back (collective; before destroy_process_group)"""
        self.wait()
        if self.comm is not None:
            self.comm_stream.synchronize()
            self.comm.destroy()
            self.comm = None


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


def timed_steps(step, steps, warmup, finish=None, device=None, timer=None, info=None):
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
    if info is not None:
        info["host_issue_s"] = time.perf_counter() - t0         # the host's share: when it has issued the last step (the device may still be busy)
    if finish is not None:
        finish()
    fence()
    dt = time.perf_counter() - t0
    t_own = dt
    if info is not None:
        info["rank_s"] = [t_own]
        info["n_ranks_seen"] = 1
    if multi:
        # every rank's own wall time (its clock between the two fences), so that a straggler shows in the line: min / max / mean / slowest rank
        world = dist.get_world_size()
        mine = torch.zeros(world, dtype=torch.float64, device=device if on_gpu else "cpu")
        mine[dist.get_rank()] = t_own
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        per_rank = [float(v) for v in mine.tolist()]
        dt = max(per_rank)
        if info is not None:
            info["rank_s"] = per_rank
            info["n_ranks_seen"] = world
    return dt


def rank_time_stats(info, steps):
    """per-rank ms_per_step statistics of a timed_steps(..., info=info) run for the N > 1 line"""
    ts = [t / max(1, steps) * 1e3 for t in info.get("rank_s", [])]
    if not ts:
        return None
    slow = max(range(len(ts)), key=lambda r: ts[r])
    return {"min": min(ts), "max": max(ts), "mean": sum(ts) / len(ts), "slowest_rank": slow, "per_rank": ts}


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
