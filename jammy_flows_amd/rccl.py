"""RCCL called directly (ctypes on the librccl.so that torch.distributed's "nccl" backend has loaded) for the ONE exchange of the row-sharded hot
path: the all-gather of the per-row log-probabilities.

The question it answers: at 8 GPUs a rank's step is 2^17 rows = ~0.1 ms of kernels, issued with ~35 us of host work (one jf_plan_launch); the torch
collective adds ~50 us of host work per step -- is that torch's wrapper (Python, c10d logger, a Work object, events) or RCCL's own enqueue?
ncclAllGather through ctypes is one foreign call on a communication stream that waits for the producing step by event; nothing else changes: same
library, same algorithm selection, same xGMI rings.

The communicator is made the standard way: rank 0 draws a ncclUniqueId, it travels through the already initialised torch.distributed group
(broadcast_object_list), every rank calls ncclCommInitRank.  `available()` says whether the library and its symbols are there; everything that can
fail raises RcclError, and parallel.PipelinedGather falls back to torch.distributed on EVERY rank if ANY rank failed to come up (one all-reduce of a
flag), so the ranks never disagree about which path they are on.

MEASURED (one-rank group on one MI355X, scripts/probe/gather_cost.py, C3 2^17-row steps on two streams): host issue per step 33-38 us without an
exchange, 80 us with this path, 85-89 us with torch.distributed; step time 0.103-0.106 / 0.109-0.111 / 0.107-0.111 ms.  Most of the host cost is
RCCL's own enqueue, not torch's wrapper, so the direct path buys ~7 us of host time and nothing on the device: it is OPT-IN (JF_RCCL_DIRECT=1) and
torch.distributed stays the default carrier -- what pays instead is fewer, larger collectives (PipelinedGather(group_steps=k))."""
import ctypes
import os

import torch
import torch.distributed as dist

NCCL_UNIQUE_ID_BYTES = 128          # rccl.h:40
_DTYPES = {torch.float32: 7, torch.float64: 8, torch.int32: 2, torch.int64: 4, torch.uint8: 1, torch.float16: 6, torch.bfloat16: 9}     # rccl.h:455-470


class RcclError(RuntimeError):
    pass


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * NCCL_UNIQUE_ID_BYTES)]


_LIB = None


def _lib():
    """the librccl.so inside torch's own lib directory (the one its process groups run on); loading it again by path returns the same handle"""
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            raise RcclError("librccl.so not found next to torch (%s)" % path)
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        lib.ncclGetUniqueId.restype = ctypes.c_int
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        lib.ncclCommInitRank.restype = ctypes.c_int
        lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclAllGather.restype = ctypes.c_int
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclCommDestroy.restype = ctypes.c_int
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        _LIB = lib
    return _LIB


def available():
    if os.environ.get("JF_RCCL_DIRECT", "0") != "1":
        return False
    try:
        _lib()
        return True
    except (RcclError, OSError, AttributeError):
        return False


def _check(rc, what):
    if rc != 0:
        msg = _lib().ncclGetErrorString(rc)
        raise RcclError("%s failed: %s (%d)" % (what, msg.decode() if msg else "?", rc))


class Communicator:
    """one RCCL communicator over the ranks of an initialised torch.distributed group (collective constructor: every rank of the group calls it)"""

    def __init__(self, device, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RcclError("torch.distributed is not initialised (the unique id travels through it)")
        self.device = torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.comm = ctypes.c_void_p()
        uid = _UniqueId()
        # rank 0 ALWAYS takes part in the broadcast: a failure before it (library missing, ncclGetUniqueId) travels as an error marker and is
        # raised on every rank after it -- the others would otherwise wait in the broadcast forever
        box = [None]
        lib, early = None, None
        try:
            lib = _lib()
            if self.rank == 0:
                _check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
                box = [ctypes.string_at(ctypes.byref(uid), NCCL_UNIQUE_ID_BYTES)]     # (all 128 bytes: `.internal` stops at a NUL)
        except (RcclError, OSError, AttributeError) as e:
            early = e
            if self.rank == 0:
                box = [("error", "%s: %s" % (type(e).__name__, e))]
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = box[0]
        if early is not None:
            raise RcclError("rank %d: %s" % (self.rank, early))
        if isinstance(raw, tuple) and raw and raw[0] == "error":
            raise RcclError("rank 0 could not create the unique id: %s" % raw[1])
        if not isinstance(raw, (bytes, bytearray)) or len(raw) != NCCL_UNIQUE_ID_BYTES:
            raise RcclError("unique id did not arrive")
        ctypes.memmove(ctypes.byref(uid), bytes(raw), NCCL_UNIQUE_ID_BYTES)
        with torch.cuda.device(self.device):
            _check(lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")

    def all_gather(self, recv, send, stream):
        """recv[(rank r's block)] <- send of rank r, enqueued on `stream` (a torch.cuda.Stream); contiguous tensors, recv.numel() == world x send.numel()"""
        if recv.numel() != self.world * send.numel() or recv.dtype != send.dtype or not (recv.is_contiguous() and send.is_contiguous()):
            raise RcclError("all_gather: recv must be contiguous, of send's dtype and world x its size")
        _check(_lib().ncclAllGather(send.data_ptr(), recv.data_ptr(), send.numel(), _DTYPES[send.dtype], self.comm, stream.cuda_stream), "ncclAllGather")

    def destroy(self):
        """explicit (PipelinedGather.close): nothing is destroyed at interpreter exit, where the HIP runtime may already be gone"""
        if self.comm:
            comm, self.comm = self.comm, ctypes.c_void_p()
            _check(_lib().ncclCommDestroy(comm), "ncclCommDestroy")
