"""A private RCCL communicator driven through librccl's C API (ctypes), for the SyncBatchNorm statistics exchange INSIDE captured hipGraphs.

Why not torch.distributed for these: ProcessGroupNCCL hands every collective's completion event to its watchdog thread, and a watchdog that
queries an event recorded inside an active stream capture aborts the process (`hipErrorCapturedEvent`: observed in roughly one capture out of
ten on ROCm 7.2 / torch 2.10). A collective issued straight through `ncclAllReduce` on the capturing stream has no such
bookkeeping: RCCL records its kernels as ordinary graph nodes, and the 142 small all-reduces of a SyncBN step (`sync_bn: true` in both
configs/maggie_{image,video}.yaml; engine/train.py:159-161) replay with the rest of the step instead of being launched one by one from the host.

The communicator is created collectively (rank 0's ncclUniqueId travels through the existing torch.distributed group) and used for in-place
fp32 sum all-reduces on the CURRENT torch stream only. There is no watchdog behind it: ranks must issue the same sequence of exchanges, which
the rank-safe graphs guarantee (every rank replays the same graphs; maggie_amd/network/arch/maggie.py:_rank_safe_graphs)."""
import ctypes
import os

import torch
import torch.distributed as dist

from . import hip

_NCCL_FLOAT, _NCCL_SUM = 7, 0
_LIB = None


class _UniqueId(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_char * 128)]


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
        lib = ctypes.CDLL(path if os.path.isfile(path) else 'librccl.so')          # the copy torch itself loaded
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclCommDestroy):
            f.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def _check(rc, what):
    if rc != 0:
        raise hip.MaggieHipError('RCCL %s failed: %s' % (what, _lib().ncclGetErrorString(rc).decode()))


class DirectComm:
    def __init__(self, group=None):
        """Collective over `group` (default: the world). The current HIP device must already be this rank's (torch.cuda.set_device)."""
        lib = _lib()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        uid = _UniqueId()
        rc0 = lib.ncclGetUniqueId(ctypes.byref(uid)) if self.rank == 0 else 0
        box = [(rc0, bytes(bytearray(uid))) if self.rank == 0 else None]      # a failure on rank 0 travels WITH the broadcast: nobody waits for an id that never comes
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        self.comm = None
        _check(box[0][0], 'ncclGetUniqueId (rank 0)')
        ctypes.memmove(ctypes.addressof(uid), box[0][1], 128)
        comm = ctypes.c_void_p()
        _check(lib.ncclCommInitRank(ctypes.byref(comm), self.world, uid, self.rank), 'ncclCommInitRank')
        self.comm = comm
        self.calls = 0

    def self_test(self):
        """One small all-reduce outside any capture: rank r contributes (r + 1) * [1, 2, 3, 4] -> world * (world + 1) / 2 * [1, 2, 3, 4] everywhere."""
        t = torch.tensor([1.0, 2.0, 3.0, 4.0], device='cuda') * float(self.rank + 1)
        self.all_reduce_sum_(t)
        self.calls -= 1
        want = torch.tensor([1.0, 2.0, 3.0, 4.0]) * (self.world * (self.world + 1) / 2.0)
        return bool(torch.equal(t.cpu(), want))

    def all_reduce_sum_(self, t):
        """In-place sum over the ranks of a contiguous fp32 device tensor, on the current torch stream (eager or capturing)."""
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise hip.MaggieHipError('DirectComm.all_reduce_sum_: contiguous fp32 device tensor expected')
        if self.comm is None:
            raise hip.MaggieHipError('DirectComm used after destroy()')
        p = ctypes.c_void_p(t.data_ptr())
        _check(_lib().ncclAllReduce(p, p, t.numel(), _NCCL_FLOAT, _NCCL_SUM, self.comm, hip.stream()), 'ncclAllReduce')
        self.calls += 1
        return t

    def all_reduce_sum_to(self, src, dst):
        """Out-of-place form: dst = sum over the ranks of src (src untouched)."""
        for t in (src, dst):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                raise hip.MaggieHipError('DirectComm.all_reduce_sum_to: contiguous fp32 device tensors expected')
        if self.comm is None:
            raise hip.MaggieHipError('DirectComm used after destroy()')
        _check(_lib().ncclAllReduce(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), src.numel(), _NCCL_FLOAT, _NCCL_SUM, self.comm,
                                    hip.stream()), 'ncclAllReduce')
        self.calls += 1
        return dst

    def destroy(self):
        if self.comm is not None:
            torch.cuda.synchronize()
            _lib().ncclCommDestroy(self.comm)
            self.comm = None
