"""Mailbox all-reduce for the SyncBatchNorm statistics exchange (csrc/mailbox.hip): one ordinary kernel per exchange over peer-mapped
mailboxes -- no RCCL, no host work, capturable into hipGraphs. OPT-IN across devices (round 5, ADVICE high): `MAGGIE_SYNCBN_COMM=auto` picks it
only when all ranks share ONE device -- the placement it has been exercised on (two and four PROCESSES ON ONE GPU, tests/test_gpu_graphs.py);
across GPUs it relies on fine-grained device memory + system-scope atomics over xGMI, which this project had no second GPU to validate on, so the
default there is the private RCCL communicator (rccl_direct.py) and `MAGGIE_SYNCBN_COMM=mailbox` selects this one explicitly.
A peer that does not arrive within the spin budget (MAGGIE_MAILBOX_TIMEOUT_S, default 600 s -- the order of a process-group timeout: the
reference's flow lets ranks skew by minutes, e.g. rank-0-only validation between steps) raises the error word (1 + its rank): MaGGIe.forward
reads it with the step's other flags and raises MaggieHipError naming the rank -- never a silent wrong normalisation.

The control plane (mailbox handle exchange) goes through the existing torch.distributed group (any backend: the handles are 64-byte objects)."""
import ctypes
import os

import torch
import torch.distributed as dist

from . import hip

MAX_RANKS, PACK = 8, 1088


class _Mailbox(ctypes.Structure):
    _fields_ = [('peer', ctypes.c_void_p * MAX_RANKS), ('world', ctypes.c_int32), ('rank', ctypes.c_int32)]


class MailboxComm:
    def __init__(self, group=None, device=None, spin_seconds=None):
        if spin_seconds is None:
            #  the long budget is for peers on OTHER devices (a late peer = a slow step somewhere else). Ranks that share one device
            # -- the only placement that is selected automatically -- compete with the very peer they wait for: a dead one must not hold the device,
            # and every captured graph replay, for ten minutes before the error word is read
            shared = os.environ.get('MAGGIE_ONE_GPU') == '1' or torch.cuda.device_count() == 1
            spin_seconds = float(os.environ.get('MAGGIE_MAILBOX_TIMEOUT_S', '30' if shared else '600'))
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > MAX_RANKS:
            raise hip.MaggieHipError('MailboxComm: at most %d ranks (one node)' % MAX_RANKS)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        lib = hip.lib()
        self._own = ctypes.c_void_p()
        self._opened = []
        # Every step of the set-up is followed by a vote over the group, so that ALL ranks leave this constructor the same way (communicator or
        # exception): a rank that failed alone would otherwise go on to other collectives while its peers still wait in this one.
        handle = (ctypes.c_char * 64)()
        rc = lib.mg_mailbox_create(ctypes.c_int(self.world), ctypes.byref(self._own), handle)
        handles = [None] * self.world
        dist.all_gather_object(handles, (int(rc), bytes(handle)), group=group)
        bad = [r for r, (c, _) in enumerate(handles) if c != 0]
        if bad:
            self.destroy()
            raise hip.MaggieHipError('MailboxComm: mg_mailbox_create failed on rank(s) %s (no fine-grained device memory / IPC export there)' % bad)
        self._mb = _Mailbox()
        self._mb.world, self._mb.rank = self.world, self.rank
        rc_open = 0
        for r, (_, h) in enumerate(handles):
            if r == self.rank:
                self._mb.peer[r] = self._own.value
            else:
                p = ctypes.c_void_p()
                rc_open = rc_open or int(lib.mg_mailbox_open(ctypes.c_char_p(h), ctypes.byref(p)))
                if p.value:
                    self._mb.peer[r] = p.value
                    self._opened.append(p)
        self._state = torch.zeros(2, dtype=torch.int32, device=self.device)       # [exchange counter, error word]
        self._spin = int(spin_seconds * 1e8)                                     # wall_clock64 ticks (100 MHz)
        self.calls = 0
        votes = [None] * self.world                                              # also the barrier: nobody deposits before every mailbox is mapped everywhere
        dist.all_gather_object(votes, rc_open, group=group)
        if any(votes):
            self.destroy()
            raise hip.MaggieHipError('MailboxComm: mapping a peer mailbox failed on rank(s) %s' % [r for r, v in enumerate(votes) if v])
        # ... and the protocol itself is tried before anything relies on it: two exchanges with a 10 s spin budget (peer-mapped fine-grained memory and
        # system-scope atomics between DIFFERENT devices are the one thing the single-GPU tests of this project cannot exercise)
        ok = self._self_test()
        dist.all_gather_object(votes, bool(ok), group=group)
        if not all(votes):
            self.destroy()
            raise hip.MaggieHipError('MailboxComm: the exchange self-test failed on rank(s) %s' % [r for r, v in enumerate(votes) if not v])

    def _self_test(self):
        spin, self._spin = self._spin, int(10.0 * 1e8)       # (ranks sharing ONE GPU time-slice: an exchange can take a good fraction of a second)
        try:
            base = torch.arange(1, 9, dtype=torch.float32, device=self.device)
            ok = True
            for it in range(2):
                t = base * float(self.rank + 1) + float(it)
                self.all_reduce_sum_(t)
                want = torch.arange(1, 9, dtype=torch.float32) * (self.world * (self.world + 1) / 2.0) + float(it * self.world)
                ok = ok and bool(torch.equal(t.cpu(), want))
            ok = ok and int(self._state[1].item()) == 0
            self._state[1].zero_()
            self.calls = 0
            return ok
        except Exception:
            return False
        finally:
            self._spin = spin

    def all_reduce_sum_(self, t):
        """In-place sum over the ranks of a contiguous fp32 device tensor, on the current stream (eager or capturing)."""
        return self.all_reduce_sum_to(t, t)

    def all_reduce_sum_to(self, src, dst):
        """dst = sum over the ranks of src (src untouched unless dst is src). Packs larger than a mailbox cell (PACK floats: BatchNorm layers wider
        than 543 channels) go through several exchanges of PACK elements -- the same chunks in the same order on every rank."""
        for t in (src, dst):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.numel() > 0):
                raise hip.MaggieHipError('MailboxComm.all_reduce_sum: contiguous fp32 device tensors expected')
        if src.numel() != dst.numel():
            raise hip.MaggieHipError('MailboxComm.all_reduce_sum_to: source and destination differ in size')
        s = self._state
        n = src.numel()
        for off in range(0, n, PACK):
            m = min(PACK, n - off)
            hip.call('mg_mailbox_allreduce_to', ctypes.byref(self._mb), ctypes.c_void_p(src.data_ptr() + 4 * off), ctypes.c_void_p(dst.data_ptr() + 4 * off),
                     ctypes.c_int(m), ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(s.data_ptr() + 4), ctypes.c_long(self._spin), hip.stream())
            self.calls += 1
        return dst

    def can_finalize(self, C):
        return 2 * C + 1 <= PACK

    def bn_finalize(self, stats, count, gamma, beta, running_mean, running_var, momentum, eps, count_dev=None):
        """SyncBN forward statistics in ONE launch (mg_mailbox_bn_finalize): this rank's statistics ([nrep][2C] or [2C]: sum x | sum x^2) and row
        count -> (scale, shift, mean, invstd) views of one [4C] buffer, the global count [1]; running statistics updated in place.
        `count_dev` (int32 [1] on the device): the row count when only the device knows it (the sparse head's live rows)."""
        nrep = stats.shape[0] if stats.dim() == 2 else 1
        C = stats.shape[-1] // 2
        outs = torch.empty(4 * C, dtype=torch.float32, device=stats.device)
        cnt = torch.empty(1, dtype=torch.float32, device=stats.device)
        s = self._state
        hip.call('mg_mailbox_bn_finalize_dev', ctypes.byref(self._mb), hip.ptr(stats), ctypes.c_int(nrep), ctypes.c_float(float(count)), hip.ptr(count_dev),
                 ctypes.c_int(C), hip.ptr(gamma), hip.ptr(beta), hip.ptr(running_mean), hip.ptr(running_var), ctypes.c_float(momentum), ctypes.c_float(eps),
                 hip.ptr(outs), hip.ptr(cnt), ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(s.data_ptr() + 4), ctypes.c_long(self._spin), hip.stream())
        self.calls += 1
        return outs[:C], outs[C:2 * C], outs[2 * C:3 * C], outs[3 * C:], cnt

    @property
    def error_word(self):
        """Device int32 [1]: 0, or 1 + the rank that did not arrive at an exchange within the spin budget (read with the step's other flags)."""
        return self._state[1:2]

    def raise_for(self, word):
        if int(word) != 0:
            raise hip.MaggieHipError('SyncBatchNorm statistics exchange (mailbox): rank %d did not arrive within the spin budget of %.0f s on rank %d -- '
                                     'the step\'s normalisation is invalid' % (int(word) - 1, self._spin / 1e8, self.rank))

    def check(self):
        """Host read of the error word (a peer that did not arrive within the spin budget)."""
        self.raise_for(int(self._state[1].item()))

    def destroy(self):
        lib = hip.lib()
        torch.cuda.synchronize()
        for p in self._opened:
            lib.mg_mailbox_close(p)
        self._opened = []
        if self._own:
            lib.mg_mailbox_free(self._own)
            self._own = ctypes.c_void_p()
