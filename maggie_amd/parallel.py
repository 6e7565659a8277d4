"""Data-parallel helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The hot path shards on the batch axis only: the only data-path exchange is the gradient all-reduce
(plus the BatchNorm statistics of `sync_bn: true`; `nn.SyncBatchNorm` holders are honoured by functional.BNAct).

Two ways to all-reduce gradients:
  * torch.nn.parallel.DistributedDataParallel (`wrap_ddp`, what the reference's engine/train.py:159-164 does) -- works unchanged,
    also with the hipGraph trunk;
  * `GradSync` -- the MI355X-first variant. With the trunk in a hipGraph, its ~150 parameter gradients leave the backward
    graph as views of ONE flat buffer (graphs.GraphedCallable.export_param_grads), so the whole trunk is a single large RCCL
    all-reduce with no per-parameter hooks, bucket copies or autograd-graph traversal; the detail-stage gradients (ready
    before the backward graph is even launched) are coalesced into a second, small one. xGMI is point-to-point: few large
    ring all-reduces are exactly what it wants."""
import torch
import torch.distributed as dist


def reduce_bn_stats(pack, C, group):
    """pack = [sum(C), sumsq(C), count] of the local rows -> global (mean, biased var, count)."""
    pack = pack.clone()
    dist.all_reduce(pack, group=group)
    n = float(pack[2 * C])
    mean = pack[:C] / n
    var = (pack[C:2 * C] / n - mean * mean).clamp_min(0)
    return mean, var, n


SYNCBN_COMM = None       # rccl_direct.DirectComm: the statistics exchange goes through a private RCCL communicator (capturable into hipGraphs)


def _mailbox_possible(group, validated_only=False):
    """Every rank of the group on ONE node, at most 8 of them, every pair of their devices with peer access (or the same device): the conditions
    under which the mailbox kernels can see each other's fine-grained memory. Collective (object all-gather on the existing group).
    `validated_only`: additionally require that all ranks share ONE device -- the only placement this project's tests have ever run the mailbox
    on (fine-grained memory + system-scope atomics between DIFFERENT devices over xGMI is unvalidated: ADVICE round 4, high)."""
    import socket
    world = dist.get_world_size(group)
    if world > 8:
        return False
    me = (socket.gethostname(), torch.cuda.current_device())
    everyone = [None] * world
    dist.all_gather_object(everyone, me, group=group)
    if len({h for h, _ in everyone}) != 1:
        return False
    devs = sorted({d for _, d in everyone})
    if validated_only and len(devs) != 1:
        return False
    ok = all(a == b or torch.cuda.can_device_access_peer(a, b) for a in devs for b in devs) if len(devs) <= torch.cuda.device_count() else False
    votes = [None] * world
    dist.all_gather_object(votes, bool(ok), group=group)
    return all(votes)


SYNCBN_COMM_FAILED = False     # no in-graph exchange could be set up (decided once, collectively): sync_bn then runs through torch.distributed, eagerly


def _all_ranks(ok, group):
    votes = [None] * dist.get_world_size(group)
    dist.all_gather_object(votes, bool(ok), group=group)
    return all(votes)


def syncbn_direct_comm(group=None):
    """The communicator of the in-graph SyncBN statistics exchange, created on first use -- COLLECTIVELY: every rank must get here at the same
    point (MaGGIe._graph_policy calls it on every rank's first training forward; `setup_syncbn` does it explicitly at set-up time).
    MAGGIE_SYNCBN_COMM = auto (default, round 5): the private RCCL communicator whenever the ranks sit on DIFFERENT devices -- RCCL waits for a
    late peer the way the reference's NCCL SyncBatchNorm does (rank-0-only validation / checkpointing between steps, engine/train.py:294, a
    dataloader stall, an uneven first capture), and it is the exchange that has run across GPUs. The mailbox kernels (one plain kernel per
    exchange, fused with the BatchNorm finalize, rank-ordered sums) are the automatic choice only where they have been validated: all ranks on
    ONE device (the multi-process tests). `mailbox` opts into them across the devices of one node (peer access, fine-grained mailboxes mapped
    everywhere and a trial exchange all voted on; spin budget MAGGIE_MAILBOX_TIMEOUT_S, default 600 s); `rccl` forces RCCL; both raise when the
    form cannot be had. No in-graph exchange at all -> None: the caller keeps the eager torch.distributed exchange. Every decision is a vote
    over the group: all ranks end up on the same path."""
    global SYNCBN_COMM, SYNCBN_COMM_FAILED
    if SYNCBN_COMM is not None or SYNCBN_COMM_FAILED:
        return SYNCBN_COMM
    import logging
    import os
    kind = os.environ.get('MAGGIE_SYNCBN_COMM', 'auto')
    if kind not in ('auto', 'mailbox', 'rccl'):
        raise ValueError('MAGGIE_SYNCBN_COMM must be auto, mailbox or rccl, not %r' % kind)
    why = []
    if kind == 'mailbox' and not _mailbox_possible(group):
        raise RuntimeError('MAGGIE_SYNCBN_COMM=mailbox: the ranks are not on one node with peer access between all of their devices (at most 8)')
    if kind == 'mailbox' or (kind == 'auto' and _mailbox_possible(group, validated_only=True)):
        from .mailbox import MailboxComm
        comm = None
        try:
            comm = MailboxComm(group)                             # votes inside: all ranks get a communicator or all get the exception
        except Exception as e:                                   # noqa: BLE001 -- whatever went wrong, the answer is the next form
            why.append('mailbox: %s' % e)
        if _all_ranks(comm is not None, group):
            SYNCBN_COMM = comm
            return SYNCBN_COMM
        if comm is not None:
            comm.destroy()
        if kind == 'mailbox':
            raise RuntimeError('MAGGIE_SYNCBN_COMM=mailbox: the mailbox exchange could not be set up (%s)' % '; '.join(why))
    from .rccl_direct import DirectComm
    comm = None
    try:
        comm = DirectComm(group)
        if not comm.self_test():
            raise RuntimeError('trial all-reduce gave wrong sums')
    except Exception as e:                                       # noqa: BLE001
        why.append('rccl: %s' % e)
        if comm is not None:
            comm.destroy()
        comm = None
    if _all_ranks(comm is not None, group):
        SYNCBN_COMM = comm
        return SYNCBN_COMM
    if comm is not None:
        comm.destroy()
    if kind == 'rccl':
        raise RuntimeError('MAGGIE_SYNCBN_COMM=rccl: the private RCCL communicator could not be set up (%s)' % '; '.join(why))
    SYNCBN_COMM_FAILED = True
    logging.warning('MaGGIe (MI355X build): no in-graph SyncBN exchange on this group (%s) -- sync_bn runs through torch.distributed, eagerly',
                    '; '.join(why) or 'a peer rank failed')
    return None


def setup_syncbn(model=None, group=None):
    """Create the SyncBN exchange communicator NOW (collective) instead of lazily inside the first training forward: call it on every rank right after
    `nn.SyncBatchNorm.convert_sync_batchnorm(model)` when ranks may reach their first forward at different times (e.g. rank 0 evaluating first).
    Refuses BatchNorm layers bound to a sub-group: the communicator spans the default group."""
    if model is not None:
        for m in model.modules():
            if isinstance(m, torch.nn.SyncBatchNorm) and m.process_group is not None and m.process_group is not dist.group.WORLD:
                raise RuntimeError('MaGGIe (MI355X build): the in-graph SyncBN exchange spans the default process group; a SyncBatchNorm with its own '
                                   'process_group needs MAGGIE_SYNCBN_GRAPHS=0 (eager exchange through torch.distributed)')
    return syncbn_direct_comm(group)


def syncbn_destroy_comm():
    global SYNCBN_COMM, SYNCBN_COMM_FAILED
    SYNCBN_COMM_FAILED = False
    if SYNCBN_COMM is not None:
        SYNCBN_COMM.destroy()
        SYNCBN_COMM = None


def _sum_over_ranks_(t, group):
    if SYNCBN_COMM is not None and t.is_cuda:
        SYNCBN_COMM.all_reduce_sum_(t)
    else:
        dist.all_reduce(t, group=group)
    return t


def syncbn_exchange_forward(local_pack, group):
    """Forward statistics exchange of SyncBatchNorm as functional.BNAct does it: local_pack = [sum x (C), sum x^2 (C), rows] of this
    rank's rows -> the same pack summed over the group (a copy; variable row counts per rank are fine: the count rides along).
    torch's SyncBatchNorm all-gathers (mean, invstd, count) instead (engine/train.py:160-161); the pooled moments are the same numbers."""
    return _sum_over_ranks_(local_pack.clone(), group)


def syncbn_exchange_backward(local_sums, group):
    """Backward exchange: local_sums = [sum g (C), sum g * xhat (C)] of this rank's rows -> (global sums for the dx formula, the LOCAL sums
    untouched). torch's SyncBatchNorm all-reduces these two vectors for dx only and keeps grad_weight / grad_bias local (DDP then averages
    them over the ranks); returning the all-reduced sums as dgamma / dbeta would make every BN parameter gradient world_size times too
    large."""
    if SYNCBN_COMM is not None and local_sums.is_cuda:
        return SYNCBN_COMM.all_reduce_sum_to(local_sums, torch.empty_like(local_sums)), local_sums      # out of place: no copy kernel
    return _sum_over_ranks_(local_sums.clone(), group), local_sums


def shard_items(n_items, rank, world):
    """Weak-scaling item assignment: rank r owns items r, r+world, ..."""
    return list(range(rank, n_items, world))


def max_over_ranks(value):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    if dist.get_backend() == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def wrap_ddp(model, device_index):
    """SyncBatchNorm conversion + DDP exactly like the reference's engine/train.py:159-164."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    return DDP(model, device_ids=[device_index], find_unused_parameters=False, gradient_as_bucket_view=True)


def all_reduce_mean(flat, group=None):
    """In-place mean over the process group: one `all_reduce(AVG)` on RCCL ("nccl"); gloo has no AVG, so SUM then divide."""
    if dist.get_backend(group) == 'nccl':
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(flat, group=group)
        flat.div_(dist.get_world_size(group))
    return flat


class OverlappedGradSync:
    """Gradient all-reduce overlapped with backward on a side HIP stream (north_star; what DDP's bucket hooks give the reference,
    engine/train.py:163-164), MI355X-first: the unit is not a 25 MiB bucket but the flat gradient buffer of one captured backward graph.

    A training step's backward is three hipGraphs in reverse-autograd order -- detail stage (sparse head, ~0.3 M parameters), dense
    decoder + ASPP (~14 M), encoder (~16 M). Each graph ends by packing its parameter gradients into one flat buffer; when a graph
    finishes, its buffer is handed to `reduce_async`, which all-reduces (mean) it over RCCL on a side stream while the NEXT graph runs on
    the main stream. Only the encoder's chunk (the last, ~63 MB: one large ring all-reduce, what point-to-point xGMI wants) is exposed,
    and the optimizer waits for it with `wait()`. `reduced` = ids of the parameters whose gradient already went through a collective
    this step; the optimizer reduces the rest itself (eager steps before a geometry's graphs exist). The collective sequence is the same
    on every rank because every rank sees the same geometries in the same order (a failed capture is fatal in distributed runs)."""

    def __init__(self, group=None):
        self.group = group
        self.stream = None
        self.events = []
        self.reduced = set()

    def attach(self, model):
        """Route the gradient buffers of every captured graph of `model` (present and future) through this object."""
        model.__dict__['_grad_overlap'] = self
        for store in ('_trunk_graphs', '_trunk_enc_graphs', '_detail_graphs'):
            for entry in model.__dict__.get(store, {}).values():
                g = entry[0] if isinstance(entry, tuple) else entry
                if hasattr(g, 'grad_hook'):
                    g.grad_hook = self.reduce_async
        return self

    def reduce_async(self, flats, params):
        if not flats[0].is_cuda:                                  # CPU tensors (gloo tests of the bookkeeping): nothing to overlap with
            for f in flats:
                all_reduce_mean(f, self.group)
            self.reduced.update(id(p) for p in params)
            return
        main = torch.cuda.current_stream()
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        self.stream.wait_stream(main)                             # the fresh buffers were filled on the main stream
        with torch.cuda.stream(self.stream):
            for f in flats:
                all_reduce_mean(f, self.group)
                f.record_stream(self.stream)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.events.append(ev)
        self.reduced.update(id(p) for p in params)

    def wait(self):
        """The current stream waits for every exchange started since the last wait; -> the ids of the parameters already reduced."""
        if self.events:
            main = torch.cuda.current_stream()
            for ev in self.events:
                main.wait_event(ev)
        done, self.events, self.reduced = self.reduced, [], set()
        return done


def reduce_remaining_runs(flat, params, offsets, total, done, group=None):
    """All-reduce (mean) the slices of the optimizer's flat gradient buffer that belong to parameters NOT in `done` (ids), one collective
    per contiguous run of such parameters. `offsets[i]` = start of parameter i in `flat`, `total` = its length. Every rank computes the
    same runs because every rank's graphs cover the same parameters. -> number of collectives issued."""
    ends = list(offsets[1:]) + [total]
    n, i = 0, 0
    while i < len(params):
        if id(params[i]) in done:
            i += 1
            continue
        j = i
        while j + 1 < len(params) and id(params[j + 1]) not in done:
            j += 1
        all_reduce_mean(flat[offsets[i]:ends[j]], group)
        n += 1
        i = j + 1
    return n


def grad_buffers(params, fill_missing=False):
    """Tensors that together hold every gradient of `params` exactly once: a flat 1-D base buffer where all of it is covered
    by gradient views (the hipGraph trunk's export: ~150 gradients = ONE tensor), the gradients themselves otherwise."""
    bases, loose = {}, []
    for p in params:
        g = p.grad
        if g is None:
            if not fill_missing:
                continue
            g = p.grad = torch.zeros_like(p)
        b = g._base
        if b is not None and b.dim() == 1 and b.is_contiguous() and g.is_contiguous():
            ent = bases.setdefault(id(b), [b, 0, []])
            ent[1] += g.numel()
            ent[2].append(g)
        else:
            loose.append(g)
    whole = []
    for b, covered, views in bases.values():
        if covered == b.numel():
            whole.append(b)
        else:                                                     # partially used base: its views count as loose gradients
            loose.extend(views)
    return whole, loose


def clip_grad_norm_(params, max_norm, eps=1e-6):
    """torch.nn.utils.clip_grad_norm_(params, max_norm) (2-norm; what engine/train.py:274 calls), computed over the flat
    gradient buffers: a handful of launches instead of one multi-tensor pass over every parameter. Returns the total norm."""
    whole, loose = grad_buffers(list(params))
    bufs = whole + loose
    if not bufs:
        return torch.zeros(())
    norms = torch._foreach_norm(bufs)
    total = torch.linalg.vector_norm(torch.stack([n.float() for n in norms]))
    coef = (max_norm / (total + eps)).clamp(max=1.0)
    torch._foreach_mul_(bufs, coef)
    return total


class GradSync:
    """Average the gradients of `model` over the process group after backward(): `sync = GradSync(model)` once, then
    `loss.backward(); sync(); optimizer.step()`.

    The collective layout is RANK-INVARIANT: every trainable parameter, in `model.parameters()` order, has a fixed slot in one flat fp32
    buffer; gradients are gathered into it with one multi-tensor copy (a parameter without a gradient on this rank contributes zeros),
    ONE all-reduce averages the buffer, one multi-tensor copy hands the results back. Whether a rank's gradients happen to be views of a
    hipGraph's exported flat buffer or loose tensors (a rank whose capture failed, an evicted graph, a different geometry history) no
    longer changes what is sent -- the earlier per-buffer scheme could issue collectives of different sizes on different ranks.
    (`FlatAdamW(sync_group=True)` does the same inside its own flat buffer and is the faster path; this class serves torch optimizers.)"""

    def __init__(self, model, group=None):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.avg = dist.is_available() and dist.is_initialized() and dist.get_backend(group) == 'nccl'
        self._flat = None

    def _all_reduce(self, flat):
        all_reduce_mean(flat, self.group)

    def __call__(self):
        if self.world <= 1 and not (dist.is_available() and dist.is_initialized()):
            return
        ps = self.params
        n = sum(p.numel() for p in ps)
        dev = ps[0].device
        if self._flat is None or self._flat.numel() != n or self._flat.device != dev:
            self._flat = torch.zeros(n, dtype=torch.float32, device=dev)
            self._views, o = [], 0
            for p in ps:
                self._views.append(self._flat[o:o + p.numel()].view(p.shape))
                o += p.numel()
        have = [(v, p.grad) for v, p in zip(self._views, ps) if p.grad is not None]
        if len(have) != len(ps):
            self._flat.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self._all_reduce(self._flat)
        with torch.no_grad():
            for v, p in zip(self._views, ps):
                if p.grad is None:
                    p.grad = v.clone()
            torch._foreach_copy_([p.grad for _, p in zip(self._views, ps)], self._views)
