"""Data-parallel helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The hot path shards on the batch axis only: gradients are all-reduced by
torch.nn.parallel.DistributedDataParallel (bucketed, overlapped with backward on RCCL's stream); the only other exchange
is the BatchNorm statistics of `sync_bn: true` (`nn.SyncBatchNorm` holders are honoured by functional.BNAct)."""
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
