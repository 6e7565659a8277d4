"""Child process of tests/test_gpu_graphs.py::test_mailbox_allreduce_two_processes_on_one_gpu: rank r of a 2-rank gloo group, BOTH ranks on cuda:0.
(a) raw exchanges of random packs, eager and replayed from a hipGraph, against the sum computed over gloo on the host;
(b) functional.batch_norm_act with an nn.SyncBatchNorm holder on DIFFERENT row counts per rank (forward + backward) against BatchNorm over the
    concatenated rows computed with torch on the host -- the whole SyncBN plumbing of the HIP path with a real second rank.
usage: python tests/mailbox_worker.py <rank> <port>  -> 'RESULT {...}'"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, port = int(sys.argv[1]), sys.argv[2]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK=str(rank), WORLD_SIZE='2', MAGGIE_SYNCBN_COMM='mailbox')

import torch                        # noqa: E402
import torch.distributed as dist    # noqa: E402

dist.init_process_group('gloo', rank=rank, world_size=2)
torch.cuda.set_device(0)
dev = torch.device('cuda:0')
from maggie_amd import functional as MF, parallel     # noqa: E402

comm = parallel.syncbn_direct_comm()
res = {'rank': rank}

# ---- (a) raw exchanges
g = torch.Generator().manual_seed(100 + rank)
worst = 0.0
for it, n in enumerate([1, 7, 129, 1025, 1088, 64, 513] * 6):
    x = torch.randn(n, generator=g)
    ref = x.clone()
    dist.all_reduce(ref)                                   # gloo, on the host
    y = comm.all_reduce_sum_(x.to(dev).contiguous())
    worst = max(worst, float((y.cpu() - ref).abs().max()))
comm.check()
res['raw_max_err'] = worst

# replayed from a hipGraph: the kernel takes its exchange number from the device counter, so every replay is a new exchange
xs = [torch.zeros(257, device=dev), torch.zeros(1025, device=dev)]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for x in xs:
        comm.all_reduce_sum_(x)                            # warm-up outside the capture (same call sequence on both ranks)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    for x in xs:
        comm.all_reduce_sum_(x)
worst = 0.0
for rep in range(5):
    hosts = [torch.randn(x.numel(), generator=g) for x in xs]
    for x, h in zip(xs, hosts):
        x.copy_(h)
    graph.replay()
    torch.cuda.synchronize()
    for x, h in zip(xs, hosts):
        ref = h.clone()
        dist.all_reduce(ref)
        worst = max(worst, float((x.cpu() - ref).abs().max()))
comm.check()
res['graph_max_err'] = worst

# ---- (b) SyncBN through functional.batch_norm_act: rank 0 has 40 rows, rank 1 has 72
C, eps = 64, 1e-5
g0 = torch.Generator().manual_seed(5)
xs_all = [torch.randn(40, C, generator=g0) * 2 + 1, torch.randn(72, C, generator=g0) - 0.5]
ws_all = [torch.randn(40, C, generator=g0), torch.randn(72, C, generator=g0)]
gamma0, beta0 = torch.randn(C, generator=g0), torch.randn(C, generator=g0)
bn = torch.nn.SyncBatchNorm(C, eps=eps).to(dev)
with torch.no_grad():
    bn.weight.copy_(gamma0)
    bn.bias.copy_(beta0)
bn.train()
x = xs_all[rank].to(dev).requires_grad_(True)
y = MF.batch_norm_act(x, bn, MF.ACT_NONE)
(y * ws_all[rank].to(dev)).sum().backward()
comm.check()
# reference: BatchNorm over the concatenated rows, L = L_0 + L_1
xa = torch.cat(xs_all).requires_grad_(True)
ga, ba = gamma0.clone().requires_grad_(True), beta0.clone().requires_grad_(True)
ya = torch.nn.functional.batch_norm(xa, None, None, ga, ba, True, 0.1, eps)
(ya * torch.cat(ws_all)).sum().backward()
lo = 0 if rank == 0 else 40
hi = lo + xs_all[rank].shape[0]
res['y_err'] = float((y.detach().cpu() - ya.detach()[lo:hi]).abs().max())
res['dx_err'] = float((x.grad.cpu() - xa.grad[lo:hi]).abs().max())
# dgamma / dbeta stay LOCAL (nn.SyncBatchNorm semantics): their sum over the ranks is the full-batch gradient
dg, db = bn.weight.grad.cpu().clone(), bn.bias.grad.cpu().clone()
dist.all_reduce(dg)
dist.all_reduce(db)
res['dgamma_err'] = float((dg - ga.grad).abs().max() / ga.grad.abs().max())
res['dbeta_err'] = float((db - ba.grad).abs().max() / ba.grad.abs().max())
res['running_mean_err'] = float((bn.running_mean.cpu() - 0.1 * xa.detach().mean(0)).abs().max())

# ---- (c) round 5: conv -> SyncBatchNorm -> LeakyReLU -> conv in bf16 with the BatchNorm on the consumer's operand path (functional.BNLazy under the
# mailbox exchange: statistics over BOTH ranks' rows, the normalised activation never stored) against the stored form, bit for bit -- outputs, input
# gradient, both weight gradients, dgamma / dbeta, running statistics. Rank 0 has 2 samples, rank 1 has 3.
def chain(lazy):
    MF.LAZY_BN_SYNC = lazy
    gc = torch.Generator().manual_seed(77)
    w1 = (torch.randn(64, 9, 32, generator=gc) / 17).to(dev, torch.bfloat16).requires_grad_(True)
    w2 = (torch.randn(32, 9, 64, generator=gc) / 24).to(dev, torch.bfloat16).requires_grad_(True)
    xin = [torch.randn(2, 40, 48, 32, generator=gc), torch.randn(3, 40, 48, 32, generator=gc)][rank].to(dev, torch.bfloat16).requires_grad_(True)
    dout = [torch.randn(2, 40, 48, 32, generator=gc), torch.randn(3, 40, 48, 32, generator=gc)][rank].to(dev, torch.bfloat16)
    bn2 = torch.nn.SyncBatchNorm(64).to(dev)
    with torch.no_grad():
        bn2.weight.copy_(torch.rand(64, generator=gc) + 0.5)
        bn2.bias.copy_(torch.randn(64, generator=gc) * 0.3)
    bn2.train()
    MF.ARENA.reset(dev)
    h = MF.conv_bn_act(xin, w1, bn2, MF.ACT_LRELU, 3, 3, 1, 1, 1, lazy_out=True)
    was_lazy = isinstance(h, MF.LazyAct)
    out = MF.conv2d(h, w2, None, 3, 3, 1, 1, 1)
    out.backward(dout)
    torch.cuda.synchronize()
    return was_lazy, [t.detach().float().cpu() for t in (out, xin.grad, w1.grad, w2.grad, bn2.weight.grad, bn2.bias.grad, bn2.running_mean, bn2.running_var)]


keep = MF.LAZY_BN_SYNC
try:
    lazy_on, a = chain(True)
    lazy_off, b = chain(False)
finally:
    MF.LAZY_BN_SYNC = keep
comm.check()
res['lazy_sync_taken'] = bool(lazy_on and not lazy_off)
res['lazy_sync_same_bits'] = all(torch.equal(p_, q_) for p_, q_ in zip(a, b))
res['lazy_sync_max_diff'] = max(float((p_ - q_).abs().max()) for p_, q_ in zip(a, b))
res['calls'] = comm.calls
dist.barrier()
parallel.syncbn_destroy_comm()
dist.destroy_process_group()
print('RESULT ' + json.dumps(res))
