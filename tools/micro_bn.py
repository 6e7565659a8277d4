"""BatchNorm row kernels on the trunk's layer shapes: achieved HBM bandwidth per kernel (bf16)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maggie_amd import kernels as K, functional as MF
dev = torch.device('cuda:0')
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
print('%-22s %10s %10s %10s %10s   (us, GB/s)' % ('rows x C', 'colstats', 'affine', 'bwd_reduce', 'bwd_apply'))
for M, C in ((4 * 512 * 512, 32), (4 * 256 * 256, 32), (4 * 256 * 256, 64), (4 * 128 * 128, 64), (4 * 64 * 64, 128), (4 * 32 * 32, 256), (4 * 16 * 16, 512)):
    x = torch.randn(M, C, device=dev).bfloat16(); dy = torch.randn(M, C, device=dev).bfloat16()
    scale = torch.rand(C, device=dev) + 0.5; shift = torch.randn(C, device=dev); mean = torch.zeros(C, device=dev); invstd = torch.ones(C, device=dev)
    y = K.affine_act(x, scale, shift, act=K.ACT_RELU)
    nb = M * C * 2
    st = torch.zeros((K.STAT_REPLICAS, 2 * C), device=dev)
    sums = torch.zeros(2 * C, device=dev)
    r = [t(lambda: K.colstats(x, st)), t(lambda: K.affine_act(x, scale, shift, act=K.ACT_RELU, out=y)),
         t(lambda: K.bn_backward(dy, y, x, scale, mean, invstd, M, act=K.ACT_RELU, reduce_only=True, sums=sums)),
         t(lambda: K.bn_backward(dy, y, x, scale, mean, invstd, M, act=K.ACT_RELU, apply_only=True, sums=sums))]
    traffic = [nb, 2 * nb, 3 * nb, 4 * nb]
    print('%-22s ' % ('%d x %d' % (M, C)) + ' '.join('%5.1f/%4.0f' % (u, b / u / 1e3) for u, b in zip(r, traffic)))
