"""wgrad micro-benchmark of the trunk's 3x3 layer shapes (bf16). usage: [MG_WGRAD_HALO=0|1] [MG_WGRAD_HALO_BLOCKS=n] python tools/micro_wgrad.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maggie_amd import kernels as K
dev = torch.device('cuda:0')


def bench(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


shapes = [(4, 32, 32, 512), (4, 32, 32, 256), (4, 64, 64, 128), (4, 128, 128, 64), (4, 256, 256, 32), (4, 512, 512, 16), (4, 512, 256, 32), (4, 256, 128, 64)]
tot = 0.0
for (N, Cin, Cout, HW) in shapes:
    x = torch.randn(N * HW * HW, Cin, device=dev).bfloat16()
    gy = torch.randn(N * HW * HW, Cout, device=dev).bfloat16()
    geo = dict(N=N, Hin=HW, Win=HW, Hout=HW, Wout=HW, R=3, S=3, stride=1, pad=1, dil=1)
    fl = 2.0 * N * HW * HW * Cin * Cout * 9
    tf = bench(lambda: K.conv_wgrad(x, gy, cout=Cout, mode=K.MODE_CONV, **geo))
    tot += tf
    print('HALO=%s B=%s  N%d C%d->%d %dx%d: %.1f us (%.0f TF)' % (os.environ.get('MG_WGRAD_HALO', '1'), os.environ.get('MG_WGRAD_HALO_BLOCKS', '512'), N, Cin, Cout, HW, HW, tf, fl / tf / 1e6))
print('total %.1f us' % tot)
