"""fprop micro-benchmark of the trunk's channel-aligned layer shapes (bf16): register-staged loop (MG_FPROP_ASYNC=0) vs the direct-to-LDS
ring (default), interleaved in ONE process per variant env. usage: MG_FPROP_ASYNC=0|1 [MG_ASYNC_NS=3|4] python tools/micro_fprop.py"""
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


# (N, Cin, Cout, H, k, mode, stride): encoder stages at 512x512 batch 4, decoder, ASPP-like, dgrad (mode 1)
shapes = [(4, 32, 32, 512, 3, 0, 1), (4, 32, 32, 512, 3, 1, 1), (4, 32, 32, 256, 3, 0, 1), (4, 32, 64, 128, 3, 0, 1), (4, 64, 64, 128, 3, 0, 1), (4, 128, 128, 64, 3, 0, 1), (4, 256, 256, 32, 3, 0, 1), (4, 512, 512, 16, 3, 0, 1), (4, 512, 256, 32, 3, 0, 1),
          (4, 256, 128, 64, 3, 0, 1), (4, 128, 128, 64, 3, 1, 1), (4, 64, 64, 128, 3, 1, 1), (4, 1280, 512, 16, 1, 0, 1), (4, 128, 64, 64, 1, 0, 1)]
tot = 0.0
for (N, Cin, Cout, HW, k, mode, stride) in shapes:
    x = torch.randn(N * HW * HW, Cin, device=dev).bfloat16()
    w = (torch.randn(Cout, k * k, Cin, device=dev) / (k * k * Cin) ** 0.5).bfloat16()
    geo = dict(N=N, Hin=HW, Win=HW, R=k, S=k, stride=stride, pad=k // 2, dil=1)
    fl = 2.0 * N * HW * HW * Cin * Cout * k * k
    tf = bench(lambda: K.conv_fprop(x, w, mode=mode, **geo))
    tot += tf
    print('ASYNC=%s NS=%s  N%d C%d->%d %dx%d k%d mode %d: %.1f us (%.0f TF)' % (os.environ.get('MG_FPROP_ASYNC', '1'), os.environ.get('MG_ASYNC_NS', '4'), N, Cin, Cout, HW, HW, k, mode, tf, fl / tf / 1e6))
print('total %.1f us' % tot)
