import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from maggie_amd import kernels as K
dev = torch.device('cuda:0')
def bench(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
shapes = [(4, 128, 128, 64, 3), (4, 256, 256, 32, 3), (4, 64, 64, 128, 3), (4, 512, 512, 16, 3), (4, 32, 32, 512, 3)]
if os.environ.get('MICRO_DEEP'):
    shapes = [(4, 512, 512, 16, 3), (4, 256, 256, 32, 3), (4, 512, 256, 32, 3), (4, 256, 512, 32, 3), (4, 1280, 512, 16, 1), (4, 512, 256, 16, 3),
              (4, 256, 512, 16, 3), (4, 256, 256, 16, 3), (4, 2048, 256, 16, 1), (4, 128, 128, 32, 3), (4, 512, 512, 32, 3)]
for (N, Cin, Cout, HW, k) in shapes:
    x = torch.randn(N * HW * HW, Cin, device=dev).bfloat16()
    dy = torch.randn(N * HW * HW, Cout, device=dev).bfloat16()
    w = torch.randn(Cout, k * k, Cin, device=dev).bfloat16()
    geo = dict(N=N, Hin=HW, Win=HW, Hout=HW, Wout=HW, R=k, S=k, stride=1, pad=1, dil=1)
    fl = 2.0 * N * HW * HW * Cin * Cout * k * k
    tw = bench(lambda: K.conv_wgrad(x, dy, cout=Cout, mode=K.MODE_CONV, **geo))
    tf = bench(lambda: K.conv_fprop(x, w, mode=K.MODE_CONV, **{k_: v for k_, v in geo.items() if k_ not in ('Hout', 'Wout')}))
    print('MG_WGRAD_BLOCKS=%s  N%d C%d->%d %dx%d: wgrad %.1f us (%.0f TF)  fprop %.1f us (%.0f TF)' % (os.environ.get('MG_WGRAD_BLOCKS'), N, Cin, Cout, HW, HW, tw, fl / tw / 1e6, tf, fl / tf / 1e6))
