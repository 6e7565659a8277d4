import sys; sys.path.insert(0, '/root/repo')
import torch
from maggie_amd import kernels as K
dev = torch.device('cuda:0')
for M, C in ((733000, 64), (733000, 32), (183000, 64), (136000, 32), (11000, 64)):
    dy = torch.randn(M, C, device=dev).bfloat16(); y = torch.randn(M, C, device=dev).bfloat16()
    for _ in range(3): K.bias_act_bwd(dy, y, True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): K.bias_act_bwd(dy, y, True)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(M, C, '%.1f us  %.0f GB/s' % (us, M * C * 6 / us / 1e3))
