#!/usr/bin/env python3
"""A second process that keeps the GPU's CUs, LDS and memory paths busy (conv launches of this library over the trunk's shapes, back to back) while
something else is under test: hand-counted waits that are too generous only show when another process stretches the latencies.
usage: noise_proc.py SECONDS"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maggie_amd import functional as MF          # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(1)
shapes = [(4, 64, 64, 128), (4, 128, 128, 64), (4, 32, 32, 256), (4, 256, 256, 32), (4, 16, 16, 512)]
ops = []
for n, h, w, c in shapes:
    x = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    wt = (torch.randn(c, 3, 3, c, device=dev) * 0.05).to(torch.bfloat16)
    ops.append((x, wt))
t_end = time.time() + float(sys.argv[1])
k = 0
while time.time() < t_end:
    for x, wt in ops:
        for _ in range(20):
            MF.conv2d(x, wt, None, 3, 3, 1, 1, 1)
    torch.cuda.synchronize()
    k += 1
print('noise rounds', k)
