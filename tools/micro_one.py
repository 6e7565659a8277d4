"""One conv shape, a few launches (for rocprofv3 --pmc passes). usage: micro_one.py N Cin Cout HW k [iters]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maggie_amd import kernels as K
N, Cin, Cout, HW, k = map(int, sys.argv[1:6])
it = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = torch.device('cuda:0')
x = torch.randn(N * HW * HW, Cin, device=dev).bfloat16()
w = torch.randn(Cout, k * k, Cin, device=dev).bfloat16()
for _ in range(it):
    K.conv_fprop(x, w, mode=K.MODE_CONV, N=N, Hin=HW, Win=HW, R=k, S=k, stride=1, pad=k // 2, dil=1)
torch.cuda.synchronize()
