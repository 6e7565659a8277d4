"""Numerics of the halo fprop forms on a grid of shapes (plain epilogue) against torch on the GPU in fp32. usage: python tools/halo_check.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from maggie_amd import kernels as K
dev = torch.device('cuda:0')
torch.manual_seed(0)
for N in (1, 2):
    for Cin in (96, 128, 256):
        for Cout in (64, 128, 256):
            for H, W in ((16, 16), (9, 17), (24, 40)):
                x = torch.randn(N, Cin, H, W, device=dev).bfloat16()
                w = (torch.randn(Cout, Cin, 3, 3, device=dev) / (9 * Cin) ** 0.5).bfloat16()
                ref = F.conv2d(x.float(), w.float(), None, 1, 1)
                xd = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous()
                wd = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous()
                y = K.conv_fprop(xd, wd, mode=K.MODE_CONV, N=N, Hin=H, Win=W, R=3, S=3, stride=1, pad=1, dil=1)
                y = y.float().reshape(N, H, W, Cout).permute(0, 3, 1, 2)
                err = (y - ref).abs().max().item() / ref.abs().max().item()
                print('N%d C%d->%d %dx%d: rel err %.4f %s' % (N, Cin, Cout, H, W, err, 'BAD' if err > 0.02 else ''))
