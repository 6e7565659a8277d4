"""ASPP -- mirrors maggie/network/module/aspp.py:4-56 (1x1, three dilated 3x3, pooled branch, 1x1 fuse; all +BN+ReLU)."""
import torch
from torch import nn

from ... import functional as MF
from .base import ConvWeight


class ASPP(nn.Module):
    def __init__(self, in_channel, out_channel, conv=None, norm=nn.BatchNorm2d):
        super().__init__()
        mid = 256
        d = [1, 2, 4, 8]
        self.aspp1 = ConvWeight(in_channel, mid, 1, 1, 0, d[0])
        self.aspp2 = ConvWeight(in_channel, mid, 3, 1, d[1], d[1])
        self.aspp3 = ConvWeight(in_channel, mid, 3, 1, d[2], d[2])
        self.aspp4 = ConvWeight(in_channel, mid, 3, 1, d[3], d[3])
        self.aspp5 = ConvWeight(in_channel, mid, 1, 1, 0, 1)
        self.aspp1_bn, self.aspp2_bn, self.aspp3_bn = norm(mid), norm(mid), norm(mid)
        self.aspp4_bn, self.aspp5_bn = norm(mid), norm(mid)
        self.conv2 = ConvWeight(mid * 5, out_channel, 1, 1, 0, 1)
        self.bn2 = norm(out_channel)

    def plain_convs(self):
        """Conv holders whose per-step layout / dtype conversion rides on the batched weight pipeline (functional.plain_krsc)."""
        return [self.aspp1, self.aspp2, self.aspp3, self.aspp4, self.aspp5, self.conv2]

    def forward(self, x):
        """x: (N, H, W, 512) NHWC."""
        dt = x.dtype
        N, H, W_, C = x.shape

        xs = MF.Fan(x, 5)                                         # five consumers: their gradients meet in one launch (fp32 sum, rounded once)

        def branch(conv, bn):
            w = MF.plain_krsc(conv, dt)                           # (looked up on the calling stream: host-side only)
            xb = xs()
            return lambda: MF.conv_bn_act(xb, w, bn, MF.ACT_RELU, conv.kernel_size, conv.kernel_size, 1, conv.padding, conv.dilation)

        x5_in = xs()

        def pooled_branch():
            pooled = MF.spatial_mean(x5_in)                        # AdaptiveAvgPool2d(1): one launch each way (was cast + strided reduce + cast)
            return MF.conv_bn_act(pooled, w5, self.aspp5_bn, MF.ACT_RELU, 1, 1, 1, 0, 1)

        w5 = MF.plain_krsc(self.aspp5, dt)
        # five independent conv + BatchNorm strings over a (N, 16, 16) map: every kernel under-fills the chip, so they run as parallel
        # branches (functional.parallel_branches) -- forward and, through autograd's per-node streams, backward
        fns = [branch(c, b) for c, b in ((self.aspp1, self.aspp1_bn), (self.aspp2, self.aspp2_bn), (self.aspp3, self.aspp3_bn),
                                         (self.aspp4, self.aspp4_bn))] + [pooled_branch]
        outs = MF.parallel_branches(fns, x.device, 'aspp')
        x5 = outs[4]
        outs = list(outs[:4]) + [MF.spatial_broadcast(x5, H, W_)]                   # nearest upsample of a 1x1 map
        y = torch.cat(outs, -1)
        w2 = MF.plain_krsc(self.conv2, dt)
        return MF.conv_bn_act(y, w2, self.bn2, MF.ACT_RELU, 1, 1, 1, 0, 1, link_out=True)     # sole consumer: the decoder's first (transposed) conv
