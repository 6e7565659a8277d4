"""Decoder up-sampling residual block -- mirrors maggie/network/decoder/resnet.py:9-45 (BasicBlock only; the MGM baseline
decoders of that file are out of scope)."""
import torch.nn as nn

from ... import functional as MF
from ..module import SpectralNorm, conv3x3, ConvWeight, Marker


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, upsample=None, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.stride = stride
        if self.stride > 1:
            self.conv1 = SpectralNorm(ConvWeight(inplanes, inplanes, 4, 2, 1, 1, bias=False, transposed=True))
        else:
            self.conv1 = SpectralNorm(conv3x3(inplanes, inplanes))
        self.bn1 = norm_layer(inplanes)
        self.conv2 = SpectralNorm(conv3x3(inplanes, planes))
        self.bn2 = norm_layer(planes)
        self.upsample = upsample

    def forward(self, x, post_add=None, link_out=False):
        """`post_add` (the encoder shortcut feature) is added after the block's activation; fused into the last kernel in
        inference. `link_out`: the caller feeds the result to the next block's first conv and to nothing else (functional.BnLink)."""
        dt = x.dtype
        # `carry`: the skip branch takes x back from the first conv, whose data-gradient kernel then adds the skip gradient in its epilogue
        if self.stride > 1:
            out, x = MF.conv_bn_act(x, self.conv1.krsc(dt, x.shape[-1]), self.bn1, MF.ACT_LRELU, 4, 4, 2, 1, 1, transposed=True, carry=True, link_out=True, lazy_out=True)
        else:
            out, x = MF.conv_bn_act(x, self.conv1.krsc(dt, x.shape[-1]), self.bn1, MF.ACT_LRELU, 3, 3, 1, 1, 1, carry=True, link_out=True, lazy_out=True)
        identity, res_mode = x, 1
        if self.upsample is not None:
            u = self.upsample
            if isinstance(u[0], Marker):
                # UpsamplingNearest2d(2) -> SN 1x1 -> BN  ==  (SN 1x1 -> BN) at low resolution, replicated 2x2 in the residual add. Mean and
                # biased variance are those of the replicated tensor; its BatchNorm counts 4 samples per row (unbiased running variance)
                identity = MF.conv_bn_act(x, u[1].krsc(dt, x.shape[-1]), u[2], MF.ACT_NONE, 1, 1, 1, 0, 1, count_mult=4)
                res_mode = 2
            else:
                identity = MF.conv_bn_act(x, u[0].krsc(dt, x.shape[-1]), u[1], MF.ACT_NONE, 1, 1, 1, 0, 1)
        return MF.conv_bn_act(out, self.conv2.krsc(dt, out.shape[-1]), self.bn2, MF.ACT_LRELU, 3, 3, 1, 1, 1, res=identity,
                              res_mode=res_mode, res2=post_add, link_out=link_out and post_add is None)
