"""Convolution parameter holders (the MI355X build keeps parameters in the reference's layouts and names; the
compute happens in maggie_amd.functional). Mirrors maggie/network/module/base.py:3-10."""
import torch
from torch import nn


class ConvWeight(nn.Module):
    """Holds `weight` (Cout, Cin, k, k) [+ `bias`] like nn.Conv2d (transposed: (Cin, Cout, k, k) like
    nn.ConvTranspose2d) together with the geometry; never runs a torch convolution."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, padding=0, dilation=1, bias=False, transposed=False):
        super().__init__()
        self.in_channels, self.out_channels = in_planes, out_planes
        self.kernel_size, self.stride, self.padding, self.dilation, self.transposed = kernel_size, stride, padding, dilation, transposed
        shape = (in_planes, out_planes, kernel_size, kernel_size) if transposed else (out_planes, in_planes, kernel_size, kernel_size)
        self.weight = nn.Parameter(torch.empty(shape))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_planes))
        else:
            self.register_parameter('bias', None)

    def extra_repr(self):
        return '%d -> %d, k=%d, s=%d, p=%d, d=%d%s' % (self.in_channels, self.out_channels, self.kernel_size, self.stride,
                                                      self.padding, self.dilation, ', transposed' if self.transposed else '')


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    """3x3 convolution with padding (bias-free)"""
    assert groups == 1
    return ConvWeight(in_planes, out_planes, 3, stride, dilation, dilation, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    """1x1 convolution (bias-free)"""
    return ConvWeight(in_planes, out_planes, 1, stride, 0, 1, bias=False)


class Marker(nn.Module):
    """Parameter-free placeholder that keeps nn.Sequential indices identical to the reference's
    (ReLU / LeakyReLU / AvgPool2d / UpsamplingNearest2d / Sigmoid slots)."""

    def __init__(self, kind=''):
        super().__init__()
        self.kind = kind

    def extra_repr(self):
        return self.kind
