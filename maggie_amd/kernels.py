"""Typed Python wrappers over the C ABI (one function per entry point of include/maggie_hip.h).

Tensors are NHWC / rows x channels with the channel dimension contiguous. Nothing here computes on the host:
each function validates shapes, allocates the output through torch and launches the HIP kernel on the current stream.
"""
import ctypes

import torch

from . import hip
from .hip import ConvParams, MODE_CONV, MODE_TCONV, MODE_GATHER, ACT_NONE, ACT_RELU, ACT_LRELU  # noqa: F401


def _ld(t):
    """Row pitch (elements) of a channel-contiguous tensor."""
    assert t.stride(-1) == 1, 'channel dimension must be contiguous'
    return t.stride(-2) if t.dim() >= 2 else t.shape[-1]


def conv_out_size(mode, hin, r, stride, pad, dil, output_padding=0):
    if mode == MODE_CONV:
        return (hin + 2 * pad - dil * (r - 1) - 1) // stride + 1
    return (hin - 1) * stride - 2 * pad + dil * (r - 1) + 1 + output_padding


def _conv_params(x, w, y, mode, N, Hin, Win, Hout, Wout, R, S, stride, pad, dil, M, Cin, Cout, nbr=None, scale=None,
                 shift=None, res=None, res_mode=0, res2=None, act=ACT_NONE, pre_act=False, slope=0.2, stats=None,
                 yoff=0):
    hip.need_cuda(x, w, y, nbr, scale, shift, res, res2, stats)
    p = ConvParams()
    p.x, p.w, p.y, p.nbr = hip.ptr(x), hip.ptr(w), hip.ptr(y), hip.ptr(nbr)
    p.scale, p.shift, p.res, p.res2, p.stats = hip.ptr(scale), hip.ptr(shift), hip.ptr(res), hip.ptr(res2), hip.ptr(stats)
    p.dtype, p.mode = hip.dtype_code(x), mode
    p.N, p.Hin, p.Win, p.Cin, p.Hout, p.Wout, p.Cout = N, Hin, Win, Cin, Hout, Wout, Cout
    p.R, p.S, p.stride, p.pad, p.dil, p.M = R, S, stride, pad, dil, M
    p.ldx, p.ldy, p.yoff = _ld(x), _ld(y), yoff
    p.ldr = _ld(res) if res is not None else 0
    p.ldr2 = _ld(res2) if res2 is not None else 0
    p.act, p.pre_act, p.res_mode, p.slope = act, int(bool(pre_act)), (res_mode if res is not None else 0), slope
    return p


def conv_fprop(x, w, *, mode=MODE_CONV, N=1, Hin=1, Win=1, Hout=None, Wout=None, R=1, S=1, stride=1, pad=0, dil=1,
               M=None, nbr=None, scale=None, shift=None, res=None, res_mode=1, res2=None, act=ACT_NONE, pre_act=False,
               slope=0.2, stats=None, out=None, yoff=0, cout=None):
    """Y = epilogue(implicit GEMM). x: (..., Cin) channel-contiguous; w: (Cout, R*S, Cin) same dtype.
    Dense modes: rows of x are (n, h, w) of an (N, Hin, Win) map; gather mode: rows of x are sparse sites, `nbr` (M, R*S).
    `out`/`yoff` let the result land in a channel slice of a wider buffer (zero-copy concat)."""
    Cin = x.shape[-1]
    Cout = w.shape[0] if cout is None else cout
    assert w.shape[-1] == Cin and w.dtype == x.dtype and w.is_contiguous(), (w.shape, x.shape, w.dtype, x.dtype)
    if mode == MODE_GATHER:
        M = nbr.shape[0] if M is None else M
        Hout = Wout = 1
        assert nbr.dtype == torch.int32 and nbr.is_contiguous() and nbr.shape[1] == R * S
    else:
        if Hout is None:
            Hout = conv_out_size(mode, Hin, R, stride, pad, dil)
            Wout = conv_out_size(mode, Win, S, stride, pad, dil)
        M = N * Hout * Wout
    if out is None:
        out = torch.empty((M, Cout), dtype=x.dtype, device=x.device)
    if res is not None:
        assert res.dtype == x.dtype
    if res2 is not None:
        assert res2.dtype == x.dtype
    if stats is not None:
        assert stats.dtype == torch.float32 and stats.numel() >= 2 * Cout
    for v in (scale, shift):
        if v is not None:
            assert v.dtype == torch.float32 and v.numel() >= Cout
    p = _conv_params(x, w, out, mode, N, Hin, Win, Hout, Wout, R, S, stride, pad, dil, M, Cin, Cout, nbr, scale, shift,
                     res, res_mode, res2, act, pre_act, slope, stats, yoff)
    hip.call('mg_conv_fprop', ctypes.byref(p), hip.stream())
    return out


def conv_wgrad(x, dy, *, cout, mode=MODE_CONV, N=1, Hin=1, Win=1, Hout=1, Wout=1, R=1, S=1, stride=1, pad=0, dil=1,
               M=None, nbr=None, yoff=0, out=None):
    """dW[co, tap, ci] = sum_m dY[m, co] X[src(m,tap), ci]; fp32 (Cout, R*S, Cin). `dy` may be a channel slice
    (yoff) of a wider buffer."""
    Cin = x.shape[-1]
    if mode == MODE_GATHER:
        M = nbr.shape[0] if M is None else M
    else:
        M = N * Hout * Wout
    if out is None:
        out = torch.zeros((cout, R * S, Cin), dtype=torch.float32, device=x.device)
    assert dy.dtype == x.dtype
    p = _conv_params(x, None, dy, mode, N, Hin, Win, Hout, Wout, R, S, stride, pad, dil, M, Cin, cout, nbr,
                     stats=out, yoff=yoff)
    hip.call('mg_conv_wgrad', ctypes.byref(p), hip.stream())
    return out
