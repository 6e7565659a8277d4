"""Typed Python wrappers over the C ABI (one function per entry point of include/maggie_hip.h).

Tensors are NHWC / rows x channels with the channel dimension contiguous. Nothing here computes on the host:
each function validates shapes, allocates the output through torch and launches the HIP kernel on the current stream.
"""
import ctypes

import torch

from . import hip
from .hip import ConvParams, MODE_CONV, MODE_TCONV, MODE_GATHER, ACT_NONE, ACT_RELU, ACT_LRELU  # noqa: F401

_DT_TAG = {torch.float32: 'f32', torch.bfloat16: 'bf16', torch.float16: 'f16'}


def _ld(t):
    """Row pitch (elements) of a channel-contiguous tensor."""
    assert t.shape[-1] == 1 or t.stride(-1) == 1, 'channel dimension must be contiguous'
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


def _set_xf(p, xf, Cin):
    if xf is None:
        return
    sc, sh, act, slope = xf
    assert sc.dtype == torch.float32 and sh.dtype == torch.float32 and sc.numel() >= Cin and sh.numel() >= Cin and sc.is_contiguous() and sh.is_contiguous()
    hip.need_cuda(sc, sh)
    p.xf_scale, p.xf_shift, p.xf_act, p.xf_slope = hip.ptr(sc), hip.ptr(sh), int(act), float(slope)


_XF_OK = {}


def conv_xform_ok(dtype, N, H, W, Cin, Cout, R=3, S=3, stride=1, pad=1, dil=1, which=0):
    """Does the kernel form a dense (N, H, W, Cin) -> Cout convolution of this geometry dispatches to apply the operand transform in flight?
    which: 0 forward (mg_conv_fprop), 1 weight gradient (mg_conv_wgrad*). Asked of the library (mg_conv_xform_ok), cached per geometry."""
    key = (dtype, N, H, W, Cin, Cout, R, S, stride, pad, dil, which)
    ok = _XF_OK.get(key)
    if ok is None:
        p = ConvParams()
        p.dtype, p.mode = hip.code_of(dtype), MODE_CONV
        Ho, Wo = conv_out_size(MODE_CONV, H, R, stride, pad, dil), conv_out_size(MODE_CONV, W, S, stride, pad, dil)
        p.N, p.Hin, p.Win, p.Cin, p.Hout, p.Wout, p.Cout = N, H, W, Cin, Ho, Wo, Cout
        p.R, p.S, p.stride, p.pad, p.dil, p.M = R, S, stride, pad, dil, N * Ho * Wo
        p.ldx, p.ldy, p.yoff = Cin, Cout, 0
        fn = hip.lib().mg_conv_xform_ok
        fn.restype = ctypes.c_int
        ok = _XF_OK[key] = bool(fn(ctypes.byref(p), ctypes.c_int(which)))
    return ok


def conv_fprop(x, w, *, mode=MODE_CONV, N=1, Hin=1, Win=1, Hout=None, Wout=None, R=1, S=1, stride=1, pad=0, dil=1,
               M=None, nbr=None, scale=None, shift=None, res=None, res_mode=1, res2=None, act=ACT_NONE, pre_act=False,
               slope=0.2, stats=None, out=None, yoff=0, cout=None, rows=None, alg_cin=None, alg_cout=None, bnb=None, xf=None):
    """Y = epilogue(implicit GEMM). x: (..., Cin) channel-contiguous; w: (Cout, R*S, Cin) same dtype.
    `xf` = (scale, shift, act, slope): operand transform -- x is the RAW output of the producing conv, the BatchNorm + activation between the two
    layers is applied in flight (mg_conv_params.xf_*; only for geometries conv_xform_ok reports).
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
    stat_mode = stat_rep = 0
    if stats is not None:
        stat_mode = 1 if stats.dim() == 1 else 0                 # 1-D [2*Cout]: single row, column sums only
        stat_rep = 0 if stat_mode else stats.shape[0]            # rows of the statistics buffer: output tile t adds to row t % rows
        assert stats.dtype == torch.float32 and stats.numel() >= (1 if stat_mode else stat_rep) * 2 * Cout and stats.is_contiguous()
        assert not (stat_mode and hip.DETERMINISTIC), 'a sums-only statistics row is accumulated with atomics: not available in deterministic mode' 
    for v in (scale, shift):
        if v is not None:
            assert v.dtype == torch.float32 and v.numel() >= Cout
    p = _conv_params(x, w, out, mode, N, Hin, Win, Hout, Wout, R, S, stride, pad, dil, M, Cin, Cout, nbr, scale, shift,
                     res, res_mode, res2, act, pre_act, slope, stats, yoff)
    p.stat_mode, p.stat_rep = stat_mode, stat_rep
    p.m_dev = hip.ptr(rows)                    # device row count (sparse head): M is then the capacity, the launch a persistent grid
    _set_xf(p, xf, Cin)
    if bnb is not None:
        # this launch produces the gradient at the OUTPUT of a training BatchNorm layer: its epilogue writes g = dz * act'(z) and accumulates
        # that layer's backward sums into `stats` (replicated layout) -- bnb = (z | None, x, mean, invstd, act), rows x Cout like `out`
        # (+ scale, shift: the layer never stored z -- operand-path BatchNorm -- and the mask is re-formed from x * scale + shift)
        bz, bx, bmean, binv, bact = bnb[:5]
        bsc, bsh = (bnb[5], bnb[6]) if len(bnb) > 5 else (None, None)
        assert stats is not None and stat_mode == 0 and bx.dtype == x.dtype and bx.shape[-1] == Cout and _ld(bx) == Cout
        assert bz is None or (bz.dtype == x.dtype and _ld(bz) == Cout)
        hip.need_cuda(bx, bmean, binv, bsc, bsh)
        p.bnb_y, p.bnb_x, p.bnb_mean, p.bnb_invstd = hip.ptr(bz), hip.ptr(bx), hip.ptr(bmean), hip.ptr(binv)
        p.bnb_act, p.bnb_ld = bact, Cout
        p.bnb_scale, p.bnb_shift = hip.ptr(bsc), hip.ptr(bsh)
    # bench accounting: `work` = ALGORITHMIC FLOPs -- the taps that exist (a stride-s transposed / data-gradient conv touches
    # R*S/s^2 taps per output row on average; the rest of what MG_MODE_TCONV multiplies are structural zeros) and the real, unpadded
    # channel counts; the executed FLOPs ride along in the tag. A device row count (sparse head) makes the work unknown here: None.
    executed = 2.0 * M * Cout * R * S * Cin
    work = None
    if rows is None:
        work = executed * ((alg_cin or Cin) / Cin) * ((alg_cout or Cout) / Cout) / (stride * stride if mode == MODE_TCONV else 1)
    if mode == MODE_TCONV and stride == 2 and dil == 1 and rows is None and Cin % (16 if x.dtype == torch.float32 else 32) == 0 \
            and Hout % 2 == 0 and Wout % 2 == 0:
        executed /= 4.0                         # phase-decomposed walk (csrc/conv_igemm.hip: tconv_phased): only the taps that exist
    tag = (_DT_TAG[x.dtype], mode, Cout, R * S * Cin, M, executed, rows is not None)
    if mode != MODE_GATHER and 0 < M <= 8192 and Cout >= 64 and rows is None:
        # deep layers with few rows: the library may split K over several blocks per tile (mg_conv_fprop_ws)
        need = _fprop_workspace_fn()(ctypes.byref(p))
        if need > 0:
            ws = _wgrad_workspace(need, x.device)
            hip.call('mg_conv_fprop_ws', ctypes.byref(p), hip.ptr(ws), ctypes.c_long(need), hip.stream(), work=work, tag=tag)
            return out
    hip.call('mg_conv_fprop', ctypes.byref(p), hip.stream(), work=work, tag=tag)
    return out


_FPROP_WS_FN = []


def _fprop_workspace_fn():
    if not _FPROP_WS_FN:
        fn = hip.lib().mg_conv_fprop_workspace
        fn.restype = ctypes.c_long
        _FPROP_WS_FN.append(fn)
    return _FPROP_WS_FN[0]


def conv_wgrad(x, dy, *, cout, mode=MODE_CONV, N=1, Hin=1, Win=1, Hout=1, Wout=1, R=1, S=1, stride=1, pad=0, dil=1,
               M=None, nbr=None, yoff=0, out=None, out_dtype=torch.float32, rows=None, alg_cin=None, alg_cout=None, park=None, xf=None, park_ws=None):
    """dW[co, tap, ci] = sum_m dY[m, co] X[src(m,tap), ci]; (Cout, R*S, Cin) accumulated in fp32 and written as
    `out_dtype` (fp32 / bf16 -- the converting reduce saves a separate cast pass). `dy` may be a channel slice (yoff) of
    a wider buffer.

    `park` (a list): the "row-split slabs -> dW" reduction is not launched; its descriptor and the slabs are appended to the list and the
    returned dW is NOT valid until wgrad_reduce_batched(park) has run on this stream (one launch for all the parked layers).
    `park_ws(floats, device)` -> where this call's slabs go (functional._UseGroup: the calls sharing one weight write them side by side)."""
    Cin = x.shape[-1]
    if mode == MODE_GATHER:
        M = nbr.shape[0] if M is None else M
    else:
        M = N * Hout * Wout
    assert dy.dtype == x.dtype
    p = _conv_params(x, None, dy, mode, N, Hin, Win, Hout, Wout, R, S, stride, pad, dil, M, Cin, cout, nbr, yoff=yoff)
    if out is not None:
        out_dtype = out.dtype
    p.dw_dtype = hip.code_of(out_dtype)
    p.m_dev = hip.ptr(rows)
    _set_xf(p, xf, Cin)                        # the x operand is a raw conv output: BatchNorm + activation applied in flight
    lib = hip.lib()
    lib.mg_conv_wgrad_workspace.restype = ctypes.c_long
    need = lib.mg_conv_wgrad_workspace(ctypes.byref(p)) if M > 0 else 0
    if out is None:
        # with a workspace (or a single row split) dW is overwritten; only the (rare) atomic fallback needs zeros
        out = torch.empty((cout, R * S, Cin), dtype=out_dtype, device=x.device) if M > 0 else \
            torch.zeros((cout, R * S, Cin), dtype=out_dtype, device=x.device)
    p.stats = hip.ptr(out)
    executed = 2.0 * M * cout * R * S * Cin
    work = None if rows is not None else executed * ((alg_cin or Cin) / Cin) * ((alg_cout or cout) / cout) / (stride * stride if mode == MODE_TCONV else 1)
    tag = (_DT_TAG[x.dtype], mode, cout, R * S * Cin, M, executed, rows is not None)
    if park is not None and need > 0:
        # this layer's own slabs: they live until the batched reduction
        ws = torch.empty(int(need), dtype=torch.float32, device=x.device) if park_ws is None else park_ws(int(need), x.device)
        d = hip.WgradParked()
        hip.call('mg_conv_wgrad_park', ctypes.byref(p), hip.ptr(ws), ctypes.c_long(need), ctypes.byref(d), hip.stream(), work=work, tag=tag)
        if d.splits > 0:
            park.append((d, ws, out))
        return out
    ws = _wgrad_workspace(need, x.device) if need > 0 else None
    hip.call('mg_conv_wgrad_ws', ctypes.byref(p), hip.ptr(ws), ctypes.c_long(need if ws is not None else 0), hip.stream(), work=work, tag=tag)
    return out


def sum_k(ts, out=None):
    """Sum of up to 16 same-shaped contiguous fp32 device tensors in ONE launch, added in list order (mg_sum_k). `out`: a contiguous fp32 tensor of
    that size to write to (it may not be one of the terms)."""
    n = ts[0].numel()
    if out is None:
        out = torch.empty_like(ts[0])
    hip.need_cuda(*ts)
    arr = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    if ts[0].dtype == torch.float32:
        hip.call('mg_sum_k', arr, c_int(len(ts)), ctypes.c_long(n), hip.ptr(out), hip.stream())
    else:                                                         # 16-bit terms: fp32 sum in list order, rounded once
        hip.call('mg_sum_k_t', arr, c_int(len(ts)), ctypes.c_long(n), hip.ptr(out), c_int(hip.dtype_code(ts[0])), hip.stream())
    return out


def spatial_mean(x, N, HW, mode=0):
    """AdaptiveAvgPool2d(1) over NHWC rows (mg_spatial_mean). mode 0: x (N, HW, C) -> (N, C) mean; 1: x = dy (N, C) -> dx (N, HW, C) = dy / HW;
    2: x (N, HW, C) -> (N, C) sum. In modes 0 / 2 x may be a channel slice of a wider (N, HW, C') map (uniform row stride): read in place."""
    C = x.shape[-1]
    ld = C
    if mode != 1 and x.dim() == 3 and x.stride(2) == 1 and x.stride(0) == HW * x.stride(1) and x.stride(1) >= C:
        ld = x.stride(1)
    else:
        x = x.contiguous()
    hip.need_cuda(x)
    out = torch.empty((N, HW, C) if mode == 1 else (N, C), dtype=x.dtype, device=x.device)
    hip.call('mg_spatial_mean', hip.ptr(x), hip.ptr(out), c_int(hip.dtype_code(x)), c_int(N), c_int(HW), c_int(C), c_int(mode), c_int(ld), hip.stream())
    return out


COPY_K = __import__('os').environ.get('MAGGIE_COPY_K', '1') != '0'      # 0: torch._foreach_copy_ (one copy kernel per tensor) -- A/B switch


def copy_k(dsts, srcs):
    """dst.copy_(src) for lists of device tensors, 16 per launch (mg_copy_k); pairs that are not plain byte copies (different dtype / shape, a
    non-contiguous side) go through torch._foreach_copy_."""
    plain, rest_d, rest_s = [], [], []
    for d, s_ in zip(dsts, srcs):
        if COPY_K and d.dtype == s_.dtype and d.shape == s_.shape and d.is_contiguous() and s_.is_contiguous() and d.is_cuda and s_.is_cuda:
            if d.numel() and d.data_ptr() != s_.data_ptr():
                plain.append((d, s_))
        else:
            rest_d.append(d)
            rest_s.append(s_)
    for i in range(0, len(plain), 16):
        grp = plain[i:i + 16]
        n = len(grp)
        sp = (ctypes.c_void_p * n)(*[s_.data_ptr() for _, s_ in grp])
        dp = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in grp])
        nb = (ctypes.c_long * n)(*[d.numel() * d.element_size() for d, _ in grp])
        hip.call('mg_copy_k', sp, dp, nb, c_int(n), hip.stream())
    if rest_d:
        torch._foreach_copy_(rest_d, rest_s)


def wgrad_reduce_batched(park):
    """Run every parked slab reduction of `park` (conv_wgrad(park=...)) in one launch per 64 layers and empty the list."""
    n = len(park)
    if n == 0:
        return
    arr = (hip.WgradParked * n)()
    cur = torch.cuda.current_stream()
    for i, (d, ws, out) in enumerate(park):
        arr[i] = d
        # the GEMM may have run on another stream (the encoder's shortcut branches: functional.on_side_lane); autograd has already ordered this
        # stream behind it -- the allocator must learn that slabs and dW are read / written HERE as well
        ws.record_stream(cur)
        out.record_stream(cur)
    hip.call('mg_wgrad_reduce_batched', arr, c_int(n), hip.stream())
    del park[:]


_WS = {}


def _wgrad_workspace(floats, device):
    """One growing fp32 scratch buffer per device and stream (kernels on one stream execute in order, so reuse is safe)."""
    if torch.cuda.is_current_stream_capturing():
        # graph-owned scratch: the shared buffer may be re-allocated after the capture
        return torch.empty(int(floats), dtype=torch.float32, device=device)
    key = (device, hip.stream().value)
    buf = _WS.get(key)
    if buf is None or buf.numel() < floats:
        buf = torch.empty(int(floats * 1.25) + 1024, dtype=torch.float32, device=device)
        _WS[key] = buf
    return buf


# ------------------------------------------------------------------------------------------------------------------
# BatchNorm / row-wise kernels
# ------------------------------------------------------------------------------------------------------------------
from .hip import RowwiseParams, c_int, c_float, c_long  # noqa: E402


STAT_REPLICAS = 32


def stat_rows():
    """Rows of a BatchNorm statistics scratch filled by the column-statistics kernels: 32 replicas that only spread same-address atomics, or --
    deterministic mode -- 1024 >= the row-block count, one row per row block (MG_DET_STAT_ROWS, csrc/common.h)."""
    return 1024 if hip.DETERMINISTIC else STAT_REPLICAS


def conv_stat_rows(M, N=1, Hout=1, Wout=1):
    """Rows of the statistics buffer a conv epilogue adds into: in deterministic mode at least the number of output tiles of ANY kernel form
    (spatial halo tiles of >= 4 x 16 pixels, row tiles of >= 64 rows, the four padded phases of a stride-2 transposed walk), so that every
    word receives exactly one addition; rows nobody writes stay zero."""
    key = (hip.DETERMINISTIC, M, N, Hout, Wout)
    rows = _STAT_ROWS.get(key)
    if rows is None:                                               # asked of the library, which owns the tile shapes (mg_conv_stat_rows)
        fn = hip.lib().mg_conv_stat_rows
        fn.restype = ctypes.c_int
        rows = _STAT_ROWS[key] = int(fn(c_int(M), c_int(N), c_int(Hout), c_int(Wout)))
    return rows


_STAT_ROWS = {}


def ACC(n, device, dtype=torch.float32):
    """Accumulator the callee clears itself; functional.py re-points this at its zero arena (ZeroArena.acc)."""
    return torch.empty(n, dtype=dtype, device=device)


def colstats(x, stats=None):
    """stats[stat_rows()][2C] (fp32) += column sum / sum of squares of x (M, C), spread over the rows."""
    M, C = x.shape[0], x.shape[-1]
    if stats is None:
        stats = torch.zeros((stat_rows(), 2 * C), dtype=torch.float32, device=x.device)
    assert stats.numel() >= stat_rows() * 2 * C
    hip.need_cuda(x, stats)
    hip.call('mg_colstats', hip.ptr(x), c_int(hip.dtype_code(x)), c_int(M), c_int(C), c_int(_ld(x)), hip.ptr(stats), hip.stream())
    return stats


def stat_rows_sum(stats):
    """[nrep][n] fp32 statistics rows -> [n]: their sum in the library's fixed order (mg_stat_rows_sum)."""
    if stats.dim() == 1:
        return stats
    nrep, n = stats.shape
    out = ACC(n, stats.device)                 # cleared by the callee; inside a capture a slice of the graph's zero arena (no fill launch)
    hip.need_cuda(stats)
    hip.call('mg_stat_rows_sum', hip.ptr(stats.contiguous()), c_int(nrep), c_int(n), hip.ptr(out), hip.stream())
    return out


def bias_act_bwd(dy, y, want_db, rows=None):
    """g = dy * (y > 0) (y None: g = dy) and db = g.sum(0) in fp32 (None unless want_db) -- one pass (mg_bias_act_bwd).
    `rows`: device row count (int32 tensor), dy.shape[0] is then the capacity."""
    M, C = dy.shape
    g = torch.empty_like(dy) if y is not None else dy
    db = ACC(C, dy.device) if want_db else None
    hip.call('mg_bias_act_bwd_dev', hip.ptr(dy), hip.ptr(y), hip.ptr(g if y is not None else None), c_int(hip.dtype_code(dy)), c_int(M), c_int(C),
             hip.ptr(db), hip.ptr(rows), hip.stream())
    return g, db


def colstats_centered(x, stats=None, have_sum=False):
    """Exact two-pass statistics (for small row counts): stats[0:C] = sum, stats[C:2C] = sum (x - mean)^2.
    `stats` must be zero on entry (a slice of the per-step zero arena) -- or, with have_sum, already hold the column sums in
    stats[0:C] (accumulated by the producing conv's epilogue) and zeros in stats[C:2C]."""
    M, C = x.shape[0], x.shape[-1]
    if stats is None:
        stats = torch.zeros(2 * C, dtype=torch.float32, device=x.device)
    hip.need_cuda(x)
    hip.call('mg_colstats_centered', hip.ptr(x), c_int(hip.dtype_code(x)), c_int(M), c_int(C), c_int(_ld(x)), hip.ptr(stats),
             c_int(int(have_sum)), hip.stream())
    return stats


def bn_finalize(stats, count, gamma, beta, running_mean, running_var, momentum, eps, count_ptr=None, centered=False):
    """-> scale, shift, mean, invstd (each fp32 [C]); updates the running statistics in place when given."""
    nrep = stats.shape[0] if stats.dim() == 2 else 1
    C = stats.shape[-1] // 2
    out = torch.empty((4, C), dtype=torch.float32, device=stats.device)
    hip.need_cuda(stats, gamma, beta, running_mean, running_var)
    hip.call('mg_bn_finalize', hip.ptr(stats), c_int(nrep), hip.ptr(count_ptr), c_float(float(count)), c_int(C), c_int(int(centered)), hip.ptr(gamma), hip.ptr(beta),
             hip.ptr(running_mean), hip.ptr(running_var), c_float(momentum), c_float(eps), hip.ptr(out[0]), hip.ptr(out[1]),
             hip.ptr(out[2]), hip.ptr(out[3]), hip.stream())
    return out[0], out[1], out[2], out[3]


def bn_fold(gamma, beta, running_mean, running_var, eps):
    C = running_mean.numel()
    out = torch.empty((2, C), dtype=torch.float32, device=running_mean.device)
    hip.need_cuda(gamma, beta, running_mean, running_var)
    hip.call('mg_bn_fold', c_int(C), hip.ptr(gamma), hip.ptr(beta), hip.ptr(running_mean), hip.ptr(running_var), c_float(eps),
             hip.ptr(out[0]), hip.ptr(out[1]), hip.stream())
    return out[0], out[1]


def _rowwise(x, M, C):
    p = RowwiseParams()
    p.dtype, p.M, p.C = hip.dtype_code(x), M, C
    p.x, p.ldx = hip.ptr(x), _ld(x)
    return p


def affine_act(x, scale=None, shift=None, res=None, res_mode=1, res2=None, act=ACT_NONE, slope=0.2, H=1, W=1, out=None, yoff=0, rows=None):
    """y = act(x*scale + shift + res) + res2 over rows x channels."""
    M, C = x.shape[0], x.shape[-1]
    if out is None:
        out = torch.empty((M, C), dtype=x.dtype, device=x.device)
    hip.need_cuda(x, scale, shift, res, res2, out)
    p = _rowwise(x, M, C)
    p.y, p.ldy, p.yoff = hip.ptr(out), _ld(out), yoff
    p.scale, p.shift = hip.ptr(scale), hip.ptr(shift)
    p.res, p.ldr, p.res_mode = hip.ptr(res), (_ld(res) if res is not None else 0), (res_mode if res is not None else 0)
    p.res2, p.ldr2 = hip.ptr(res2), (_ld(res2) if res2 is not None else 0)
    p.act, p.slope, p.H, p.W = act, slope, H, W
    p.m_dev = hip.ptr(rows)
    hip.call('mg_affine_act', ctypes.byref(p), hip.stream())
    return out


def bn_backward(dy, y, x, scale, mean, invstd, count, act=ACT_NONE, slope=0.2, want_dres=False, mask_x_pos=False, yoff=0,
                count_ptr=None, sums=None, want_dx=True, reduce_only=False, apply_only=False, rows=None, shift=None):
    """BatchNorm(+activation) backward over rows x channels.
    Returns (dx, dres, sums) with sums = [sum g, sum g*xhat] (= dbeta, dgamma). `shift` with y None: the activation output was never stored
    (operand-path BatchNorm); its sign is re-formed from x * scale + shift (mask_from_x)."""
    M, C = x.shape[0], x.shape[-1]
    hip.need_cuda(dy, y, x, scale, mean, invstd)
    p = _rowwise(x, M, C)
    p.dy, p.lddy = hip.ptr(dy), _ld(dy)
    p.y, p.ldy, p.yoff = hip.ptr(y), (_ld(y) if y is not None else 0), yoff
    if y is None and shift is not None and act != ACT_NONE:
        p.y, p.ldy, p.yoff, p.mask_from_x, p.shift = hip.ptr(x), C, 0, 1, hip.ptr(shift)
    p.scale, p.mean, p.invstd = hip.ptr(scale), hip.ptr(mean), hip.ptr(invstd)
    if sums is None:
        sums = torch.zeros(2 * C, dtype=torch.float32, device=x.device)
    p.sums = hip.ptr(sums)
    p.count, p.count_ptr = float(count), hip.ptr(count_ptr)
    p.act, p.slope, p.mask_x_pos = act, slope, int(mask_x_pos)
    p.m_dev = hip.ptr(rows)
    dx = dres = None
    if not apply_only:
        hip.call('mg_bn_bwd_reduce', ctypes.byref(p), hip.stream())
    if reduce_only:
        return None, None, sums
    if want_dx:
        dx = torch.empty((M, C), dtype=x.dtype, device=x.device)
        p.dx, p.lddx = hip.ptr(dx), C
    if want_dres:
        dres = torch.empty((M, C), dtype=x.dtype, device=x.device)
        p.dres, p.lddres = hip.ptr(dres), C
    hip.call('mg_bn_bwd_apply', ctypes.byref(p), hip.stream())
    return dx, dres, sums


def bn_bwd_apply_linked(g, x, outs, sums_rep, count, mask_x_pos=False, rows=None):
    """BatchNorm backward, apply pass only: `g` = dy * act'(y) and `sums_rep` ([nrep, 2C] replicas of sum g | sum g * xhat) were produced by
    the consumer conv's data-gradient epilogue (conv_fprop(bnb=...)). -> dx, sums [2C] = dbeta | dgamma. The residual gradient of the layer
    IS g."""
    M, C = x.shape[0], x.shape[-1]
    p = _rowwise(x, M, C)
    p.dy, p.lddy = hip.ptr(g), _ld(g)
    base = outs.data_ptr()
    p.scale, p.mean, p.invstd = ctypes.c_void_p(base), ctypes.c_void_p(base + 8 * C), ctypes.c_void_p(base + 12 * C)
    p.count, p.mask_x_pos = float(count), int(mask_x_pos)
    p.m_dev = hip.ptr(rows)
    dx = torch.empty((M, C), dtype=x.dtype, device=x.device)
    p.dx, p.lddx = hip.ptr(dx), C
    sums = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    hip.call('mg_bn_bwd_apply_linked', ctypes.byref(p), hip.ptr(sums_rep), c_int(sums_rep.shape[0] if sums_rep.dim() == 2 else 1), hip.ptr(sums),
             hip.stream())
    return dx, sums


def stats_ws_floats(C, exact=False):
    """Size of the statistics scratch mg_bn_train_fwd wants: [2C] for the exact two-pass variance, [32][2C] replicas otherwise -- in
    deterministic mode always [1024][2C] (one row per row block; the exact form then only runs inside one workgroup, without scratch)."""
    if hip.DETERMINISTIC:
        return 2 * stat_rows() * C
    return (2 if exact else 2 * STAT_REPLICAS) * C


def bn_train_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, act, slope, res=None, res_mode=1, H=1, W=1, stats=None,
                 exact=False, stats_ws=None, rows=None, count_mult=1, apply=True):
    """Training BatchNorm forward in ONE C call (statistics -> finalize -> apply): -> y, outs = scale|shift|mean|invstd (4C).
    `stats_ws`: zeroed scratch of stats_ws_floats(C, exact) floats for the statistics; None: allocated and zeroed here.
    apply=False: statistics and finalize only (y is None): the consumer applies scale | shift to its operand in flight (conv_fprop(xf=...))."""
    M, C = x.shape[0], x.shape[-1]
    y = torch.empty((M, C), dtype=x.dtype, device=x.device) if apply else None
    outs = torch.empty(4 * C, dtype=torch.float32, device=x.device)
    zeroed = stats_ws is not None
    two_pass = exact or rows is not None                      # a device row count always takes the two-pass form (outside deterministic mode)
    if stats_ws is not None:
        assert stats_ws.numel() >= stats_ws_floats(C, two_pass), (stats_ws.numel(), C, exact)
    if stats_ws is None and stats is None:
        stats_ws = torch.empty(stats_ws_floats(C, two_pass), dtype=torch.float32, device=x.device)
    p = _rowwise(x, M, C)
    p.y, p.ldy, p.yoff = hip.ptr(y), C, 0
    if res is not None:
        p.res, p.ldr, p.res_mode = hip.ptr(res), _ld(res), res_mode
    p.act, p.slope, p.H, p.W = act, slope, H, W
    p.m_dev = hip.ptr(rows)                    # device row count: always the exact two-pass variance over the live rows
    p.count_mult = int(count_mult)             # every row stands for this many samples of the reference's BatchNorm (unbiased running variance)
    srows = 0 if stats is None else (stats.shape[0] if stats.dim() == 2 else 1)
    hip.call('mg_bn_train_fwd', ctypes.byref(p), hip.ptr(stats_ws), c_int(int(zeroed)), hip.ptr(outs), hip.ptr(stats), c_int(srows), c_int(int(exact)),
             hip.ptr(gamma), hip.ptr(beta), hip.ptr(running_mean), hip.ptr(running_var), c_float(momentum), c_float(eps), hip.stream())
    return y, outs


def bn_train_bwd(dy, y, x, outs, act, slope, want_dres=False, mask_x_pos=False, sums=None, rows=None):
    """Training BatchNorm backward in ONE C call (reduce + apply): -> dx, dres, sums [2C] = dbeta | dgamma.
    `sums`: zeroed [2C] accumulator (None: allocated and zeroed here)."""
    M, C = x.shape[0], x.shape[-1]
    p = _rowwise(x, M, C)
    p.dy, p.lddy = hip.ptr(dy), _ld(dy)
    # y None: the activation output was never stored (operand-path BatchNorm): its sign is re-formed from x (mask_from_x; `y` then only has to be a
    # valid address)
    p.y, p.ldy, p.yoff, p.mask_from_x = hip.ptr(x if y is None else y), C, 0, int(y is None)
    base = outs.data_ptr()
    p.scale, p.shift, p.mean, p.invstd = ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * C), ctypes.c_void_p(base + 8 * C), ctypes.c_void_p(base + 12 * C)
    zeroed = sums is not None
    if sums is None:
        sums = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    p.sums, p.count = hip.ptr(sums), float(M)
    p.act, p.slope, p.mask_x_pos = act, slope, int(mask_x_pos)
    p.m_dev = hip.ptr(rows)
    dx = torch.empty((M, C), dtype=x.dtype, device=x.device)
    p.dx, p.lddx = hip.ptr(dx), C
    dres = None
    if want_dres:
        dres = torch.empty((M, C), dtype=x.dtype, device=x.device)
        p.dres, p.lddres = hip.ptr(dres), C
    hip.call('mg_bn_train_bwd', ctypes.byref(p), c_int(int(zeroed)), hip.stream())
    return dx, dres, sums


def pool2x2(x, op, N, Ho, Wo):
    """x: NHWC rows. op 0 avg-pool, 1 sum-pool (input 2Ho x 2Wo); 2 = 0.25*nearest-up, 3 = nearest-up (input Ho/2 x Wo/2)."""
    C = x.shape[-1]
    assert x.is_contiguous()
    out = torch.empty((N * Ho * Wo, C), dtype=x.dtype, device=x.device)
    hip.need_cuda(x)
    hip.call('mg_pool2x2', hip.ptr(x), hip.ptr(out), c_int(hip.dtype_code(x)), c_int(op), c_int(N), c_int(Ho), c_int(Wo), c_int(C),
             hip.stream())
    return out


# ------------------------------------------------------------------------------------------------------------------
# region ops on bit planes
# ------------------------------------------------------------------------------------------------------------------
LOWER_THRES = 1.0 / 255.0
UPPER_THRES = 254.0 / 255.0


def words(W):
    return (W + 63) // 64


def bits_pack(a, mode=0, lo=LOWER_THRES, hi=UPPER_THRES):
    """a: (..., H, W) fp32 or uint8 planes (contiguous) -> int64 bit planes (P, H, Ww)."""
    H, W = a.shape[-2:]
    P = a.numel() // (H * W)
    assert a.is_contiguous()
    hip.need_cuda(a)
    dt = hip.U8 if a.dtype == torch.uint8 else hip.F32
    assert dt == hip.U8 or a.dtype == torch.float32
    bits = torch.empty((P, H, words(W)), dtype=torch.int64, device=a.device)
    hip.call('mg_bits_pack', hip.ptr(a), c_int(dt), hip.ptr(bits), c_int(P), c_int(H), c_int(W), c_int(mode), c_float(lo), c_float(hi),
             hip.stream())
    return bits


def bits_select(bits, a, b, W):
    """out = bit ? a : b over fp32 planes shaped like a (numel = P*H*W of the bit planes)."""
    P, H, Ww = bits.shape
    hip.need_cuda(bits, a, b)
    out = torch.empty_like(a)
    hip.call('mg_bits_select', hip.ptr(bits), hip.ptr(a), hip.ptr(b), hip.ptr(out), c_int(P), c_int(H), c_int(W), hip.stream())
    return out


def bits_select_bwd(bits, dy, W, want_a=True, want_b=True):
    P, H, Ww = bits.shape
    da = torch.empty_like(dy) if want_a else None
    db = torch.empty_like(dy) if want_b else None
    hip.call('mg_bits_select_bwd', hip.ptr(bits), hip.ptr(dy), hip.ptr(da), hip.ptr(db), c_int(P), c_int(H), c_int(W), hip.stream())
    return da, db


def bits_unpack_u8(bits, W, shape=None):
    P, H, Ww = bits.shape
    out = torch.empty((P, H, W), dtype=torch.uint8, device=bits.device)
    hip.call('mg_bits_unpack_u8', hip.ptr(bits), hip.ptr(out), c_int(P), c_int(H), c_int(W), hip.stream())
    return out if shape is None else out.view(*shape)


def bits_unpack_f32(bits, W, shape=None):
    P, H, Ww = bits.shape
    out = torch.empty((P, H, W), dtype=torch.float32, device=bits.device)
    hip.call('mg_bits_unpack_f32', hip.ptr(bits), hip.ptr(out), c_int(P), c_int(H), c_int(W), hip.stream())
    return out if shape is None else out.view(*shape)


def bits_dilate(bits, W, width=None, widths=None, andmask=None):
    """Binary dilation with OpenCV's ellipse of the given width (scalar) or per-plane device int32 `widths`."""
    P, H, Ww = bits.shape
    out = torch.empty_like(bits)
    if widths is not None:
        assert widths.dtype == torch.int32 and widths.numel() == P and widths.is_cuda
    hip.call('mg_bits_dilate', hip.ptr(bits), hip.ptr(out), hip.ptr(andmask), c_int(P), c_int(H), c_int(W), hip.ptr(widths),
             c_int(0 if width is None else int(width)), hip.stream())
    return out


def bits_downsample(bits, Wf):
    P, Hf, _ = bits.shape
    Hc, Wc = (Hf - 1) // 2 + 1, (Wf - 1) // 2 + 1
    out = torch.empty((P, Hc, words(Wc)), dtype=torch.int64, device=bits.device)
    hip.call('mg_bits_downsample', hip.ptr(bits), hip.ptr(out), c_int(P), c_int(Hf), c_int(Wf), hip.stream())
    return out, Hc, Wc


def bits_rank(bits, W):
    """-> rowoff (P*H+1,) int32 [last = number of active sites], wordoff (P,H,Ww) int32."""
    P, H, Ww = bits.shape
    tmp = torch.empty((P * H,), dtype=torch.int32, device=bits.device)
    rowoff = torch.empty((P * H + 1,), dtype=torch.int32, device=bits.device)
    wordoff = torch.empty((P, H, Ww), dtype=torch.int32, device=bits.device)
    hip.call('mg_bits_rank', hip.ptr(bits), c_int(P), c_int(H), c_int(W), hip.ptr(tmp), hip.ptr(rowoff), hip.ptr(wordoff), hip.stream())
    return rowoff, wordoff


def bits_truncate_(bits, wordoff, W, cap, count, overflow):
    """In place: at most `cap` sites of the level stay active (rank order); `count` (int32 [1] view of the rank table) is clamped and the sticky
    int32 [1] `overflow` flag raised when sites were dropped."""
    P, H, Ww = bits.shape
    hip.call('mg_bits_truncate', hip.ptr(bits), hip.ptr(wordoff), c_int(P), c_int(H), c_int(W), c_int(int(cap)), hip.ptr(count), hip.ptr(overflow),
             hip.stream())


def bits_coords(bits, wordoff, W, R):
    P, H, Ww = bits.shape
    coords = torch.empty((R, 3), dtype=torch.int32, device=bits.device)
    if R > 0:
        hip.call('mg_bits_coords', hip.ptr(bits), hip.ptr(wordoff), c_int(P), c_int(H), c_int(W), hip.ptr(coords), hip.stream())
    return coords


def gather_table(coords, ksize, kind, src_bits, src_wordoff, Hs, Ws, rows=None):
    R = coords.shape[0]
    nbr = torch.empty((R, ksize * ksize), dtype=torch.int32, device=coords.device)
    if R > 0:
        hip.call('mg_gather_table_dev', hip.ptr(coords), c_int(R), c_int(ksize), c_int(kind), hip.ptr(src_bits), hip.ptr(src_wordoff),
                 c_int(Hs), c_int(Ws), hip.ptr(nbr), hip.ptr(rows), hip.stream())
    return nbr


# ------------------------------------------------------------------------------------------------------------------
# dense <-> sparse, alpha planes, input packing
# ------------------------------------------------------------------------------------------------------------------
def gather_rows(dense, coords, n_i, mul=None, out=None, yoff=0, rows=None):
    """dense: (N, Hd, Wd, C) NHWC contiguous; coords (R,3) at that level; mul: fp32 (N, n_tok, C) or None."""
    N, Hd, Wd, C = dense.shape
    R = coords.shape[0]
    assert dense.is_contiguous() and coords.is_contiguous() and coords.dtype == torch.int32
    if out is None:
        out = torch.empty((R, C), dtype=dense.dtype, device=dense.device)
    hip.need_cuda(dense, coords, mul, out)
    if mul is not None:
        assert mul.dtype == torch.float32 and mul.is_contiguous() and mul.shape[-1] == C
    hip.call('mg_gather_rows_dev', hip.ptr(dense), c_int(hip.dtype_code(dense)), hip.ptr(coords), c_int(R), c_int(n_i), c_int(Hd), c_int(Wd),
             c_int(C), hip.ptr(mul), c_int(mul.shape[1] if mul is not None else 0), hip.ptr(out), c_int(_ld(out)), c_int(yoff), hip.ptr(rows),
             hip.stream())
    return out


def gather_rows_bwd(dout, coords, n_i, dense_shape, mul=None, dense=None, yoff=0, want_ddense=True, want_dmul=False, rows=None):
    N, Hd, Wd, C = dense_shape
    R = coords.shape[0]
    if hip.DETERMINISTIC and not want_ddense and want_dmul and mul is not None and 256 % (C // (8 if dout.element_size() == 2 else 4)) == 0:
        # bit-reproducible form: per-plane row ranges, fixed-order sums (csrc/sparse.hip: gather_rows_dmul_det_kernel)
        dmul = torch.empty_like(mul)
        hip.need_cuda(dout, coords, dense, mul)
        assert dense.is_contiguous() and mul.dtype == torch.float32
        hip.call('mg_gather_rows_dmul_det', hip.ptr(dout), c_int(hip.dtype_code(dout)), c_int(_ld(dout)), c_int(yoff), hip.ptr(coords), c_int(R), c_int(n_i),
                 c_int(N), c_int(Hd), c_int(Wd), c_int(C), c_int(mul.shape[1]), hip.ptr(dense), hip.ptr(dmul), hip.ptr(rows), hip.stream())
        return None, dmul
    det_dmul = None
    if hip.DETERMINISTIC and want_ddense and want_dmul and mul is not None and 256 % (C // (8 if dout.element_size() == 2 else 4)) == 0:
        # both asked for: the multiplier gradient in its reproducible form; the dense gradient below still scatters with atomics (rows of several
        # instance planes meet at one pixel) -- the product uses gather_rows_bwd_dense for it, which has none
        _, det_dmul = gather_rows_bwd(dout, coords, n_i, dense_shape, mul=mul, dense=dense, yoff=yoff, want_ddense=False, want_dmul=True, rows=rows)
        want_dmul = False
    ddense = torch.zeros(dense_shape, dtype=torch.float32, device=dout.device) if want_ddense else None
    dmul = torch.zeros_like(mul) if (want_dmul and mul is not None) else None
    hip.call('mg_gather_rows_bwd_dev', hip.ptr(dout), c_int(hip.dtype_code(dout)), c_int(_ld(dout)), c_int(yoff), hip.ptr(coords), c_int(R),
             c_int(n_i), c_int(Hd), c_int(Wd), c_int(C), hip.ptr(mul), c_int(mul.shape[1] if mul is not None else 0), hip.ptr(dense),
             hip.ptr(ddense), hip.ptr(dmul), hip.ptr(rows), hip.stream())
    return ddense, (det_dmul if det_dmul is not None else dmul)


def gather_rows_bwd_dense(dout, bits, wordoff, n_i, dense_shape, mul=None, yoff=0):
    """Atomic-free input gradient of gather_rows: (N, Hd, Wd, C) in dout's dtype."""
    N, Hd, Wd, C = dense_shape
    ddense = torch.empty(dense_shape, dtype=dout.dtype, device=dout.device)
    hip.call('mg_gather_rows_bwd_dense', hip.ptr(dout), c_int(hip.dtype_code(dout)), c_int(_ld(dout)), c_int(yoff), hip.ptr(bits), hip.ptr(wordoff),
             c_int(n_i), c_int(N), c_int(Hd), c_int(Wd), c_int(C), hip.ptr(mul), c_int(mul.shape[1] if mul is not None else 0), hip.ptr(ddense),
             hip.stream())
    return ddense


def scatter_plane(vals, col, coords, P, H, W, fill=-99.0, rows=None):
    R = coords.shape[0]
    plane = torch.empty((P, H, W), dtype=torch.float32, device=coords.device)
    hip.call('mg_scatter_plane_dev', hip.ptr(vals), c_int(hip.dtype_code(vals) if vals is not None else 0),
             c_int(_ld(vals) if vals is not None and R > 0 else 1), c_int(col), hip.ptr(coords), c_int(R), c_int(P), c_int(H), c_int(W),
             c_float(fill), hip.ptr(plane), hip.ptr(rows), hip.stream())
    return plane


def gather_plane(plane, coords, like, width=1, rows=None):
    """-> (R, width) tensor of dtype `like` with the plane values at the sites in column 0 (the other columns zero)."""
    R = coords.shape[0]
    P, H, W = plane.shape
    out = torch.empty((R, width), dtype=like, device=plane.device)
    hip.call('mg_gather_plane_dev', hip.ptr(plane), hip.ptr(coords), c_int(R), c_int(H), c_int(W), hip.ptr(out),
             c_int(hip.code_of(like)), c_int(width), c_int(0), hip.ptr(rows), c_int(int(width > 1)), hip.stream())
    return out


def mask_embed(image, masks, table, dtype):
    """image (N,3,H,W) fp32, masks (N,n_m,Hm,Wm) fp32, table (n_m+1, n_embed) fp32 -> (N,H,W,8) NHWC of `dtype`."""
    N, _, H, W = image.shape
    n_m, Hm, Wm = masks.shape[1:]
    assert image.is_contiguous() and masks.is_contiguous() and image.dtype == torch.float32 and masks.dtype == torch.float32
    hip.need_cuda(image, masks, table)
    out = torch.empty((N, H, W, 8), dtype=dtype, device=image.device)
    hip.call('mg_mask_embed', hip.ptr(image), hip.ptr(masks), hip.ptr(table), c_int(N), c_int(H), c_int(W), c_int(n_m), c_int(Hm), c_int(Wm),
             c_int(table.shape[1]), hip.ptr(out), c_int(hip.dtype_code(out)), hip.stream())
    return out


def mask_embed_bwd(dx, masks, table_shape):
    N, H, W, _ = dx.shape
    n_m, Hm, Wm = masks.shape[1:]
    dtable = torch.zeros(table_shape, dtype=torch.float32, device=dx.device)
    hip.call('mg_mask_embed_bwd', hip.ptr(dx), c_int(hip.dtype_code(dx)), hip.ptr(masks), c_int(N), c_int(H), c_int(W), c_int(n_m), c_int(Hm),
             c_int(Wm), c_int(table_shape[1]), hip.ptr(dtable), hip.stream())
    return dtable


def upsample_tanh(x, strides, N, C, h, w, scale, apply_tanh=True, pscale=None, any_nonzero=None):
    """x addressed as (n, c, y, x) with element strides -> fp32 (N, C, h*scale, w*scale) = (tanh(bilinear(x)) + 1)/2.
    `pscale`: fp32 [N*C] 0 / 1 plane scale; `any_nonzero`: zeroed int32 [1], set to 1 when any output element is non-zero."""
    out = torch.empty((N, C, h * scale, w * scale), dtype=torch.float32, device=x.device)
    sn, sc, sy, sx = strides
    hip.need_cuda(x, pscale, any_nonzero)
    assert pscale is None or (pscale.dtype == torch.float32 and pscale.numel() == N * C and pscale.is_contiguous())
    hip.call('mg_upsample_tanh_ex', hip.ptr(x), c_int(hip.dtype_code(x)), c_long(sn), c_long(sc), c_long(sy), c_long(sx), c_int(N), c_int(C),
             c_int(h), c_int(w), c_int(scale), c_int(int(apply_tanh)), hip.ptr(out), hip.ptr(pscale), hip.ptr(any_nonzero), hip.stream())
    return out


def upsample_tanh_bwd(dout, out, strides, N, C, h, w, scale, din, apply_tanh=True, pscale=None):
    """din: fp32 buffer with the input's strides (pre-zeroed), accumulated atomically."""
    sn, sc, sy, sx = strides
    hip.call('mg_upsample_tanh_bwd_ex', hip.ptr(dout), hip.ptr(out), c_long(sn), c_long(sc), c_long(sy), c_long(sx), c_int(N), c_int(C), c_int(h),
             c_int(w), c_int(scale), c_int(int(apply_tanh)), hip.ptr(din), hip.ptr(pscale), hip.stream())
    return din


def plane_flags(planes, as_float=False):
    """int32 [P]: 1 where a (.., H, W) fp32 plane holds any value > 0 (mg_plane_flags); as_float: fp32 0.0 / 1.0 instead."""
    H, W = planes.shape[-2:]
    P = planes.numel() // (H * W)
    flags = ACC(P, planes.device, torch.float32 if as_float else torch.int32)
    hip.need_cuda(planes)
    assert planes.dtype == torch.float32 and planes.is_contiguous()
    hip.call('mg_plane_flags_ex', hip.ptr(planes), c_int(P), c_int(H * W), hip.ptr(flags), c_int(int(as_float)), hip.stream())
    return flags


# ------------------------------------------------------------------------------------------------------------------
# SpectralNorm weight preparation
# ------------------------------------------------------------------------------------------------------------------
def spectral_norm(w_bar, u, v, transposed, dtype, pad_in):
    """w_bar fp32 (A, B, k, k); u (A), v (B*k*k) updated IN PLACE (one power iteration). Returns
    (w_sn (Cout, k*k, pad_in) in `dtype`, work scratch [.., sigma])."""
    A, B, kh, kw = w_bar.shape
    taps = kh * kw
    cout = B if transposed else A
    hip.need_cuda(w_bar, u, v)
    assert w_bar.dtype == torch.float32 and w_bar.is_contiguous() and u.dtype == torch.float32 and v.dtype == torch.float32
    out = torch.empty((cout, taps, pad_in), dtype=dtype, device=w_bar.device)
    work = torch.empty((B * taps + A + 4,), dtype=torch.float32, device=w_bar.device)
    hip.call('mg_spectral_norm', hip.ptr(w_bar), hip.ptr(u), hip.ptr(v), c_int(A), c_int(B), c_int(taps), c_int(int(transposed)),
             c_int(pad_in), hip.ptr(out), c_int(hip.dtype_code(out)), hip.ptr(work), hip.stream())
    return out, work


def spectral_norm_bwd(G, w_bar, u, v, transposed, work):
    A, B, kh, kw = w_bar.shape
    taps = kh * kw
    pad_in = G.shape[-1]
    G = G.float().contiguous()
    dW = torch.empty_like(w_bar)
    hip.call('mg_spectral_norm_bwd', hip.ptr(G), hip.ptr(w_bar), hip.ptr(u), hip.ptr(v), c_int(A), c_int(B), c_int(taps),
             c_int(int(transposed)), c_int(pad_in), hip.ptr(work), hip.ptr(dW), hip.stream())
    return dW


# ------------------------------------------------------------------------------------------------------------------
# instance-token <-> feature cross attention (fp32)
# ------------------------------------------------------------------------------------------------------------------
def attn_tok_fwd(qk, btab, feat, ids, scale):
    """tokens <- features. qk (B,T,D), btab (B,T,NID), feat (B,L,D) fp32 contiguous, ids (B,L) int32 -> p (B,T,L), ctx (B,T,D)."""
    hip.need_cuda(qk, btab, feat, ids)
    B, L, D = feat.shape
    T, NID = qk.shape[1], btab.shape[2]
    p = torch.empty((B, T, L), dtype=torch.float32, device=feat.device)
    ctx = ACC(B * T * D, feat.device).view(B, T, D)
    hip.call('mg_attn_tok_fwd', hip.ptr(qk), hip.ptr(btab), hip.ptr(feat), hip.ptr(ids), c_int(B), c_int(T), c_int(L), c_int(D), c_int(NID),
             c_float(scale), hip.ptr(p), hip.ptr(ctx), hip.stream())
    return p, ctx


def attn_tok_bwd(p, feat, qk, ids, dctx, dp, scale, NID):
    B, L, D = feat.shape
    T = qk.shape[1]
    dev = feat.device
    gbuf = torch.empty((B, T, L), dtype=torch.float32, device=dev)
    # the three atomic accumulators are carved from ONE buffer: the C side zeroes adjacent buffers with a single fill launch
    acc = ACC(B * T * (1 + D + NID), dev)
    rowdot = acc[:B * T].view(B, T)
    dqk = acc[B * T:B * T * (1 + D)].view(B, T, D)
    dbtab = acc[B * T * (1 + D):].view(B, T, NID)
    dfeat = torch.empty((B, L, D), dtype=torch.float32, device=dev)
    hip.call('mg_attn_tok_bwd', hip.ptr(p), hip.ptr(feat), hip.ptr(qk), hip.ptr(ids), hip.ptr(dctx), hip.ptr(dp), c_int(B), c_int(T), c_int(L),
             c_int(D), c_int(NID), c_float(scale), hip.ptr(gbuf), hip.ptr(rowdot), hip.ptr(dqk), hip.ptr(dbtab), hip.ptr(dfeat), hip.stream())
    return dqk, dbtab, dfeat


def attn_feat_fwd(feat, kq, b2, vp, obias, pad, ids, scale, tn=False):
    """features <- tokens. feat (B,L,D), kq / vp (B,T,D), b2 (B,NID,T) [tn: (B,T,NID)], obias (D) or None, pad (B,T) uint8 or None -> out (B,L,D),
    p (B,L,T)."""
    hip.need_cuda(feat, kq, b2, vp, ids)
    B, L, D = feat.shape
    T, NID = kq.shape[1], (b2.shape[2] if tn else b2.shape[1])
    out = torch.empty((B, L, D), dtype=torch.float32, device=feat.device)
    p = torch.empty((B, L, T), dtype=torch.float32, device=feat.device)
    hip.call('mg_attn_feat_fwd_ex', hip.ptr(feat), hip.ptr(kq), hip.ptr(b2), hip.ptr(vp), hip.ptr(obias), hip.ptr(pad), hip.ptr(ids), c_int(B), c_int(T),
             c_int(L), c_int(D), c_int(NID), c_float(scale), hip.ptr(out), hip.ptr(p), c_int(int(tn)), hip.stream())
    return out, p


def attn_feat_bwd(dout, p, feat, kq, vp, ids, scale, NID, want_bias, tn=False):
    B, L, D = feat.shape
    T = kq.shape[1]
    dev = feat.device
    dfeat = torch.empty((B, L, D), dtype=torch.float32, device=dev)
    n1, n2 = B * T * D, B * NID * T
    acc = ACC(2 * n1 + n2 + (D if want_bias else 0), dev)                                           # one buffer, one fill launch (see attn_tok_bwd)
    dkq = acc[:n1].view(B, T, D)
    dvp = acc[n1:2 * n1].view(B, T, D)
    db2 = acc[2 * n1:2 * n1 + n2].view((B, T, NID) if tn else (B, NID, T))
    dob = acc[2 * n1 + n2:] if want_bias else None
    hip.call('mg_attn_feat_bwd_ex', hip.ptr(dout), hip.ptr(p), hip.ptr(feat), hip.ptr(kq), hip.ptr(vp), hip.ptr(ids), c_int(B), c_int(T), c_int(L),
             c_int(D), c_int(NID), c_float(scale), hip.ptr(dfeat), hip.ptr(dkq), hip.ptr(dvp), hip.ptr(db2), hip.ptr(dob), c_int(int(tn)), hip.stream())
    return dfeat, dkq, dvp, db2, dob


# ------------------------------------------------------------------------------------------------------------------
# temporal (video) elementwise kernels: ConvGRU gate math, eval-time alpha aggregation
# ------------------------------------------------------------------------------------------------------------------
def gru_gate_fwd(rz, x, h):
    M, C = x.numel() // x.shape[-1], x.shape[-1]
    xrh = torch.empty(x.shape[:-1] + (2 * C,), dtype=x.dtype, device=x.device)
    hip.call('mg_gru_gate_fwd', hip.ptr(rz), hip.ptr(x), hip.ptr(h), c_int(hip.dtype_code(x)), c_int(M), c_int(C), hip.ptr(xrh), hip.stream())
    return xrh


def gru_gate_bwd(dxrh, rz, h):
    M, C = h.numel() // h.shape[-1], h.shape[-1]
    dx, dh = torch.empty_like(h), torch.empty_like(h)
    drz = torch.zeros_like(rz)                                    # this call fills the r half, the z half comes from gru_out_bwd
    hip.call('mg_gru_gate_bwd', hip.ptr(dxrh), hip.ptr(rz), hip.ptr(h), c_int(hip.dtype_code(h)), c_int(M), c_int(C), hip.ptr(dx), hip.ptr(drz),
             hip.ptr(dh), hip.stream())
    return drz, dx, dh


def gru_out_fwd(rz, cpre, h):
    M, C = h.numel() // h.shape[-1], h.shape[-1]
    hn = torch.empty_like(h)
    hip.call('mg_gru_out_fwd', hip.ptr(rz), hip.ptr(cpre), hip.ptr(h), c_int(hip.dtype_code(h)), c_int(M), c_int(C), hip.ptr(hn), hip.stream())
    return hn


def gru_out_bwd(dhn, rz, cpre, h):
    M, C = h.numel() // h.shape[-1], h.shape[-1]
    dc, dh = torch.empty_like(h), torch.empty_like(h)
    drz = torch.zeros_like(rz)
    hip.call('mg_gru_out_bwd', hip.ptr(dhn), hip.ptr(rz), hip.ptr(cpre), hip.ptr(h), c_int(hip.dtype_code(h)), c_int(M), c_int(C), hip.ptr(drz),
             hip.ptr(dc), hip.ptr(dh), hip.stream())
    return drz, dc, dh


def temporal_fuse_(alphas3, prev, df3, db3):
    """In place on alphas3 (n_f >= 3, ...) fp32 contiguous: frames (t-1, t, ..., last = the reference's t+1); prev (...) or None;
    df3 / db3 like alphas3. Frames 1 and 2 are rewritten (maggie_temp.py:71,75)."""
    assert alphas3.is_contiguous() and df3.is_contiguous() and db3.is_contiguous() and alphas3.dtype == torch.float32
    n = alphas3[0].numel()
    hip.call('mg_temporal_fuse', hip.ptr(alphas3), hip.ptr(prev), hip.ptr(df3), hip.ptr(db3), c_long(n), c_int(alphas3.shape[0]), hip.stream())
    return alphas3


# ------------------------------------------------------------------------------------------------------------------
# row kernels of the sparse head with a device row count (maggie_amd/csrc/rows.hip)
# ------------------------------------------------------------------------------------------------------------------
def rows_sigmoid_mul(a, g, rows=None):
    """a * sigmoid(g); `a` may be a channel slice (view) of a wider row buffer."""
    M, C = g.shape
    out = torch.empty_like(g)
    hip.call('mg_rows_sigmoid_mul_fwd', hip.ptr(a), c_int(_ld(a)), hip.ptr(g), hip.ptr(out), c_int(hip.dtype_code(g)), c_int(M), c_int(C), hip.ptr(rows),
             hip.stream())
    return out


def rows_sigmoid_mul_bwd(dout, a, g, rows=None):
    M, C = g.shape
    da, dg = torch.empty_like(g), torch.empty_like(g)
    hip.call('mg_rows_sigmoid_mul_bwd', hip.ptr(dout), hip.ptr(a), c_int(_ld(a)), hip.ptr(g), hip.ptr(da), hip.ptr(dg), c_int(hip.dtype_code(g)),
             c_int(M), c_int(C), hip.ptr(rows), hip.stream())
    return da, dg


def rows_add(a, b, out=None, rows=None):
    """a + b over the live rows; a / b / out may be channel slices of wider buffers (out=a: in place)."""
    M, C = a.shape
    if out is None:
        out = torch.empty((M, C), dtype=a.dtype, device=a.device)
    hip.call('mg_rows_add', hip.ptr(a), c_int(_ld(a)), hip.ptr(b), c_int(_ld(b)), hip.ptr(out), c_int(_ld(out)), c_int(hip.dtype_code(a)), c_int(M),
             c_int(C), hip.ptr(rows), hip.stream())
    return out


def rows_dropout(x, p, state, salt, rows=None):
    """x * keep / (1 - p) with the counter-based mask of (state = device int64 [seed, step], salt); its own backward."""
    M, C = x.shape
    assert x.is_contiguous() and state.dtype == torch.int64 and state.numel() >= 2
    y = torch.empty_like(x)
    hip.call('mg_rows_dropout', hip.ptr(x), hip.ptr(y), c_int(hip.dtype_code(x)), c_int(M), c_int(C), c_float(float(p)), hip.ptr(state), c_int(int(salt)),
             hip.ptr(rows), hip.stream())
    return y


def rows_add_layernorm(x, r, gamma, beta, eps, rows=None):
    """LayerNorm(x + r) * gamma + beta per row -> y, rstat (M, 2) fp32 = (mean, rstd)."""
    M, C = x.shape
    assert x.is_contiguous() and r.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    y = torch.empty_like(x)
    rstat = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    hip.call('mg_rows_add_layernorm_fwd', hip.ptr(x), hip.ptr(r), hip.ptr(gamma), hip.ptr(beta), c_float(float(eps)), hip.ptr(y), hip.ptr(rstat),
             c_int(hip.dtype_code(x)), c_int(M), c_int(C), hip.ptr(rows), hip.stream())
    return y, rstat


def rows_add_layernorm_bwd(dy, x, r, gamma, rstat, rows=None):
    """-> dz (gradient of both x and r), dgamma, dbeta (fp32 [C])."""
    M, C = x.shape
    dz = torch.empty_like(x)
    dgb = ACC(2 * C, x.device)
    hip.call('mg_rows_add_layernorm_bwd', hip.ptr(dy), hip.ptr(x), hip.ptr(r), hip.ptr(gamma), hip.ptr(rstat), hip.ptr(dz), hip.ptr(dgb[:C]),
             hip.ptr(dgb[C:]), c_int(hip.dtype_code(x)), c_int(M), c_int(C), hip.ptr(rows), hip.stream())
    return dz, dgb[:C], dgb[C:]


def bits_patch_if_empty_(bits, count, H, W, y0, y1, x0, x1):
    """In place: if the device word `count` is 0, set bits [y0:y1, x0:x1] of every plane."""
    P = bits.shape[0]
    hip.call('mg_bits_patch_if_empty', hip.ptr(bits), hip.ptr(count), c_int(P), c_int(H), c_int(W), c_int(y0), c_int(y1), c_int(x0), c_int(x1),
             hip.stream())
    return bits
