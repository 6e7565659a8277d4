"""GPU parity of the implicit-GEMM convolution family (C ABI: mg_conv_fprop / mg_conv_wgrad) against a plain
PyTorch fp32 CPU reference of the same op. fp32 kernels use the exact f32 MFMA: tolerance 2e-4 relative to the
output scale; bf16 kernels: inputs are rounded to bf16 first, tolerance 2e-2 (bf16 output rounding)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _krsc(w):                         # (Cout, Cin, R, S) -> (Cout, R*S, Cin)
    co, ci, r, s = w.shape
    return w.permute(0, 2, 3, 1).reshape(co, r * s, ci).contiguous()


DTYPES = [torch.float32, torch.bfloat16, torch.float16]      # fp16: the reference's `--precision 16` storage type (same kernels, v_mfma_f32_16x16x32_f16)


def _tol(dtype):
    return {torch.float32: 2e-4, torch.bfloat16: 2.5e-2, torch.float16: 4e-3}[dtype]


CASES = [
    # N, Cin, Cout, H, W, k, stride, pad, dil
    (2, 8, 32, 20, 24, 3, 1, 1, 1),
    (1, 32, 64, 17, 19, 3, 2, 1, 1),
    (2, 64, 128, 12, 12, 3, 1, 2, 2),
    (1, 128, 256, 9, 9, 1, 1, 0, 1),
    (1, 256, 16, 8, 8, 3, 1, 1, 1),
    (3, 40, 72, 11, 13, 3, 1, 1, 1),
    (1, 32, 1, 16, 16, 3, 1, 1, 1),
    # 3x3 / stride 1 / pad 1 with Cin % 32 == 0, Cout >= 64 (bf16): the spatial halo-tile kernel -- ragged tiles in x and y, one and
    # several channel slabs, 8 x 16 and 4 x 16 pixel tiles; the other bf16 aligned shapes above / below run the direct-to-LDS im2col ring
    (2, 64, 64, 24, 40, 3, 1, 1, 1),
    (1, 128, 128, 9, 17, 3, 1, 1, 1),
    (4, 64, 128, 32, 32, 3, 1, 1, 1),
    (1, 32, 64, 16, 16, 3, 1, 1, 1),
    (1, 96, 192, 5, 33, 3, 1, 1, 1),
    (2, 96, 64, 16, 24, 3, 1, 1, 1),                 # odd number of 32-channel slabs through the two-slab ring of the 32-wide halo form
    (1, 160, 96, 8, 16, 3, 1, 1, 1),                 # five slabs, Cout not a multiple of 64
    (1, 64, 128, 20, 20, 3, 2, 1, 1),
    (2, 32, 32, 40, 24, 3, 1, 1, 1),
    (1, 8, 16, 33, 17, 3, 1, 1, 1),                  # the 8-channel-input kernels (16x16-pixel halo tiles, taps x channels as K), ragged tiles, Cout 16
    (1, 32, 16, 16, 24, 3, 1, 1, 1),                 # 16-wide halo tiles (the four waves split the rows); the (2, 8, 32, ...) case above runs them as its data gradient
    (2, 64, 8, 9, 33, 3, 1, 1, 1),                   # 8 output channels, two slabs, ragged tiles
    (2, 64, 128, 16, 24, 1, 2, 0, 1),                # stride-2 1x1 (downsample): its data gradient is the phase walk with three empty phases
    (2, 32, 64, 24, 40, 3, 2, 1, 1),                 # stride-2 3x3: phase-decomposed data gradient (1 / 2 / 2 / 4 taps), ragged phase tiles                 # halo-tile weight gradient at one 32 x 32 channel tile, ragged spatial tiles
    (2, 128, 64, 14, 14, 1, 1, 0, 1),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', CASES)
def test_conv_fprop_dgrad_wgrad(case, dtype):
    from maggie_amd import kernels as K
    dev = _dev()
    N, Cin, Cout, H, W, k, stride, pad, dil = case
    rs = np.random.RandomState(hash(case) % 1000)
    x = torch.from_numpy(rs.normal(size=(N, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rs.normal(size=(Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32))
    if dtype != torch.float32:
        x, w = x.to(dtype).float(), w.to(dtype).float()
    x.requires_grad_(True)
    w.requires_grad_(True)
    y_ref = F.conv2d(x, w, None, stride, pad, dil)
    Ho, Wo = y_ref.shape[-2:]
    gy = torch.from_numpy(rs.normal(size=tuple(y_ref.shape)).astype(np.float32))
    if dtype != torch.float32:
        gy = gy.to(dtype).float()
    y_ref.backward(gy)

    xd = _nhwc(x.detach()).to(dev, dtype)
    wd = _krsc(w.detach()).to(dev, dtype)
    y = K.conv_fprop(xd, wd, mode=K.MODE_CONV, N=N, Hin=H, Win=W, R=k, S=k, stride=stride, pad=pad, dil=dil)
    y = y.float().cpu().reshape(N, Ho, Wo, Cout).permute(0, 3, 1, 2)
    scale = y_ref.abs().max().item()
    assert (y - y_ref.detach()).abs().max().item() <= _tol(dtype) * scale

    # dgrad = TCONV-mode implicit GEMM with (Cin, taps, Cout) weights
    gyd = _nhwc(gy).to(dev, dtype)
    wt = w.detach().permute(1, 2, 3, 0).reshape(Cin, k * k, Cout).contiguous().to(dev, dtype)
    if Cout % 8 == 0:
        dx = K.conv_fprop(gyd, wt, mode=K.MODE_TCONV, N=N, Hin=Ho, Win=Wo, Hout=H, Wout=W, R=k, S=k, stride=stride,
                          pad=pad, dil=dil)
        dx = dx.float().cpu().reshape(N, H, W, Cin).permute(0, 3, 1, 2)
        assert (dx - x.grad).abs().max().item() <= _tol(dtype) * x.grad.abs().max().item()

    dw = K.conv_wgrad(xd, gyd, cout=Cout, mode=K.MODE_CONV, N=N, Hin=H, Win=W, Hout=Ho, Wout=Wo, R=k, S=k, stride=stride,
                      pad=pad, dil=dil)
    dw = dw.cpu().reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
    assert (dw - w.grad).abs().max().item() <= _tol(dtype) * w.grad.abs().max().item()
    if dtype != torch.float32:                        # converting reduce: dW written directly in the 16-bit type
        dwb = K.conv_wgrad(xd, gyd, cout=Cout, mode=K.MODE_CONV, N=N, Hin=H, Win=W, Hout=Ho, Wout=Wo, R=k, S=k, stride=stride,
                           pad=pad, dil=dil, out_dtype=dtype)
        assert dwb.dtype == dtype
        dwb = dwb.float().cpu().reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
        assert (dwb - dw).abs().max().item() <= 1e-2 * dw.abs().max().item()


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_epilogue_and_stats(dtype):
    from maggie_amd import kernels as K
    dev = _dev()
    N, Cin, Cout, H, W = 2, 32, 64, 16, 16
    rs = np.random.RandomState(3)
    q = (lambda t: t.to(dtype).float()) if dtype != torch.float32 else (lambda t: t)
    x = q(torch.from_numpy(rs.normal(size=(N, Cin, H, W)).astype(np.float32)))
    w = q(torch.from_numpy((rs.normal(size=(Cout, Cin, 3, 3)) / 17).astype(np.float32)))
    res = q(torch.from_numpy(rs.normal(size=(N, Cout, H // 2, W // 2)).astype(np.float32)))
    res2 = q(torch.from_numpy(rs.normal(size=(N, Cout, H, W)).astype(np.float32)))
    scale = torch.from_numpy(rs.uniform(0.5, 1.5, Cout).astype(np.float32))
    shift = torch.from_numpy(rs.normal(size=Cout).astype(np.float32))
    y_ref = F.conv2d(x, w, None, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None]
    y_ref = F.leaky_relu(y_ref + F.interpolate(res, scale_factor=2, mode='nearest'), 0.2) + res2
    stats = torch.zeros((K.conv_stat_rows(N * H * W, N, H, W), 2 * Cout), device=dev)
    big = torch.zeros((N * H * W, 2 * Cout), device=dev, dtype=dtype)
    K.conv_fprop(_nhwc(x).to(dev, dtype), _krsc(w).to(dev, dtype), N=N, Hin=H, Win=W, R=3, S=3, pad=1,
                 scale=scale.to(dev), shift=shift.to(dev), res=_nhwc(res).to(dev, dtype), res_mode=2,
                 res2=_nhwc(res2).to(dev, dtype), act=K.ACT_LRELU, slope=0.2, stats=stats, out=big, yoff=Cout)
    y = big[:, Cout:].float().cpu().reshape(N, H, W, Cout).permute(0, 3, 1, 2)
    assert big[:, :Cout].abs().max().item() == 0
    assert (y - y_ref).abs().max().item() <= _tol(dtype) * y_ref.abs().max().item()
    s = stats.sum(0).cpu()
    assert torch.allclose(s[:Cout], y.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    assert torch.allclose(s[Cout:], (y * y).sum((0, 2, 3)), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_transpose_k4s2(dtype):
    """ConvTranspose2d(k=4, s=2, p=1) forward == TCONV-mode implicit GEMM (decoder/resnet.py:20)."""
    from maggie_amd import kernels as K
    dev = _dev()
    N, Cin, Cout, H, W = 2, 64, 48, 7, 9
    rs = np.random.RandomState(4)
    q = (lambda t: t.to(dtype).float()) if dtype != torch.float32 else (lambda t: t)
    x = q(torch.from_numpy(rs.normal(size=(N, Cin, H, W)).astype(np.float32)))
    w = q(torch.from_numpy((rs.normal(size=(Cin, Cout, 4, 4)) / 20).astype(np.float32)))
    y_ref = F.conv_transpose2d(x, w, None, 2, 1)
    wk = w.permute(1, 2, 3, 0).reshape(Cout, 16, Cin).contiguous().to(dev, dtype)
    y = K.conv_fprop(_nhwc(x).to(dev, dtype), wk, mode=K.MODE_TCONV, N=N, Hin=H, Win=W, Hout=2 * H, Wout=2 * W, R=4, S=4,
                     stride=2, pad=1)
    y = y.float().cpu().reshape(N, 2 * H, 2 * W, Cout).permute(0, 3, 1, 2)
    assert (y - y_ref).abs().max().item() <= _tol(dtype) * y_ref.abs().max().item()


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_transpose_phased_epilogue(dtype):
    """The phase-major row order of the stride-2 transposed walk must be invisible to the epilogue: scale / shift, post residual, channel
    slice of a wider buffer and the BatchNorm statistics, on a geometry whose phases are not whole row tiles (Mp = 480)."""
    from maggie_amd import kernels as K
    dev = _dev()
    N, Cin, Cout, H, W = 2, 64, 64, 12, 20
    rs = np.random.RandomState(14)
    q = (lambda t: t.to(dtype).float()) if dtype != torch.float32 else (lambda t: t)
    x = q(torch.from_numpy(rs.normal(size=(N, Cin, H, W)).astype(np.float32)))
    w = q(torch.from_numpy((rs.normal(size=(Cin, Cout, 4, 4)) / 20).astype(np.float32)))
    res2 = q(torch.from_numpy(rs.normal(size=(N, Cout, 2 * H, 2 * W)).astype(np.float32)))
    scale = torch.from_numpy(rs.uniform(0.5, 1.5, Cout).astype(np.float32))
    shift = torch.from_numpy(rs.normal(size=Cout).astype(np.float32))
    y_ref = F.relu(F.conv_transpose2d(x, w, None, 2, 1) * scale[None, :, None, None] + shift[None, :, None, None]) + res2
    wk = w.permute(1, 2, 3, 0).reshape(Cout, 16, Cin).contiguous().to(dev, dtype)
    stats = torch.zeros((K.conv_stat_rows(N * 4 * H * W, N, 2 * H, 2 * W), 2 * Cout), device=dev)
    big = torch.zeros((N * 4 * H * W, 2 * Cout), device=dev, dtype=dtype)
    K.conv_fprop(_nhwc(x).to(dev, dtype), wk, mode=K.MODE_TCONV, N=N, Hin=H, Win=W, Hout=2 * H, Wout=2 * W, R=4, S=4, stride=2, pad=1,
                 scale=scale.to(dev), shift=shift.to(dev), res2=_nhwc(res2).to(dev, dtype), act=K.ACT_RELU, stats=stats, out=big, yoff=Cout)
    y = big[:, Cout:].float().cpu().reshape(N, 2 * H, 2 * W, Cout).permute(0, 3, 1, 2)
    assert big[:, :Cout].abs().max().item() == 0
    assert (y - y_ref).abs().max().item() <= _tol(dtype) * y_ref.abs().max().item()
    s = stats.sum(0).cpu()
    assert torch.allclose(s[:Cout], y.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    assert torch.allclose(s[Cout:], (y * y).sum((0, 2, 3)), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize('dtype', DTYPES)
def test_gather_conv(dtype):
    """MG_MODE_GATHER against the oracle's gather restatement (submanifold 3x3 over a random active set)."""
    from maggie_amd import kernels as K
    from oracle import region, refmodel
    dev = _dev()
    rs = np.random.RandomState(5)
    act = rs.uniform(size=(3, 24, 20)) > 0.6
    nbr = region.subm_neighbors(act)
    R_, Cin, Cout = nbr.shape[0], 64, 32
    q = (lambda t: t.to(dtype).float()) if dtype != torch.float32 else (lambda t: t)
    feat = q(torch.from_numpy(rs.normal(size=(R_, Cin)).astype(np.float32)))
    w = q(torch.from_numpy((rs.normal(size=(Cout, 3, 3, Cin)) / 24).astype(np.float32)))
    bias = torch.from_numpy(rs.normal(size=Cout).astype(np.float32))
    y_ref = refmodel._gather_conv(feat, nbr, w, bias)
    y = K.conv_fprop(feat.to(dev, dtype), w.reshape(Cout, 9, Cin).to(dev, dtype), mode=K.MODE_GATHER,
                     nbr=torch.from_numpy(nbr).to(dev), R=3, S=3, shift=bias.to(dev))
    assert (y.float().cpu() - y_ref).abs().max().item() <= _tol(dtype) * y_ref.abs().max().item()
    gy = q(torch.from_numpy(rs.normal(size=(R_, Cout)).astype(np.float32)))
    w_ = w.clone().requires_grad_(True)
    refmodel._gather_conv(feat, nbr, w_, None).backward(gy)
    dw = K.conv_wgrad(feat.to(dev, dtype), gy.to(dev, dtype), cout=Cout, mode=K.MODE_GATHER,
                      nbr=torch.from_numpy(nbr).to(dev), R=3, S=3)
    assert (dw.cpu().reshape(Cout, 3, 3, Cin) - w_.grad).abs().max().item() <= _tol(dtype) * w_.grad.abs().max().item()


@pytest.mark.parametrize('dtype', DTYPES)
def test_parked_weight_gradient_reductions_give_the_same_bits(dtype):
    """mg_conv_wgrad_park + mg_wgrad_reduce_batched (one launch for all the layers of a backward pass) against mg_conv_wgrad_ws (one reduce
    launch per layer): every geometry of CASES (all the kernel forms: halo, per-tap, 8-channel input, single split) plus a gather layer, fp32 and
    16-bit dW, 70+ descriptors (more than one table of 64) -- torch.equal on every dW."""
    from maggie_amd import kernels as K
    from oracle import region
    dev = _dev()
    rs = np.random.RandomState(11)
    park, pairs, n_parked = [], [], 0
    for rep in range(2):
        for (N, Cin, Cout, H, W, k, stride, pad, dil) in CASES:
            if Cout % 8:
                continue
            Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
            Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
            x = torch.from_numpy(rs.normal(size=(N * H * W, Cin)).astype(np.float32)).to(dev, dtype)
            gy = torch.from_numpy(rs.normal(size=(N * Ho * Wo, Cout)).astype(np.float32)).to(dev, dtype)
            for od in ((torch.float32,) if dtype == torch.float32 else (torch.float32, dtype)):
                kw = dict(cout=Cout, mode=K.MODE_CONV, N=N, Hin=H, Win=W, Hout=Ho, Wout=Wo, R=k, S=k, stride=stride, pad=pad, dil=dil, out_dtype=od)
                ref = K.conv_wgrad(x, gy, **kw)
                before = len(park)
                got = K.conv_wgrad(x, gy, park=park, **kw)
                n_parked += len(park) - before
                pairs.append((ref, got))
    act = rs.uniform(size=(2, 24, 20)) > 0.6
    nbr = torch.from_numpy(region.subm_neighbors(act)).to(dev)
    feat = torch.from_numpy(rs.normal(size=(nbr.shape[0], 64)).astype(np.float32)).to(dev, dtype)
    gy = torch.from_numpy(rs.normal(size=(nbr.shape[0], 32)).astype(np.float32)).to(dev, dtype)
    rows = torch.tensor([nbr.shape[0] - 37], dtype=torch.int32, device=dev)
    for r in (None, rows):
        ref = K.conv_wgrad(feat, gy, cout=32, mode=K.MODE_GATHER, nbr=nbr, R=3, S=3, out_dtype=dtype, rows=r)
        got = K.conv_wgrad(feat, gy, cout=32, mode=K.MODE_GATHER, nbr=nbr, R=3, S=3, out_dtype=dtype, rows=r, park=park)
        pairs.append((ref, got))
    assert len(park) >= n_parked and n_parked > (64 if dtype != torch.float32 else 10), (len(park), n_parked)
    K.wgrad_reduce_batched(park)
    assert park == []
    torch.cuda.synchronize()
    for i, (ref, got) in enumerate(pairs):
        assert ref.dtype == got.dtype and torch.equal(ref, got), i


SPLITK_CASES = [
    # N, Cin, Cout, H, W, k, stride, pad, dil   -- deep K, few output rows: the split-K plan of mg_conv_fprop_ws
    (2, 256, 256, 16, 16, 3, 1, 1, 1),          # K 2304 (the shallowest K that splits), M 512
    (1, 512, 128, 16, 16, 3, 1, 2, 2),          # dilated (ASPP-like), Cout 128
    (1, 392, 72, 12, 20, 3, 1, 1, 1),           # unaligned Cin (generic tap walk), ragged Cout / M
    (4, 512, 512, 8, 8, 3, 1, 1, 1),            # M 256, 8 splits
    (1, 4096, 64, 8, 8, 1, 1, 0, 1),            # 1x1, K 4096, Cout 64 tile
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', SPLITK_CASES)
def test_conv_fprop_split_k(case, dtype):
    """Layers the library runs with the K dimension split over several blocks per tile (partial slabs + finishing kernel):
    forward with the full epilogue (scale/shift, half-resolution residual, LeakyReLU, post residual, BN statistics in both
    accumulator layouts) and the data gradient (TCONV mode), against torch on the CPU; and bit-identical to the unsplit kernel."""
    import ctypes, os
    from maggie_amd import kernels as K, hip
    dev = _dev()
    N, Cin, Cout, H, W, k, stride, pad, dil = case
    rs = np.random.RandomState(sum(case))
    q = (lambda t: t.to(dtype).float()) if dtype != torch.float32 else (lambda t: t)
    x = q(torch.from_numpy(rs.normal(size=(N, Cin, H, W)).astype(np.float32))).requires_grad_(True)
    w = q(torch.from_numpy((rs.normal(size=(Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)))
    scale = torch.from_numpy(rs.uniform(0.5, 1.5, Cout).astype(np.float32))
    shift = torch.from_numpy(rs.normal(size=Cout).astype(np.float32))
    y0 = F.conv2d(x, w, None, stride, pad, dil)
    Ho, Wo = y0.shape[-2:]
    even = Ho % 2 == 0 and Wo % 2 == 0
    res = q(torch.from_numpy(rs.normal(size=(N, Cout, Ho // 2, Wo // 2) if even else (N, Cout, Ho, Wo)).astype(np.float32)))
    res2 = q(torch.from_numpy(rs.normal(size=(N, Cout, Ho, Wo)).astype(np.float32)))
    y_ref = y0 * scale[None, :, None, None] + shift[None, :, None, None]
    y_ref = F.leaky_relu(y_ref + (F.interpolate(res, scale_factor=2, mode='nearest') if even else res), 0.2) + res2
    gy = q(torch.from_numpy(rs.normal(size=tuple(y0.shape)).astype(np.float32)))
    y0.backward(gy)

    xd, wd = _nhwc(x.detach()).to(dev, dtype), _krsc(w).to(dev, dtype)
    geo = dict(N=N, Hin=H, Win=W, R=k, S=k, stride=stride, pad=pad, dil=dil)
    # the plan really splits these layers -- except the bf16 3x3 / stride 1 / pad 1 ones that the spatial halo-tile kernel takes (it beats
    # split-K there); the full-epilogue checks below then exercise the halo kernel's epilogue
    p = K._conv_params(xd.view(-1, Cin), wd, torch.empty((N * Ho * Wo, Cout), dtype=dtype, device=dev), K.MODE_CONV, N, H, W, Ho, Wo, k, k, stride,
                       pad, dil, N * Ho * Wo, Cin, Cout)
    halo = (dtype != torch.float32 and k == 3 and stride == 1 and pad == 1 and dil == 1 and Cin % 32 == 0 and Cin >= 96 and Cout >= 64
            and W >= 16 and H >= 4)
    need = K._fprop_workspace_fn()(ctypes.byref(p))
    assert need == 0 if halo else need >= 2 * N * Ho * Wo * Cout
    # statistics rows: 32 replicas (the atomic form), one row per output tile (the deterministic form), one sums-only row (feeds the exact
    # two-pass variance; accumulated with atomics, so only outside deterministic mode)
    for stat_rows in (K.STAT_REPLICAS, K.conv_stat_rows(N * Ho * Wo, N, Ho, Wo), 1):
        hip.set_deterministic(stat_rows > K.STAT_REPLICAS)      # 32 replicas / one sums-only row: the atomic forms
        try:
            stats = torch.zeros((stat_rows, 2 * Cout), device=dev) if stat_rows > 1 else torch.zeros(2 * Cout, device=dev)
            y = K.conv_fprop(xd.view(-1, Cin), wd, scale=scale.to(dev), shift=shift.to(dev), res=_nhwc(res).to(dev, dtype).view(-1, Cout),
                             res_mode=2 if even else 1, res2=_nhwc(res2).to(dev, dtype).view(-1, Cout), act=K.ACT_LRELU, slope=0.2, stats=stats, **geo)
        finally:
            hip.set_deterministic(True)
        yf = y.float().cpu().reshape(N, Ho, Wo, Cout).permute(0, 3, 1, 2)
        assert (yf - y_ref.detach()).abs().max().item() <= _tol(dtype) * y_ref.abs().max().item()
        s = stats.sum(0).cpu() if stat_rows > 1 else stats.cpu()
        assert torch.allclose(s[:Cout], yf.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
        if stat_rows > 1:
            assert torch.allclose(s[Cout:], (yf * yf).sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
        else:
            assert float(s[Cout:].abs().max()) == 0                       # stat_mode 1: column sums only
    # same result as the unsplit kernel up to fp32 summation order (bf16: at most one rounding step on a few elements)
    y_split = K.conv_fprop(xd.view(-1, Cin), wd, **geo)
    p.y = hip.ptr(torch.empty_like(y_split))
    y_plain = torch.empty_like(y_split)
    p.y = hip.ptr(y_plain)
    hip.call('mg_conv_fprop', ctypes.byref(p), hip.stream())
    d = (y_split.float() - y_plain.float()).abs().max().item()
    assert d <= (2e-5 if dtype == torch.float32 else 1.6e-2) * y_plain.float().abs().max().item()
    # dgrad: TCONV mode over the same geometry (K = taps * Cout)
    if Cout % 8 == 0:
        wt = w.permute(1, 2, 3, 0).reshape(Cin, k * k, Cout).contiguous().to(dev, dtype)
        dx = K.conv_fprop(_nhwc(gy).to(dev, dtype).view(-1, Cout), wt, mode=K.MODE_TCONV, N=N, Hin=Ho, Win=Wo, Hout=H, Wout=W, R=k, S=k,
                          stride=stride, pad=pad, dil=dil)
        dx = dx.float().cpu().reshape(N, H, W, Cin).permute(0, 3, 1, 2)
        assert (dx - x.grad).abs().max().item() <= _tol(dtype) * x.grad.abs().max().item()


LINK_CASES = [
    # N, C0, C1, C2, H, W, (k2, stride2, pad2, transposed2), act of the BatchNorm layer, residual into the BatchNorm, carry through conv 2
    (2, 32, 64, 64, 40, 48, (3, 1, 1, False), 1, False, False),      # 3x3 s1 consumer: halo-tile data gradient (bf16), im2col (fp32)
    (2, 32, 64, 32, 40, 48, (3, 1, 1, False), 2, True, True),        # LeakyReLU + residual + skip gradient added in the same epilogue (carry)
    (2, 16, 32, 64, 48, 40, (3, 2, 1, False), 1, False, False),      # stride-2 consumer: phase-decomposed data gradient, ragged phase tiles
    (4, 64, 128, 64, 24, 24, (1, 1, 0, False), 0, False, False),     # BatchNorm without activation, 1x1 consumer
    (2, 64, 64, 32, 20, 28, (4, 2, 1, True), 2, False, True),        # transposed consumer (decoder): its data gradient is a stride-2 conv
    (1, 256, 256, 256, 34, 34, (3, 1, 1, False), 1, False, False),   # deep layer with few rows
    (1, 512, 512, 256, 40, 40, (1, 1, 0, False), 1, False, False),   # K = 512 1x1: im2col ring
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', LINK_CASES)
def test_bn_backward_sums_in_the_consumer_dgrad_epilogue(case, dtype):
    """Round 3 (functional.BnLink, mg_conv_params.bnb_*): conv -> BN(+res, act) -> conv. The second conv's data-gradient epilogue writes
    g = dz * act'(z) and accumulates the BatchNorm layer's backward sums; only the apply pass is left. Against torch on the CPU, and against
    the unlinked path (separate reduce pass) of this build."""
    from maggie_amd import functional as MF
    dev = _dev()
    N, C0, C1, C2, H, W, (k2, s2, p2, tr2), act, with_res, carry = case
    rs = np.random.RandomState(sum(case[:6]))
    q = (lambda t: t.to(dtype).float()) if dtype != torch.float32 else (lambda t: t)
    x = q(torch.from_numpy(rs.normal(size=(N, C0, H, W)).astype(np.float32))).requires_grad_(True)
    w1 = q(torch.from_numpy((rs.normal(size=(C1, C0, 3, 3)) / np.sqrt(C0 * 9)).astype(np.float32))).requires_grad_(True)
    w2s = (C1, C2, k2, k2) if tr2 else (C2, C1, k2, k2)
    w2 = q(torch.from_numpy((rs.normal(size=w2s) / np.sqrt(C1 * k2 * k2)).astype(np.float32))).requires_grad_(True)
    res = q(torch.from_numpy(rs.normal(size=(N, C1, H, W)).astype(np.float32))).requires_grad_(True) if with_res else None
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C1).astype(np.float32)).requires_grad_(True)
    beta = torch.from_numpy(rs.normal(size=C1).astype(np.float32)).requires_grad_(True)
    actf = [lambda t: t, F.relu, lambda t: F.leaky_relu(t, 0.2)][act]

    y1 = F.conv2d(x, w1, None, 1, 1)
    z = F.batch_norm(y1, None, None, gamma, beta, True, 0.1, 1e-5)
    z = actf(z if res is None else z + res)
    y2 = F.conv_transpose2d(z, w2, None, s2, p2) if tr2 else F.conv2d(z, w2, None, s2, p2)
    gy = q(torch.from_numpy(rs.normal(size=tuple(y2.shape)).astype(np.float32)))
    gz = q(torch.from_numpy(rs.normal(size=tuple(z.shape)).astype(np.float32)))          # the skip branch's gradient (carry)
    (y2 * gy).sum().backward() if not carry else ((y2 * gy).sum() + (z * gz).sum()).backward()

    prev_link = MF.BN_LINK

    def run(link):
        MF.BN_LINK = link
        try:
            MF.ARENA.reset(dev)
            bn = torch.nn.BatchNorm2d(C1).to(dev)
            with torch.no_grad():
                bn.weight.copy_(gamma.detach()); bn.bias.copy_(beta.detach())
            xd = _nhwc(x.detach()).to(dev, dtype).requires_grad_(True)
            w1d = _krsc(w1.detach()).to(dev, dtype).requires_grad_(True)
            w2k = w2.detach().permute(1, 2, 3, 0) if tr2 else w2.detach().permute(0, 2, 3, 1)
            w2d = w2k.reshape(C2, k2 * k2, C1).contiguous().to(dev, dtype).requires_grad_(True)
            rd = None if res is None else _nhwc(res.detach()).to(dev, dtype).requires_grad_(True)
            zd = MF.conv_bn_act(xd, w1d, bn, act, 3, 3, 1, 1, 1, res=rd, link_out=True)
            linked = getattr(zd, '_mg_bnlink', None) is not None
            if carry:
                y2d, zc = MF.conv2d(zd, w2d, None, k2, k2, s2, p2, 1, tr2, carry=True)
                loss = (y2d.float() * _nhwc(gy).to(dev)).sum() + (zc.float() * _nhwc(gz).to(dev)).sum()
            else:
                y2d = MF.conv2d(zd, w2d, None, k2, k2, s2, p2, 1, tr2)
                loss = (y2d.float() * _nhwc(gy).to(dev)).sum()
            loss.backward()
            out = {'dx': xd.grad.float().cpu().permute(0, 3, 1, 2), 'dgamma': bn.weight.grad.cpu(), 'dbeta': bn.bias.grad.cpu(),
                   'dw1': w1d.grad.float().cpu().reshape(C1, 3, 3, C0).permute(0, 3, 1, 2), 'y2': y2d.detach().float().cpu().permute(0, 3, 1, 2)}
            if rd is not None:
                out['dres'] = rd.grad.float().cpu().permute(0, 3, 1, 2)
            return out, linked
        finally:
            MF.BN_LINK = prev_link

    got, linked = run(True)
    base, base_linked = run(False)
    assert linked and not base_linked, 'the link must be taken in the first run and not in the second'
    ref = {'dx': x.grad, 'dgamma': gamma.grad, 'dbeta': beta.grad, 'dw1': w1.grad, 'y2': y2.detach()}
    if res is not None:
        ref['dres'] = res.grad
    for k_, r_ in ref.items():
        sc_ = r_.abs().max().item()
        # against torch in the L2 sense: a pre-activation value within rounding distance of zero (fp32: summation order; bf16: the conv output
        # entering the BatchNorm is rounded here and not on the CPU) lands on the other side of the ReLU / LeakyReLU kink and moves its
        # gradient by O(1). The element-wise comparison is made against the unlinked path below, which shares the activations.
        rel = ((got[k_] - r_).norm() / r_.norm().clamp_min(1e-12)).item()
        assert rel <= (5e-3 if dtype == torch.float32 else 3e-2), (k_, rel)
        # linked vs unlinked path of this build: same arithmetic up to the order of the fp32 reductions (and, in bf16, one rounding of g
        # where the unlinked path rounds dz and re-derives g in fp32)
        assert (got[k_] - base[k_]).abs().max().item() <= (2e-5 if dtype == torch.float32 else 2e-2) * sc_, ('vs unlinked', k_)


XF_CASES = [
    # N, Cin, Cout, H, W: 3x3 / stride 1 / pad 1 consumers of a BatchNorm layer's raw input
    (2, 32, 32, 24, 40),            # one slab, 32-wide tiles (the 512 x 512 shortcut / stem layers)
    (1, 64, 64, 16, 32),            # two slabs, single-stage form
    (2, 32, 64, 20, 24),            # ragged spatial tiles
    (1, 64, 32, 9, 17),             # ragged in both directions, H not a multiple of 8
    (2, 128, 64, 16, 16),           # two-slab ring (Cin >= 96), four slabs
    (1, 96, 64, 16, 24),            # odd number of slabs through the ring
    (1, 256, 128, 8, 16),           # eight slabs, two channel tiles
    (1, 512, 64, 8, 16),            # the widest table (Cin 512)
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('act', [0, 1, 2])
@pytest.mark.parametrize('case', XF_CASES)
def test_operand_transform_equals_the_stored_batchnorm_output(case, act, dtype):
    """mg_conv_params.xf_* (round 5: conv + BatchNorm + activation as one pass in training, maggie/network/encoder/resnet.py:26-37): the forward
    and the weight-gradient kernel fed the RAW producer output y plus (scale, shift, act) must give EXACTLY what they give when fed
    z = mg_affine_act(y) -- the transform forms the same bits on the operand's way into LDS, and padding stays zero (a transformed zero would be
    act(shift) != 0). Against torch (fp32, conv2d of the stored z) as well."""
    from maggie_amd import kernels as K
    dev = _dev()
    N, Cin, Cout, H, W = case
    rs = np.random.RandomState(Cin + Cout + H)
    y = torch.from_numpy(rs.normal(0.3, 1.5, (N * H * W, Cin)).astype(np.float32)).to(dev, dtype)
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, Cin).astype(np.float32) * rs.choice([-1.0, 1.0], Cin).astype(np.float32)).to(dev)
    sh = torch.from_numpy(rs.normal(0.0, 0.7, Cin).astype(np.float32)).to(dev)
    w = torch.from_numpy((rs.normal(size=(Cout, 9, Cin)) / np.sqrt(9 * Cin)).astype(np.float32)).to(dev, dtype)
    dy = torch.from_numpy(rs.normal(size=(N * H * W, Cout)).astype(np.float32)).to(dev, dtype)
    assert K.conv_xform_ok(dtype, N, H, W, Cin, Cout, 3, 3, 1, 1, 1, 0) and K.conv_xform_ok(dtype, N, H, W, Cin, Cout, 3, 3, 1, 1, 1, 1)
    z = K.affine_act(y, sc, sh, act=act, slope=0.2)
    geo = dict(mode=K.MODE_CONV, N=N, Hin=H, Win=W, R=3, S=3, stride=1, pad=1, dil=1)
    xf = (sc, sh, act, 0.2)
    o_ref = K.conv_fprop(z, w, **geo)
    o_xf = K.conv_fprop(y, w, xf=xf, **geo)
    assert torch.equal(o_xf, o_ref), 'forward: %d of %d values differ, max %.3g' % (
        int((o_xf != o_ref).sum()), o_ref.numel(), float((o_xf.float() - o_ref.float()).abs().max()))
    # statistics epilogue rides along unchanged
    nrow = K.conv_stat_rows(N * H * W, N, H, W)
    st_a, st_b = torch.zeros((nrow, 2 * Cout), device=dev), torch.zeros((nrow, 2 * Cout), device=dev)
    K.conv_fprop(z, w, stats=st_a, **geo)
    K.conv_fprop(y, w, stats=st_b, xf=xf, **geo)
    assert torch.equal(st_a, st_b)
    for out_dtype in (torch.float32, dtype):
        g_ref = K.conv_wgrad(z, dy, cout=Cout, Hout=H, Wout=W, out_dtype=out_dtype, **geo)
        g_xf = K.conv_wgrad(y, dy, cout=Cout, Hout=H, Wout=W, out_dtype=out_dtype, xf=xf, **geo)
        assert torch.equal(g_xf, g_ref), 'weight gradient (%s): max %.3g' % (out_dtype, float((g_xf.float() - g_ref.float()).abs().max()))
    # ... and the stored form itself against torch
    zt = z.float().cpu().view(N, H, W, Cin).permute(0, 3, 1, 2)
    wt = w.float().cpu().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = F.conv2d(zt, wt, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    assert (o_xf.float().cpu() - ref).abs().max() <= _tol(dtype) * ref.abs().max()


def test_operand_transform_is_refused_where_no_kernel_form_has_it():
    """A geometry whose kernel form cannot transform its operand reports so (mg_conv_xform_ok) and the launch itself fails with -9 instead of
    silently convolving un-normalised values."""
    from maggie_amd import kernels as K, hip
    dev = _dev()
    assert not K.conv_xform_ok(torch.bfloat16, 1, 16, 16, 64, 64, 3, 3, 2, 1, 1, 0)          # stride 2: im2col form
    assert not K.conv_xform_ok(torch.bfloat16, 1, 16, 16, 64, 64, 1, 1, 1, 0, 1, 0)          # 1x1: direct-to-LDS im2col ring
    assert not K.conv_xform_ok(torch.float32, 1, 16, 16, 64, 64, 3, 3, 1, 1, 1, 0)           # fp32 storage
    y = torch.randn(256, 64, device=dev).bfloat16()
    w = torch.randn(64, 1, 64, device=dev).bfloat16()
    one = torch.ones(64, device=dev)
    with pytest.raises(hip.MaggieHipError, match='-9'):
        K.conv_fprop(y, w, mode=K.MODE_CONV, N=1, Hin=16, Win=16, R=1, S=1, stride=1, pad=0, dil=1, xf=(one, one, 1, 0.2))


def test_parked_reductions_must_meet_their_join_inside_the_same_backward():
    """a weight that comes out of the batched weight pipeline parks its slab reduction until the pipeline's backward (the
    join) flushes it. (1) A backward pass that ends WITHOUT the join -- the weight pipeline differentiated separately, a graph split that puts it
    elsewhere -- must raise instead of handing out unreduced slabs; (2) a weight consumed by a SECOND convolution (here: a transposed use, which never
    parks itself) is not parked at all: autograd adds the two gradients, both complete."""
    from maggie_amd import functional as MF, hip
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 16, 32, 64), generator=g).to(dev, torch.bfloat16).requires_grad_(True)

    def joined_weight():
        w = (torch.randn((64, 9, 64), generator=g) / 24).to(dev, torch.bfloat16).requires_grad_(True)
        w._mg_join = True                      # what SpectralNormBatch / WeightBank put on the tensors they hand out
        return w

    if not MF.PARK_WGRAD:
        pytest.skip('MAGGIE_PARK_WGRAD=0')
    w = joined_weight()
    y = MF.conv2d(x, w)
    with pytest.raises(hip.MaggieHipError, match='never flushed'):
        y.float().sum().backward()
    assert MF.PARKED == []
    # two uses of DIFFERENT geometry: counted, so neither parks (and they do not form a group); gradients are complete and equal to the sum of the
    # two uses' own gradients
    w3 = joined_weight()
    y1, y2 = MF.conv2d(x, w3), MF.conv2d(x[:, :8].contiguous(), w3)
    (y1.float().sum() + 2.0 * y2.float().sum()).backward()
    assert MF.PARKED == [] and MF._OPEN_GROUPS == []
    w_ref = w3.detach().clone().requires_grad_(True)
    (MF.conv2d(x.detach(), w_ref).float().sum() + 2.0 * MF.conv2d(x.detach()[:, :8].contiguous(), w_ref).float().sum()).backward()
    torch.cuda.synchronize()
    assert (w3.grad.float() - w_ref.grad.float()).abs().max() <= 2e-2 * w_ref.grad.float().abs().max()


def test_one_weight_used_by_several_identical_convolutions_reduces_all_slabs_once():
    """Round 5 (video: the ConvGRU gate weights serve one convolution per frame and direction, maggie/network/module/conv_gru.py:17-27). k calls of ONE
    geometry on a joined weight: the weight-gradient GEMMs write their slabs side by side, only the call whose backward runs last returns a gradient,
    and ONE parked reduction adds all k * splits slabs in fp32 -- closer to the fp32 sum than k separately rounded gradients added in bf16, and one
    launch at the join instead of k reductions + k - 1 adds. A backward pass that misses one of the uses raises."""
    from maggie_amd import functional as MF, kernels as K, hip
    dev = _dev()
    if not (MF.PARK_WGRAD and MF.GROUP_WGRAD):
        pytest.skip('grouped weight gradients switched off')
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn((1, 64, 64, 128), generator=g).to(dev, torch.bfloat16) for _ in range(5)]
    w0 = (torch.randn((128, 9, 128), generator=g) / 34).to(dev, torch.bfloat16)
    dys = [torch.randn((1, 64, 64, 128), generator=g).to(dev, torch.bfloat16) for _ in range(5)]

    class Join(torch.autograd.Function):                          # stands in for SpectralNormBatch.backward: the point where the parked work is flushed
        @staticmethod
        def forward(ctx, w):
            return w.view_as(w)

        @staticmethod
        def backward(ctx, gw):
            MF.flush_parked()
            return gw

    def grad(grouped, skip_last=False):
        MF.GROUP_WGRAD = grouped
        try:
            leaf = w0.clone().requires_grad_(True)
            w = Join.apply(leaf)
            w._mg_join = True
            ys = [MF.conv2d(x_, w) for x_ in xs]
            use = list(zip(ys, dys))[:-1] if skip_last else list(zip(ys, dys))
            torch.autograd.backward([y for y, _ in use], [d for _, d in use])
            torch.cuda.synchronize()
            return leaf.grad
        finally:
            MF.GROUP_WGRAD = True

    n_before = []
    orig = K.wgrad_reduce_batched

    def counting(park):
        n_before.append(len(park))
        return orig(park)
    K.wgrad_reduce_batched = counting
    try:
        got = grad(True)
    finally:
        K.wgrad_reduce_batched = orig
    assert n_before == [1]                                         # ONE parked reduction for the five calls
    assert MF.PARKED == [] and MF._OPEN_GROUPS == []
    sep = grad(False)
    ref = sum(K.conv_wgrad(x_.view(-1, 128), d_.view(-1, 128), cout=128, mode=K.MODE_CONV, N=1, Hin=64, Win=64, Hout=64, Wout=64, R=3, S=3, stride=1,
                           pad=1, dil=1, out_dtype=torch.float32) for x_, d_ in zip(xs, dys))
    scale = float(ref.abs().max())
    e_grp, e_sep = float((got.float() - ref).abs().max()) / scale, float((sep.float() - ref).abs().max()) / scale
    assert e_grp <= 4e-3 and e_grp <= e_sep, (e_grp, e_sep)        # one rounding to bf16 (2^-9 relative) against five + four adds
    with pytest.raises(hip.MaggieHipError, match='did not see the backward of all their uses'):
        grad(True, skip_last=True)
    assert MF.PARKED == [] and MF._OPEN_GROUPS == []
    assert torch.equal(grad(True), got)                            # and the state is clean for the next pass


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('act', [0, 2])
@pytest.mark.parametrize('Cin,Cout,kind', [(32, 32, 'gather'), (64, 32, 'gather'), (32, 8, 'gather'), (32, 32, 'lin')])
def test_operand_transform_on_the_sparse_row_matrices(Cin, Cout, kind, act, dtype):
    """The sparse head's BatchNorm1d layers on the operand path (round 5): gather 3x3 and 1x1 convolutions over row matrices with a DEVICE row count
    (persistent register-staged forward kernels, all-taps / per-tap weight-gradient kernels) fed the raw rows + (scale, shift, act) give exactly
    what they give on the stored BatchNorm output -- missing neighbours and rows beyond the live count stay zero."""
    from maggie_amd import kernels as K
    from oracle import region
    dev = _dev()
    rs = np.random.RandomState(Cin + Cout + act)
    active = rs.uniform(size=(3, 40, 56)) > 0.55
    nbr = torch.from_numpy(region.subm_neighbors(active)).to(dev)
    cap = nbr.shape[0]
    live = cap - 101
    rows = torch.tensor([live], dtype=torch.int32, device=dev)
    y = torch.from_numpy(rs.normal(0.2, 1.4, (cap, Cin)).astype(np.float32)).to(dev, dtype)
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, Cin).astype(np.float32) * rs.choice([-1.0, 1.0], Cin).astype(np.float32)).to(dev)
    sh = torch.from_numpy(rs.normal(0.0, 0.7, Cin).astype(np.float32)).to(dev)
    taps = 9 if kind == 'gather' else 1
    w = torch.from_numpy((rs.normal(size=(Cout, taps, Cin)) / np.sqrt(taps * Cin)).astype(np.float32)).to(dev, dtype)
    dy = torch.from_numpy(rs.normal(size=(cap, Cout)).astype(np.float32)).to(dev, dtype)
    z = K.affine_act(y, sc, sh, act=act, slope=0.2)         # (all capacity rows: this test's table lets live rows gather rows beyond the live count)
    xf = (sc, sh, act, 0.2)
    if kind == 'gather':
        fkw = dict(mode=K.MODE_GATHER, nbr=nbr, R=3, S=3, M=cap, rows=rows)
        wkw = dict(cout=Cout, mode=K.MODE_GATHER, nbr=nbr, R=3, S=3, M=cap, rows=rows)
    else:
        fkw = dict(mode=K.MODE_CONV, N=1, Hin=1, Win=cap, Hout=1, Wout=cap, R=1, S=1, stride=1, pad=0, dil=1, rows=rows)
        wkw = dict(cout=Cout, mode=K.MODE_CONV, N=1, Hin=1, Win=cap, Hout=1, Wout=cap, R=1, S=1, stride=1, pad=0, dil=1, rows=rows)
    o_ref = K.conv_fprop(z, w, **fkw)
    o_xf = K.conv_fprop(y, w, xf=xf, **fkw)
    assert torch.equal(o_xf[:live], o_ref[:live]), float((o_xf[:live].float() - o_ref[:live].float()).abs().max())
    for od in (torch.float32, dtype):
        g_ref = K.conv_wgrad(z, dy, out_dtype=od, **wkw)
        g_xf = K.conv_wgrad(y, dy, out_dtype=od, xf=xf, **wkw)
        assert torch.equal(g_xf, g_ref), (od, float((g_xf.float() - g_ref.float()).abs().max()))


SLAB_CASES = [
    # N, Cin, Cout, H, W: one 32-channel slab, one channel tile -- the persistent weights-once form (conv_halo3_slab_kernel)
    (4, 32, 32, 128, 128),          # 512 tiles: most workgroups walk one tile, some none
    (3, 32, 32, 200, 72),           # 1 125 tiles over 768 workgroups: one and two tiles per workgroup, ragged in x (72 = 4.5 tiles)
    (2, 32, 32, 256, 256),          # 2 048 tiles: two and three per workgroup -- the double buffer turns over
    (1, 32, 8, 40, 40),             # 8 output channels (three quarters of the lanes hold no column), ragged in y
    (2, 32, 16, 24, 56),
    (1, 32, 24, 9, 33),             # ragged both ways, H not a multiple of 8
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', SLAB_CASES)
def test_one_slab_persistent_form(case, dtype):
    """conv_halo3.hip's one-slab persistent form (round 6: Cin == 32, Cout <= 32 -- weights staged once per workgroup, halo image double-buffered, a
    list of tiles per workgroup; selected by the dispatch from 4 096 tiles up, forced here with mg_set_halo3_cfg(8, 32, 201)): forward and data
    gradient (taps mirrored) with every epilogue the layer family uses -- scale / shift / activation before or after, residuals at full and half
    resolution, output into a channel slice of a wider buffer, BatchNorm statistics -- against torch fp32 on the rounded operands and against the
    round-2 kernels (MG_HALO3 off); the operand transform must give the BITS of the stored form, statistics rows included."""
    import ctypes
    from maggie_amd import kernels as K, hip
    dev = _dev()
    N, Cin, Cout, H, W = case
    M = N * H * W
    rs = np.random.RandomState(Cout + H + W)
    lib = hip.lib()

    def t(a):
        return torch.from_numpy(a.astype(np.float32)).to(dev)
    x = t(rs.normal(size=(M, Cin))).to(dtype)
    w = t(rs.normal(size=(Cout, 9, Cin)) / np.sqrt(9 * Cin)).to(dtype)
    scale, shift = t(rs.uniform(0.5, 1.5, Cout)), t(rs.normal(size=Cout))
    res, res2 = t(rs.normal(size=(M, Cout))).to(dtype), t(rs.normal(size=(M, Cout))).to(dtype)
    variants = [dict(), dict(scale=scale, shift=shift, act=K.ACT_RELU), dict(shift=shift, act=K.ACT_LRELU, slope=0.2, res=res),
                dict(scale=scale, shift=shift, act=K.ACT_RELU, pre_act=True, res2=res2)]
    if H % 2 == 0 and W % 2 == 0:
        variants.append(dict(scale=scale, shift=shift, res=t(rs.normal(size=(N * (H // 2) * (W // 2), Cout))).to(dtype), res_mode=2))
    rows = K.conv_stat_rows(M, N, H, W)
    tol = _tol(dtype)
    try:
        for mode in (K.MODE_CONV, K.MODE_TCONV):
            geo = dict(N=N, Hin=H, Win=W, R=3, S=3, stride=1, pad=1, dil=1, mode=mode)
            xi = x.float().view(N, H, W, Cin).permute(0, 3, 1, 2)
            wk = w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
            base = F.conv2d(xi, wk.flip(2, 3) if mode == K.MODE_TCONV else wk, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
            for kw in variants:
                outs = []
                for forced in (False, True):
                    lib.mg_set_halo3(ctypes.c_int(1 if forced else 0))
                    lib.mg_set_halo3_cfg(ctypes.c_int(8 if forced else 0), ctypes.c_int(32 if forced else 0), ctypes.c_int(201 if forced else 0))
                    st = torch.zeros(rows, 2 * Cout, device=dev)
                    wide = torch.zeros(M, Cout + 16, device=dev, dtype=dtype)
                    K.conv_fprop(x, w, stats=st, out=wide, yoff=8, cout=Cout, **geo, **kw)
                    torch.cuda.synchronize()
                    outs.append((wide, st.sum(0)))
                (yo, so), (yn, sn) = outs
                sl = 1.0 if kw.get('act', K.ACT_NONE) == K.ACT_NONE else (0.0 if kw['act'] == K.ACT_RELU else 0.2)
                r = base
                if kw.get('pre_act'):
                    r = torch.maximum(r, r * sl)
                r = r * kw.get('scale', torch.ones_like(scale)) + kw.get('shift', torch.zeros_like(shift))
                if 'res' in kw:
                    rr = kw['res'].float()
                    if kw.get('res_mode') == 2:
                        rr = rr.view(N, H // 2, W // 2, Cout).repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(M, Cout)
                    r = r + rr
                if not kw.get('pre_act'):
                    r = torch.maximum(r, r * sl)
                if 'res2' in kw:
                    r = r + kw['res2'].float()
                got = yn[:, 8:8 + Cout].float()
                assert bool((yn[:, :8] == 0).all()) and bool((yn[:, 8 + Cout:] == 0).all()), 'wrote outside its channel slice'
                assert float((got - r).abs().max()) <= tol * max(1.0, float(r.abs().max())), (mode, sorted(kw))
                assert float((got - yo[:, 8:8 + Cout].float()).abs().max()) <= tol * max(1.0, float(r.abs().max()))        # the round-2 kernel
                want = torch.cat([got.sum(0), (got * got).sum(0)])
                assert float(((sn - want).abs() / (want.abs() + 1.0)).max()) <= 2e-3
        # operand transform: raw producer output + (scale, shift, act) == the stored z, bit for bit, statistics rows included
        lib.mg_set_halo3(ctypes.c_int(1))
        lib.mg_set_halo3_cfg(ctypes.c_int(8), ctypes.c_int(32), ctypes.c_int(201))
        geo = dict(N=N, Hin=H, Win=W, R=3, S=3, stride=1, pad=1, dil=1, mode=K.MODE_CONV)
        for act in (0, 1, 2):
            y = t(rs.normal(0.3, 1.5, (M, Cin))).to(dtype)
            sc = t(rs.uniform(0.5, 1.5, Cin) * rs.choice([-1.0, 1.0], Cin))
            sh = t(rs.normal(0.0, 0.7, Cin))
            z = K.affine_act(y, sc, sh, act=act, slope=0.2)
            sa, sb = torch.zeros(rows, 2 * Cout, device=dev), torch.zeros(rows, 2 * Cout, device=dev)
            o_ref = K.conv_fprop(z, w, stats=sa, **geo)
            o_xf = K.conv_fprop(y, w, stats=sb, xf=(sc, sh, act, 0.2), **geo)
            assert torch.equal(o_ref, o_xf) and torch.equal(sa, sb), (act, int((o_ref != o_xf).sum()))
    finally:
        lib.mg_set_halo3(ctypes.c_int(1))
        lib.mg_set_halo3_cfg(ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0))
