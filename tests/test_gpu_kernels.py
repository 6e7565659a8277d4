"""GPU parity of the HBM-bound kernels (BatchNorm family, region/bit-plane ops, dense<->sparse gathers, alpha planes)
through the C ABI, against the CPU oracle (oracle/region.py: bit-exact) or a plain PyTorch fp32 CPU reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def _q(dtype):
    return (lambda t: t.to(dtype).float()) if dtype != torch.float32 else (lambda t: t)


def _tol(dtype):
    return 1e-5 if dtype == torch.float32 else 1.6e-2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('act', [0, 1, 2])
def test_bn_train_forward_backward(dtype, act):
    from maggie_amd import kernels as K
    dev = _dev()
    rs = np.random.RandomState(1)
    M, C = 1000, 64
    q = _q(dtype)
    x = q(torch.from_numpy(rs.normal(1.0, 2.0, (M, C)).astype(np.float32))).requires_grad_(True)
    res = q(torch.from_numpy(rs.normal(size=(M, C)).astype(np.float32))).requires_grad_(True)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)).requires_grad_(True)
    beta = torch.from_numpy(rs.normal(size=C).astype(np.float32)).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    y_ref = F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-5) + res
    y_ref = [lambda t: t, F.relu, lambda t: F.leaky_relu(t, 0.2)][act](y_ref)
    gy = q(torch.from_numpy(rs.normal(size=(M, C)).astype(np.float32)))
    y_ref.backward(gy)

    xd, resd = x.detach().to(dev, dtype), res.detach().to(dev, dtype)
    rmd, rvd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    stats = K.colstats(xd)
    scale, shift, mean, invstd = K.bn_finalize(stats, M, gamma.detach().to(dev), beta.detach().to(dev), rmd, rvd, 0.1, 1e-5)
    st2 = K.colstats_centered(xd)
    sc2, sh2, _, _ = K.bn_finalize(st2, M, gamma.detach().to(dev), beta.detach().to(dev), None, None, 0.1, 1e-5, centered=True)
    assert torch.allclose(sc2, scale, rtol=1e-4) and torch.allclose(sh2, shift, rtol=1e-3, atol=1e-4)
    y = K.affine_act(xd, scale, shift, res=resd, act=act, slope=0.2)
    tol = _tol(dtype)
    assert (y.float().cpu() - y_ref.detach()).abs().max() <= tol * y_ref.abs().max() + 1e-6
    assert torch.allclose(rmd.cpu(), rm, atol=1e-4) and torch.allclose(rvd.cpu(), rv, atol=1e-3)
    dx, dres, sums = K.bn_backward(gy.to(dev, dtype), y, xd, scale, mean, invstd, M, act=act, slope=0.2, want_dres=True)
    assert (dx.float().cpu() - x.grad).abs().max() <= max(tol, 2e-4) * x.grad.abs().max() * 2
    assert (dres.float().cpu() - res.grad).abs().max() <= tol * res.grad.abs().max() + 1e-6
    assert torch.allclose(sums[:C].cpu(), beta.grad, rtol=2e-2, atol=2e-2 * beta.grad.abs().max().item())
    assert torch.allclose(sums[C:].cpu(), gamma.grad, rtol=2e-2, atol=2e-2 * gamma.grad.abs().max().item())


@pytest.mark.parametrize('rows', [(2, 24, 40), (4, 64, 64), (1, 8, 16)])
def test_batchnorm_large_mean_fp32_in_deterministic_mode(rows):
    """inputs with |mean| ~ 1e3 std through the PRODUCT's training BatchNorm (functional.batch_norm_act, deterministic
    mode = the default) in fp32. The one-pass E[x^2] - E[x]^2 would lose var's digits here (mean^2 / var = 1e6 -> ~6 % of var in fp32); fp32
    layers up to EXACT_STATS_ROWS rows keep the two-pass variance in its ordered form, the <= 1024-row layers the one-workgroup form."""
    from maggie_amd import functional as MF, hip
    assert hip.DETERMINISTIC
    dev = _dev()
    N, H, W = rows
    C = 64
    rs = np.random.RandomState(11)
    mean = rs.uniform(-1.0, 1.0, C).astype(np.float32) * 1000.0
    x = torch.from_numpy((rs.normal(size=(N, H, W, C)).astype(np.float32) + mean)).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    bn_ref = torch.nn.BatchNorm2d(C).double()
    bn_ref.load_state_dict(bn.state_dict())
    xr = x.detach().double().permute(0, 3, 1, 2).requires_grad_(True)
    y_ref = F.relu(bn_ref(xr))
    gy = torch.from_numpy(rs.normal(size=(N, H, W, C)).astype(np.float32))
    y_ref.backward(gy.double().permute(0, 3, 1, 2))
    bn.to(dev).train()
    xd = x.detach().to(dev).requires_grad_(True)
    outs = []
    for _ in range(2):                                          # ... and the same bits twice
        xd.grad = None
        bn2 = torch.nn.BatchNorm2d(C).to(dev)
        bn2.load_state_dict(bn.state_dict())
        y = MF.batch_norm_act(xd, bn2, act=MF.ACT_RELU)
        y.backward(gy.to(dev))
        outs.append((y.detach().clone(), xd.grad.clone(), bn2.running_var.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    y, dx, rv = outs[0]
    yr = y_ref.detach().permute(0, 2, 3, 1).float()
    # x carries ~1e3 * 2^-24 = 6e-5 of absolute rounding per element, i.e. 6e-5 std: the normalised output cannot be better than ~1e-4
    assert (y.cpu() - yr).abs().max() <= 1e-3, (y.cpu() - yr).abs().max()
    assert torch.allclose(rv.cpu(), bn_ref.running_var.float(), rtol=2e-3)
    dxr = xr.grad.permute(0, 2, 3, 1).float()
    # an activation whose pre-image sits within the forward's ~1e-4 of zero takes the other ReLU branch than in the fp64 reference (a handful of the
    # 1e5 elements; each moves dx by its whole g): those are excluded, everything else must agree
    near0 = (yr.abs() < 1e-3) & ((y.cpu() > 0) != (yr > 0))
    assert float(near0.float().mean()) <= 1e-3
    assert ((dx.cpu() - dxr).abs() * (~near0).float()).max() <= 2e-3 * dxr.abs().max(), ((dx.cpu() - dxr).abs() * (~near0).float()).max()


def test_bn_fold_and_pool():
    from maggie_amd import kernels as K
    dev = _dev()
    rs = np.random.RandomState(2)
    C = 32
    g, b, rm, rv = [torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)) for _ in range(4)]
    scale, shift = K.bn_fold(g.to(dev), b.to(dev), rm.to(dev), rv.to(dev), 1e-5)
    x = torch.from_numpy(rs.normal(size=(2, C, 8, 6)).astype(np.float32))
    y_ref = F.batch_norm(x, rm, rv, g, b, False, 0.1, 1e-5)
    xr = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(dev)
    y = K.affine_act(xr, scale, shift).cpu().reshape(2, 8, 6, C).permute(0, 3, 1, 2)
    assert torch.allclose(y, y_ref, atol=1e-5)
    p = K.pool2x2(xr, 0, 2, 4, 3).cpu().reshape(2, 4, 3, C).permute(0, 3, 1, 2)
    assert torch.allclose(p, F.avg_pool2d(x, 2, 2), atol=1e-6)
    u = K.pool2x2(xr, 3, 2, 16, 12).cpu().reshape(2, 16, 12, C).permute(0, 3, 1, 2)
    assert torch.equal(u, F.interpolate(x, scale_factor=2, mode='nearest'))


@pytest.mark.parametrize('hw', [(64, 64), (96, 160), (40, 200)])
def test_compute_unknown_bit_exact(hw):
    """threshold + ellipse dilation == oracle.region.compute_unknown, eval width and random train widths."""
    from maggie_amd import kernels as K
    from oracle import region
    dev = _dev()
    H, W = hw
    rs = np.random.RandomState(H + W)
    a = rs.uniform(size=(5, H, W)).astype(np.float32)
    a[a < 0.7] = 0.0
    a[a > 0.95] = 1.0
    a[0, :3, :3] = 0.5
    a[1, -2:, -2:] = 0.5
    a[2] = 0
    a[2, H // 2, W - 1] = 0.3
    bits = K.bits_pack(torch.from_numpy(a).to(dev))
    for k in (30, 27, 15):
        ref = region.compute_unknown(a, k, False)
        out = K.bits_unpack_u8(K.bits_dilate(bits, W, width=k // 2), W).cpu().numpy()
        assert np.array_equal(out, ref), k
    widths = rs.randint(1, 30, size=5).astype(np.int32)
    ref = region.compute_unknown(a, 30, True, widths=widths)
    out = K.bits_unpack_u8(K.bits_dilate(bits, W, widths=torch.from_numpy(widths).to(dev)), W).cpu().numpy()
    assert np.array_equal(out, ref)
    for k in range(1, 30):                                  # every structuring element incl. even widths
        ref = region.compute_unknown(a[:2], 30, True, widths=np.array([k, k]))
        out = K.bits_unpack_u8(K.bits_dilate(bits[:2].contiguous(), W, width=k), W).cpu().numpy()
        assert np.array_equal(out, ref), k


@pytest.mark.parametrize('hw', [(64, 64), (72, 136)])
def test_active_pyramid_and_tables_bit_exact(hw):
    from maggie_amd import kernels as K
    from oracle import region
    dev = _dev()
    H, W = hw
    rs = np.random.RandomState(7)
    roi = (rs.uniform(size=(3, H, W)) > 0.93).astype(np.uint8)
    roi[1] = 0
    roi[2, 10:30, 5:50] = 1
    pyr = region.active_pyramid(roi)
    b1 = K.bits_pack(torch.from_numpy(roi).to(dev), mode=1)
    levels = [(b1, H, W)]
    for _ in range(3):
        bb, hh, ww = K.bits_downsample(levels[-1][0], levels[-1][2])
        levels.append((bb, hh, ww))
    ranks, coords = [], []
    for (bb, hh, ww), act in zip(levels, pyr):
        assert np.array_equal(K.bits_unpack_u8(bb, ww).cpu().numpy().astype(bool), act)
        rowoff, wordoff = K.bits_rank(bb, ww)
        R = int(rowoff[-1].item())
        assert R == int(act.sum())
        c = K.bits_coords(bb, wordoff, ww, R)
        assert np.array_equal(c.cpu().numpy(), region.coords_of(act))
        ranks.append(wordoff)
        coords.append(c)
    for lv in (0, 2):
        bb, hh, ww = levels[lv]
        nbr = K.gather_table(coords[lv], 3, 0, bb, ranks[lv], hh, ww)
        assert np.array_equal(nbr.cpu().numpy(), region.subm_neighbors(pyr[lv]))
    for lv in (0, 1, 2):
        bc, hc, wc = levels[lv + 1]
        nbr = K.gather_table(coords[lv], 3, 1, bc, ranks[lv + 1], hc, wc)
        ref = region.inverse_neighbors(pyr[lv], pyr[lv + 1])
        assert np.array_equal(nbr.cpu().numpy(), ref)
        # strided table = transpose of the inverse table
        bf, hf, wf = levels[lv]
        down = K.gather_table(coords[lv + 1], 3, 2, bf, ranks[lv], hf, wf).cpu().numpy()
        chk = np.full_like(down, -1)
        rr, kk = np.nonzero(ref >= 0)
        chk[ref[rr, kk], kk] = rr
        assert np.array_equal(down, chk)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_gather_scatter_rows_and_planes(dtype):
    from maggie_amd import kernels as K
    from oracle import region
    dev = _dev()
    rs = np.random.RandomState(8)
    N, n_i, H, W, C = 2, 3, 16, 24, 32
    act = rs.uniform(size=(N * n_i, H, W)) > 0.8
    co = region.coords_of(act)
    R = co.shape[0]
    q = _q(dtype)
    dense = q(torch.from_numpy(rs.normal(size=(N, H, W, C)).astype(np.float32)))
    mul = torch.from_numpy(rs.normal(size=(N, 10, C)).astype(np.float32))
    fr = torch.from_numpy(co[:, 0] // n_i).long()
    inst = torch.from_numpy(co[:, 0] % n_i).long()
    yy, xx = torch.from_numpy(co[:, 1]).long(), torch.from_numpy(co[:, 2]).long()
    ref = dense[fr, yy, xx] * mul[fr, inst]
    cod = torch.from_numpy(co).to(dev)
    big = torch.zeros((R, 2 * C), device=dev, dtype=dtype)
    K.gather_rows(dense.to(dev, dtype), cod, n_i, mul=mul.to(dev), out=big, yoff=C)
    assert (big[:, C:].float().cpu() - ref).abs().max() <= _tol(dtype) * ref.abs().max()
    g = q(torch.from_numpy(rs.normal(size=(R, C)).astype(np.float32)))
    dd, dm = K.gather_rows_bwd(g.to(dev, dtype), cod, n_i, (N, H, W, C), mul=mul.to(dev), dense=dense.to(dev, dtype), want_dmul=True)
    dref = torch.zeros(N, H, W, C).index_put((fr, yy, xx), g * mul[fr, inst], accumulate=True)
    mref = torch.zeros(N, 10, C).index_put((fr, inst), g * dense[fr, yy, xx], accumulate=True)
    assert torch.allclose(dd.cpu(), dref, atol=1e-4) and torch.allclose(dm.cpu(), mref, rtol=1e-3, atol=1e-3)
    # atomic-free variant driven by the level's bit planes / ranks
    bits = K.bits_pack(torch.from_numpy(act.astype(np.uint8)).to(dev), mode=1)
    rowoff, wordoff = K.bits_rank(bits, W)
    dd2 = K.gather_rows_bwd_dense(g.to(dev, dtype), bits, wordoff, n_i, (N, H, W, C), mul=mul.to(dev))
    assert (dd2.float().cpu() - dref).abs().max() <= _tol(dtype) * dref.abs().max() + 1e-5
    vals = q(torch.from_numpy(rs.normal(size=(R, 1)).astype(np.float32)))
    plane = K.scatter_plane(vals.to(dev, dtype), 0, cod, N * n_i, H, W, -99.0).cpu()
    pref = torch.full((N * n_i, H, W), -99.0)
    pref[torch.from_numpy(co[:, 0]).long(), yy, xx] = vals[:, 0]
    assert torch.equal(plane, pref)
    back = K.gather_plane(plane.to(dev), cod, dtype).float().cpu()
    assert torch.equal(back, vals)


@pytest.mark.parametrize('scale,h,w', [(1, 8, 12), (4, 8, 12), (8, 8, 12), (4, 16, 24), (8, 8, 16), (8, 24, 16)])
def test_upsample_tanh(scale, h, w):
    """(h, w) multiples of 8 take the LDS-tiled atomic-free backward, the others the scatter kernel."""
    from maggie_amd import kernels as K
    dev = _dev()
    rs = np.random.RandomState(scale)
    N, C = 2, 3
    x = torch.from_numpy(rs.normal(size=(N, h, w, 16)).astype(np.float32) * 2)     # NHWC, only first C channels used
    xin = x[..., :C].permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    up = xin if scale == 1 else F.interpolate(xin, scale_factor=float(scale), mode='bilinear', align_corners=False)
    ref = (torch.tanh(up) + 1) / 2
    g = torch.from_numpy(rs.normal(size=tuple(ref.shape)).astype(np.float32))
    ref.backward(g)
    xd = x.to(dev)
    strides = (h * w * 16, 1, w * 16, 16)
    out = K.upsample_tanh(xd, strides, N, C, h, w, scale)
    assert torch.allclose(out.cpu(), ref.detach(), atol=2e-6)
    din = torch.zeros((N, h, w, 16), device=dev)
    K.upsample_tanh_bwd(g.to(dev), out, strides, N, C, h, w, scale, din)
    assert torch.allclose(din[..., :C].cpu().permute(0, 3, 1, 2), xin.grad, atol=1e-5)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_mask_embed(dtype):
    from maggie_amd import kernels as K
    from oracle import refmodel
    dev = _dev()
    rs = np.random.RandomState(9)
    N, H, W, n_m = 2, 32, 48, 10
    image = torch.from_numpy(rs.normal(size=(N, 3, H, W)).astype(np.float32))
    masks = torch.from_numpy((rs.uniform(size=(N, n_m, H // 8, W // 8)) > 0.6).astype(np.float32))
    table = torch.from_numpy(rs.normal(size=(n_m + 1, 3)).astype(np.float32)).requires_grad_(True)
    emb = refmodel.mask_id_embedding({'e.mask_embed_layer.weight': table}, 'e', F.interpolate(masks, size=(H, W), mode='nearest'))
    ref = torch.cat([image, emb], 1)
    out = K.mask_embed(image.to(dev), masks.to(dev), table.detach().to(dev), dtype)
    o = out.float().cpu()
    assert o[..., 6:].abs().max() == 0
    assert (o[..., :6].permute(0, 3, 1, 2) - ref.detach()).abs().max() <= _tol(dtype) * 4
    g = torch.from_numpy(rs.normal(size=(N, H, W, 8)).astype(np.float32))
    emb.backward(g[..., 3:6].permute(0, 3, 1, 2))
    dt = K.mask_embed_bwd(g.to(dev), masks.to(dev), (n_m + 1, 3))
    assert torch.allclose(dt.cpu(), table.grad, rtol=1e-3, atol=1e-3)


def test_gather_tables_real_roi_regression():
    """A detail region captured from the video model (thin bands reaching the right image border, 6 planes). An earlier
    table kernel was mis-compiled for it (non-deterministic -1 entries); must be exact and repeatable."""
    import os
    from maggie_amd import kernels as K
    from oracle import region
    dev = _dev()
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'roi_band_regression.npz'))
    shape = tuple(z['shape'])
    roi = np.unpackbits(z['bits'])[:int(np.prod(shape))].reshape(shape)
    pyr = region.active_pyramid(roi)
    H, W = shape[-2:]
    b1 = K.bits_pack(torch.from_numpy(roi).to(dev), mode=1)
    b2, h2, w2 = K.bits_downsample(b1, W)
    ro1, wo1 = K.bits_rank(b1, W)
    ro2, wo2 = K.bits_rank(b2, w2)
    c1 = K.bits_coords(b1, wo1, W, int(ro1[-1]))
    c2 = K.bits_coords(b2, wo2, w2, int(ro2[-1]))
    ref_inv = region.inverse_neighbors(pyr[0], pyr[1])
    ref_sub = region.subm_neighbors(pyr[0])
    for _ in range(3):
        assert np.array_equal(K.gather_table(c1, 3, 1, b2, wo2, h2, w2).cpu().numpy(), ref_inv)
        assert np.array_equal(K.gather_table(c1, 3, 0, b1, wo1, H, W).cpu().numpy(), ref_sub)
        down = K.gather_table(c2, 3, 2, b1, wo1, H, W).cpu().numpy()
        chk = np.full_like(down, -1)
        rr, kk = np.nonzero(ref_inv >= 0)
        chk[ref_inv[rr, kk], kk] = rr
        assert np.array_equal(down, chk)


@pytest.mark.parametrize('hw', [(64, 64), (40, 72)])
def test_fused_matting_losses_forward_backward(hw):
    """Fused L1 + Laplacian-pyramid + Sobel losses (C ABI) against the oracle's torch-CPU restatement incl. gradients."""
    from maggie_amd import functional as MF
    from oracle import refmodel
    dev = _dev()
    H, W = hw
    rs = np.random.RandomState(H)
    P = (2, 5)
    pred = torch.from_numpy(rs.uniform(size=P + (H, W)).astype(np.float32))
    tgt = torch.from_numpy(rs.uniform(size=P + (H, W)).astype(np.float32))
    wgt = torch.from_numpy((rs.uniform(size=P + (H, W)) > 0.4).astype(np.float32))
    wgt[:, 1] = 0
    wgt[:, 3] *= 2                         # os8-style weights in {0, 1, 2}
    wgt[0, 4, :3, :] = 1
    wgt[0, 4, :, -3:] = 1                  # make sure borders are exercised
    pr = pred.clone().requires_grad_(True)
    v = lambda t: t.reshape(-1, 1, H, W)
    ref = torch.stack([refmodel.regression_loss(pr, tgt, wgt), refmodel.lap_loss(v(pr), v(tgt), v(wgt)), refmodel.grad_loss(pr, tgt, wgt)])
    coefs = torch.tensor([1.3, 0.7, 2.1])
    (ref * coefs).sum().backward()
    pd = pred.clone().to(dev).requires_grad_(True)
    out = torch.stack(MF.matting_losses(pd, tgt.to(dev), wgt.to(dev)))
    assert torch.allclose(out.cpu(), ref.detach(), rtol=2e-5, atol=1e-6), (out.cpu(), ref)
    (out * coefs.to(dev)).sum().backward()
    err = (pd.grad.cpu() - pr.grad).abs().max().item()
    assert err <= 2e-3 * pr.grad.abs().max().item(), (err, pr.grad.abs().max().item())
    assert float(pd.grad[:, 1].abs().max()) == 0.0


# ----------------------------------------------------------------------------------------------------------------------
# instance-token <-> feature cross attention (mg_attn_*), fp32, against plain torch autograd on the CPU
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('L,nid', [(200, 11), (4096, 11), (333, 1)])
def test_attention_tokens_from_features(L, nid):
    from maggie_amd import functional as MF
    dev = _dev()
    rs = np.random.RandomState(L + nid)
    B, T, D = 2, 10, 128
    t = lambda *sh: torch.from_numpy(rs.normal(size=sh).astype(np.float32))
    qk, btab, feat = t(B, T, D) * 0.3, t(B, T, nid), t(B, L, D)
    ids = torch.from_numpy(rs.randint(0, nid, size=(B, L)).astype(np.int32))
    r_ctx, r_p = t(B, T, D), t(B, T, L)
    scale = 1.0 / np.sqrt(D)
    ref_in = [x.clone().requires_grad_(True) for x in (qk, btab, feat)]
    s = torch.matmul(ref_in[0], ref_in[2].transpose(1, 2)) + torch.gather(ref_in[1], 2, ids.long()[:, None, :].expand(-1, T, -1))
    p_ref = torch.softmax(s * scale, -1)
    ctx_ref = torch.matmul(p_ref, ref_in[2])
    ((ctx_ref * r_ctx).sum() + (p_ref * r_p).sum()).backward()
    gin = [x.clone().to(dev).requires_grad_(True) for x in (qk, btab, feat)]
    p, ctx = MF.attn_tokens_from_features(gin[0], gin[1], gin[2], ids.to(dev), scale)
    ((ctx * r_ctx.to(dev)).sum() + (p * r_p.to(dev)).sum()).backward()
    # (with a single ID row the bias gradient is identically 0 by softmax shift invariance: absolute floor)
    close = lambda a, b, tol: float((a.cpu() - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-2)
    assert close(p.detach(), p_ref.detach(), 2e-5) and close(ctx.detach(), ctx_ref.detach(), 2e-5)
    for g, r, name in zip(gin, ref_in, ('dqk', 'dbtab', 'dfeat')):
        assert close(g.grad, r.grad, 2e-4), name


@pytest.mark.parametrize('L,nid,masked', [(200, 11, True), (4096, 11, True), (333, 1, False)])
def test_attention_features_from_tokens(L, nid, masked):
    from maggie_amd import functional as MF
    dev = _dev()
    rs = np.random.RandomState(L + 7 * nid)
    B, T, D = 2, 10, 128
    t = lambda *sh: torch.from_numpy(rs.normal(size=sh).astype(np.float32))
    feat, kq, b2, vp, ob = t(B, L, D), t(B, T, D) * 0.3, t(B, nid, T), t(B, T, D), t(D)
    ids = torch.from_numpy(rs.randint(0, nid, size=(B, L)).astype(np.int32))
    pad = torch.zeros((B, T), dtype=torch.bool)
    if masked:
        pad[0, 2:] = True
        pad[1, 7:] = True
    r_out = t(B, L, D)
    scale = 1.0 / np.sqrt(D)
    ref_in = [x.clone().requires_grad_(True) for x in (feat, kq, b2, vp, ob)]
    s = torch.matmul(ref_in[0], ref_in[1].transpose(1, 2)) + torch.gather(ref_in[2], 1, ids.long()[:, :, None].expand(-1, -1, T))
    s = (s * scale).masked_fill(pad[:, None, :], float('-inf'))
    out_ref = torch.matmul(torch.softmax(s, -1), ref_in[3]) + ref_in[4]
    (out_ref * r_out).sum().backward()
    gin = [x.clone().to(dev).requires_grad_(True) for x in (feat, kq, b2, vp, ob)]
    out = MF.attn_features_from_tokens(gin[0], gin[1], gin[2], gin[3], gin[4], pad.to(dev) if masked else None, ids.to(dev), scale)
    (out * r_out.to(dev)).sum().backward()
    close = lambda a, b, tol: float((a.cpu() - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-2)
    assert close(out.detach(), out_ref.detach(), 2e-5)
    for g, r, name in zip(gin, ref_in, ('dfeat', 'dkq', 'db2', 'dvp', 'dobias')):
        assert close(g.grad, r.grad, 2e-4), name
    # the table in the layout the token-side linear writes it, (B, T, NID) with tn=True: same bits, gradient in that layout
    gtn = [x.clone().to(dev).requires_grad_(True) for x in (feat, kq, b2.transpose(1, 2).contiguous(), vp, ob)]
    out_tn = MF.attn_features_from_tokens(gtn[0], gtn[1], gtn[2], gtn[3], gtn[4], pad.to(dev) if masked else None, ids.to(dev), scale, tn=True)
    (out_tn * r_out.to(dev)).sum().backward()
    assert torch.equal(out_tn, out)
    for i, (a, b_) in enumerate(zip(gtn, gin)):
        assert torch.equal(a.grad, b_.grad.transpose(1, 2) if i == 2 else b_.grad), i


def test_postprocess_alpha_matches_reference_fixture_and_oracle():
    """mg_postprocess_alpha through maggie_amd.utils.postprocessing.reverse_transform_tensor: against the reference's own outputs
    (tests/golden/postprocess_pinned.npz) and, with snapping, against oracle/postprocess.py."""
    from helpers import load_golden
    from maggie_amd.utils.postprocessing import reverse_transform_tensor
    from oracle import postprocess as opp
    dev = _dev()
    gold = load_golden('postprocess_pinned.npz')
    rs = np.random.RandomState(21)
    cases = {'resize_pad': ((2, 3, 40, 56), [{'name': ['resize'], 'ori_size': (torch.tensor(37), torch.tensor(61))},
                                            {'name': ['padding'], 'pad_size': (torch.tensor(5), torch.tensor(8))}]),
             'pad_resize_same': ((1, 2, 32, 48), [{'name': 'resize', 'ori_size': (29, 48)}, {'name': 'padding', 'pad_size': (3, 0)}]),
             'resize_only': ((3, 24, 24), [{'name': 'resize', 'ori_size': (50, 33)}])}
    for key, (shape, info) in cases.items():
        x = torch.from_numpy(rs.uniform(-0.05, 1.05, size=shape).astype(np.float32))
        y = reverse_transform_tensor(x.to(dev), info).cpu().numpy()
        assert y.shape == gold[key].shape
        assert np.abs(y - gold[key]).max() <= 2e-6, key
        ys = reverse_transform_tensor(x.to(dev), info, snap=True).cpu().numpy()
        ref = opp.snap_alpha(gold[key])
        flip = (np.abs(gold[key] - 1 / 255.0) < 1e-5) | (np.abs(gold[key] - 254 / 255.0) < 1e-5)       # knife-edge values may snap either way
        assert np.abs(ys - ref)[~flip].max() <= 2e-6, key
    # a full-size call: 1080p output from a padded 512x512 prediction, values stay in [0, 1] and endpoints are exact copies
    x = torch.rand((1, 3, 2, 512, 512), device=dev)
    info = [{'name': 'resize', 'ori_size': (1080, 1920)}, {'name': 'padding', 'pad_size': (32, 0)}]
    y = reverse_transform_tensor(x, info, snap=True)
    assert y.shape == (1, 3, 2, 1080, 1920) and float(y.min()) >= 0.0 and float(y.max()) <= 1.0
    x0 = opp.snap_alpha(x[0, 0, 0, 0, 0].item())
    assert abs(float(y[0, 0, 0, 0, 0]) - float(x0)) <= 1e-6


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_conv_gru_gate_kernels(dtype):
    """mg_gru_gate_{fwd,bwd} / mg_gru_out_{fwd,bwd} against the torch formulas of conv_gru.py:22-27 (autograd on the CPU)."""
    from maggie_amd import functional as MF
    dev = _dev()
    rs = np.random.RandomState(8)
    q = (lambda t: t.to(dtype).float()) if dtype != torch.float32 else (lambda t: t)
    t = lambda *sh: q(torch.from_numpy(rs.normal(size=sh).astype(np.float32)))
    b, H, W, C = 2, 5, 6, 16
    rz, x, h, cp, g1, g2 = t(b, H, W, 2 * C), t(b, H, W, C), t(b, H, W, C), t(b, H, W, C), t(b, H, W, 2 * C), t(b, H, W, C)
    ref = [v.clone().requires_grad_(True) for v in (rz, x, h, cp)]
    r, z = torch.sigmoid(ref[0]).split(C, -1)
    xrh_ref = torch.cat([ref[1], r * ref[2]], -1)
    hn_ref = (1 - z) * ref[2] + z * torch.tanh(ref[3])
    ((xrh_ref * g1).sum() + (hn_ref * g2).sum()).backward()
    gin = [v.clone().to(dev, dtype).requires_grad_(True) for v in (rz, x, h, cp)]
    xrh = MF.GruGate.apply(gin[0], gin[1], gin[2])
    hn = MF.GruOut.apply(gin[0], gin[3], gin[2])
    ((xrh.float() * g1.to(dev)).sum() + (hn.float() * g2.to(dev)).sum()).backward()
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    close = lambda a, c: float((a.float().cpu() - c).abs().max()) <= tol * max(float(c.abs().max()), 1.0)
    assert close(xrh.detach(), xrh_ref.detach()) and close(hn.detach(), hn_ref.detach())
    for gi, ri, name in zip(gin, ref, ('drz', 'dx', 'dh', 'dc')):
        assert close(gi.grad, ri.grad), name


def test_temporal_fuse_kernel():
    """mg_temporal_fuse against the torch restatement of maggie_temp.py:34-77 (the oracle's decoder_video epilogue uses the same)."""
    from maggie_amd import kernels as K
    dev = _dev()
    rs = np.random.RandomState(5)
    a = torch.from_numpy(rs.uniform(size=(3, 4, 16, 24)).astype(np.float32))
    df = torch.from_numpy(rs.uniform(size=(3, 4, 16, 24)).astype(np.float32))
    db = torch.from_numpy(rs.uniform(size=(3, 4, 16, 24)).astype(np.float32))
    for prev in (None, torch.from_numpy(rs.uniform(size=(4, 16, 24)).astype(np.float32))):
        al = a.clone()
        pv = al[0] if prev is None else prev
        f, bk = (df > 0.5).float(), (db > 0.5).float()
        p01 = pv * (1 - f[1]) + al[1] * f[1]
        p21 = al[2] * (1 - bk[1]) + al[1] * bk[1]
        p01 = torch.where((p01 - p21).abs() > 0, al[1], p01)
        p12 = p01 * (1 - f[2]) + al[2] * f[2]
        out = K.temporal_fuse_(a.clone().to(dev), None if prev is None else prev.to(dev), df.to(dev), db.to(dev)).cpu()
        assert torch.equal(out[0], a[0]) and torch.allclose(out[1], p01, atol=1e-7) and torch.allclose(out[2], p12, atol=1e-7)
    # a 5-frame clip (BASELINE configs[4] geometry): "t+1" is the LAST frame (maggie_temp.py:48), frames 1 and 2 are rewritten
    a5 = torch.from_numpy(rs.uniform(size=(5, 2, 8, 24)).astype(np.float32))
    df5, db5 = (torch.from_numpy(rs.uniform(size=(5, 2, 8, 24)).astype(np.float32)) for _ in range(2))
    f, bk = (df5 > 0.5).float(), (db5 > 0.5).float()
    p01 = a5[0] * (1 - f[1]) + a5[1] * f[1]
    p21 = a5[4] * (1 - bk[1]) + a5[1] * bk[1]
    p01 = torch.where((p01 - p21).abs() > 0, a5[1], p01)
    p12 = p01 * (1 - f[2]) + a5[4] * f[2]
    out = K.temporal_fuse_(a5.clone().to(dev), None, df5.to(dev), db5.to(dev)).cpu()
    assert torch.equal(out[0], a5[0]) and torch.equal(out[3], a5[3]) and torch.equal(out[4], a5[4])
    assert torch.allclose(out[1], p01, atol=1e-7) and torch.allclose(out[2], p12, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_weight_bank_matches_per_parameter_conversion(dtype):
    """One-launch conversion of all sparse-head parameters (mg_weight_bank) == the per-parameter cast/pad/permute/flip chains,
    bit for bit, forward (kernel layout + input-gradient twin) and backward (gradient back in the parameter's layout)."""
    from maggie_amd import functional as MF
    from maggie_amd.network import build_model
    from maggie_amd.utils import config
    DEV = _dev()
    torch.manual_seed(5)
    model, _ = build_model(config.model_config('image'))
    dec = model.decoder.to(DEV)
    for p in dec.parameters():
        p.data.normal_()
    plan, slots = dec._weight_bank_plan()
    assert plan.n == len(slots) >= 25
    outs = MF.weight_bank(plan, dtype)
    n_twin = 0
    for (m, what), it, o in zip(slots, plan.items, outs):
        p, (co, taps, ci), co_pad, ci_pad, flip_t, is_bias = it
        if is_bias:
            ref = F.pad(p.detach().float(), (0, ci_pad - ci))
            assert o.dtype == torch.float32 and torch.equal(o, ref)
            continue
        ref = F.pad(p.detach().reshape(co, taps, ci), (0, ci_pad - ci, 0, 0, 0, co_pad - co)).to(dtype)
        assert o.dtype == dtype and o.shape == ref.shape and torch.equal(o, ref), (what, tuple(p.shape))
        twin = ref.permute(2, 1, 0)
        if flip_t:
            twin = twin.flip(1)
            n_twin += 1
        assert torch.equal(o._mg_wt, twin.contiguous())
    assert n_twin == 7                                            # the 3x3 submanifold convs
    grads = [torch.randn_like(o) for o in outs]
    torch.autograd.backward(outs, grads)
    for it, g in zip(plan.items, grads):
        p, (co, taps, ci), co_pad, ci_pad, flip_t, is_bias = it
        ref = g.float()[:ci] if is_bias else g.float()[:co, :, :ci]
        assert p.grad.dtype == torch.float32 and torch.equal(p.grad.reshape(ref.shape), ref)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('M,C', [(0, 32), (1, 8), (777, 64), (20000, 32)])
def test_bias_act_bwd(dtype, M, C):
    from maggie_amd import kernels as K
    DEV = _dev()
    torch.manual_seed(M + C)
    dy = torch.randn(M, C, device=DEV).to(dtype)
    y = torch.randn(M, C, device=DEV).to(dtype)
    g, db = K.bias_act_bwd(dy, y, True)
    ref = dy * (y > 0).to(dtype)
    assert torch.equal(g, ref)
    assert torch.allclose(db, ref.double().sum(0).float(), rtol=1e-4, atol=1e-3 * max(1.0, M ** 0.5))
    g2, db2 = K.bias_act_bwd(dy, None, True)
    assert g2 is dy and torch.allclose(db2, dy.double().sum(0).float(), rtol=1e-4, atol=1e-3 * max(1.0, M ** 0.5))
    g3, db3 = K.bias_act_bwd(dy, y, False)
    assert db3 is None and torch.equal(g3, ref)


@pytest.mark.gpu
def test_device_preprocessor_matches_reference_fixture_and_oracle():
    """maggie_amd.utils.preprocess (mg_preprocess_image / mg_preprocess_planes) against the reference's own outputs
    (tests/golden/preprocess_pinned.npz) and, at the bench geometry, against the oracle: bit-exact."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import load_golden, preprocess_inputs, PREPROCESS_CASES
    from maggie_amd.utils.preprocess import DevicePreprocessor, normalize_frames, scale_planes
    from oracle import preprocess as pre
    dev = _dev()
    gold = load_golden('preprocess_pinned.npz')
    for key in PREPROCESS_CASES:
        frames, alphas, masks, max_inst = preprocess_inputs(key)
        ids = None if max_inst is None else [int(i) for i in gold[key + '.slot_ids']]
        out = DevicePreprocessor(max_inst=10, device=dev)(frames, alphas, masks, slot_ids=ids)
        for name in ('image', 'alpha', 'mask'):
            got = out[name].cpu().numpy()
            assert got.shape == gold['%s.%s' % (key, name)].shape, (key, name)
            assert np.array_equal(got, gold['%s.%s' % (key, name)]), (key, name)
    rs = np.random.RandomState(0)
    frames = rs.randint(0, 256, size=(4, 1, 512, 512, 3)).astype(np.uint8)                # (b, n_f, H, W, 3)
    planes = rs.randint(0, 256, size=(4, 2, 512, 512)).astype(np.uint8)
    img = normalize_frames(frames, device=dev)
    assert img.shape == (4, 1, 3, 512, 512)
    assert np.array_equal(img.cpu().numpy(), pre.normalize_frames(frames, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)))
    ids = [7, 2]
    for size, thresh in ((None, 5), ((64, 64), 0), ((37, 91), 0)):
        got = scale_planes(planes, 10, ids, size, thresh, device=dev).cpu().numpy()
        assert np.array_equal(got, pre.scale_planes(planes, 10, ids, size, thresh)), (size, thresh)
    with pytest.raises(ValueError):
        scale_planes(planes, 10, [1, 1], device=dev)
    with pytest.raises(TypeError):
        normalize_frames(frames.astype(np.float32), device=dev)


@pytest.mark.gpu
def test_device_metrics_match_reference_fixture_and_oracle():
    """maggie_amd.utils.metric (mg_metric_plane_sums / mg_metric_grad / mg_metric_dtssd) against the reference's own metric classes
    (tests/golden/metric_pinned.npz) and, at 512x512, against the oracle. fp32 planes, fp64 accumulation: rtol 2e-5."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import load_golden, metric_inputs, METRIC_CASES
    from maggie_amd.utils import metric as dm
    from oracle import metric as om
    dev = _dev()
    gold = load_golden('metric_pinned.npz')
    T = lambda a: None if a is None else torch.from_numpy(a).to(dev)        # noqa: E731
    for key in METRIC_CASES:
        pred, gt, tri = metric_inputs(key)
        ms = dm.build_metric(['SAD', 'MSE', 'MAD', 'Grad', 'dtSSD'])
        for name, m in ms.items():
            r = m.update(T(pred), T(gt), T(tri))
            ref = gold['%s.%s' % (key, name)]                      # [update() return, score, count, average()]
            assert m.count == ref[2], (key, name)
            assert abs(m.score - ref[1]) <= 2e-5 * abs(ref[1]) + 1e-9, (key, name, m.score, ref[1])
            assert abs(r - ref[0]) <= 2e-5 * abs(ref[0]) + 1e-9 and abs(m.average() - ref[3]) <= 2e-5 * abs(ref[3]) + 1e-9
            m.update(T(pred), T(gt), T(tri))                       # accumulates like the reference
            assert m.count == 2 * ref[2] and abs(m.score - 2 * ref[1]) <= 4e-5 * abs(ref[1]) + 1e-9
            m.reset()
            assert m.score == 0 and m.count == 0
    rs = np.random.RandomState(9)
    shape = (3, 2, 512, 512)                                       # (T, n_i, H, W): one clip of the bench geometry
    pred = rs.rand(*shape).astype(np.float32)
    gt = np.clip(pred + rs.normal(0, 0.05, size=shape), 0, 1).astype(np.float32)
    tri = rs.randint(0, 3, size=shape).astype(np.float32)
    for name, fn in (('SAD', om.sad), ('MSE', om.mse), ('MAD', om.mad), ('Grad', om.grad), ('dtSSD', om.dtssd)):
        for t in (tri, None):
            m = dm.build_metric([name])[name]
            m.update(T(pred), T(gt), T(t))
            score, count = fn(pred, gt, t)
            assert m.count == count and abs(m.score - score) <= 5e-5 * abs(score) + 1e-9, (name, t is None, m.score, score)
    with pytest.raises(NotImplementedError):
        dm.build_metric(['Conn'])


@pytest.mark.gpu
@pytest.mark.parametrize('max_norm', [None, 0.01, 1e6])
def test_flat_adamw_matches_torch_adamw(max_norm):
    """maggie_amd.optim.FlatAdamW (mg_adamw_flat) == torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW, step for step, including a
    OneCycleLR schedule, a parameter that gets no gradient on some steps, and a state_dict round trip into torch's own AdamW."""
    from maggie_amd.optim import FlatAdamW
    dev = _dev()
    torch.manual_seed(0)
    shapes = [(64, 3, 3, 32), (128,), (17, 5), (1,), (256, 256), (33,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(s, device=dev, generator=None) * 0.1) for s in shapes]     # noqa: E731
    ours = mk()
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    o_opt, r_opt = FlatAdamW(ours, max_grad_norm=max_norm, **kw), torch.optim.AdamW(ref, **kw)
    o_s = torch.optim.lr_scheduler.OneCycleLR(o_opt, max_lr=5e-3, total_steps=12, cycle_momentum=False)
    r_s = torch.optim.lr_scheduler.OneCycleLR(r_opt, max_lr=5e-3, total_steps=12, cycle_momentum=False)
    assert all(p.data_ptr() >= o_opt.flat_p.data_ptr() for p in ours) and ours[0].data_ptr() == o_opt.flat_p.data_ptr()
    for it in range(8):
        for i, (a, b) in enumerate(zip(ours, ref)):
            if i == 3 and it % 2:                                  # no gradient this step
                a.grad = b.grad = None
                continue
            g = torch.randn_like(a) * (10.0 if it == 5 else 0.01)
            a.grad, b.grad = g.clone(), g.clone()
        if max_norm is not None:
            total = torch.nn.utils.clip_grad_norm_([p for p in ref if p.grad is not None], max_norm)
        o_opt.step(); r_opt.step(); o_s.step(); r_s.step()
        if max_norm is not None:
            assert torch.allclose(o_opt.last_grad_norm[0], total, rtol=1e-5)
        for i, (a, b) in enumerate(zip(ours, ref)):                # incl. the parameter torch skips on odd steps (own step count)
            assert torch.allclose(a, b, rtol=2e-5, atol=2e-7), (it, i, float((a - b).abs().max()))
    assert o_opt._steps[3] == 4 and o_opt._steps[0] == 8
    # torch's AdamW resumes from our state
    sd = o_opt.state_dict()
    t_opt = torch.optim.AdamW([torch.nn.Parameter(p.detach().clone()) for p in ours], **kw)
    t_opt.load_state_dict(sd)
    assert float(t_opt.state[t_opt.param_groups[0]['params'][0]]['step']) == 8 and float(t_opt.state[t_opt.param_groups[0]['params'][3]]['step']) == 4
    assert torch.equal(t_opt.state[t_opt.param_groups[0]['params'][4]]['exp_avg'], o_opt.state[ours[4]]['exp_avg'])
    # and we resume from torch's
    o2 = FlatAdamW([torch.nn.Parameter(p.detach().clone()) for p in ref if True], max_grad_norm=max_norm, **kw)
    o2.load_state_dict(r_opt.state_dict())
    assert o2._t == 8 and torch.allclose(o2.flat_m[:o2.param_groups[0]['params'][0].numel()], r_opt.state[ref[0]]['exp_avg'].reshape(-1))


@pytest.mark.gpu
def test_flat_adamw_real_model_state_dict_cross_resume():
    """The reference builds torch.optim.AdamW(model.parameters()) (maggie/engine/optim.py:117) -- frozen parameters (SpectralNorm weight_u /
    weight_v, dummy_downscale here) included in param_groups. FlatAdamW keeps the same index space: its state_dict loads into torch's AdamW
    over the same model and torch's loads into FlatAdamW (`last_opt.pth` cross-resume, engine/train.py:98-113,335-343)."""
    from maggie_amd.network import build_model
    from maggie_amd.utils import config
    from maggie_amd.optim import FlatAdamW
    dev = _dev()
    torch.manual_seed(0)
    model, _ = build_model(config.model_config('image'))
    model.to(dev)
    twin, _ = build_model(config.model_config('image'))
    twin.load_state_dict(model.state_dict())
    twin.to(dev)
    for a, b in zip(model.parameters(), twin.parameters()):
        b.requires_grad_(a.requires_grad)
    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    ours, ref = FlatAdamW(model.parameters(), **kw), torch.optim.AdamW(twin.parameters(), **kw)
    n_all = len(list(model.parameters()))
    assert len(ours.param_groups[0]['params']) == n_all == len(ref.param_groups[0]['params'])
    assert any(not p.requires_grad for p in model.parameters())
    for _ in range(2):
        for a, b in zip(model.parameters(), twin.parameters()):
            if a.requires_grad:
                g = torch.randn_like(a) * 0.01
                a.grad, b.grad = g.clone(), g.clone()
        ours.step(); ref.step()
    for a, b in zip(model.parameters(), twin.parameters()):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-7)
    sd_o, sd_r = ours.state_dict(), ref.state_dict()
    assert sd_o['param_groups'][0]['params'] == sd_r['param_groups'][0]['params'] == list(range(n_all))
    assert sorted(sd_o['state'].keys()) == sorted(sd_r['state'].keys())          # no state for a parameter that was never stepped
    k = sorted(sd_r['state'].keys())[len(sd_r['state']) // 2]
    assert torch.allclose(sd_o['state'][k]['exp_avg'], sd_r['state'][k]['exp_avg'], rtol=1e-5, atol=1e-9)
    torch.optim.AdamW(twin.parameters(), **kw).load_state_dict(sd_o)              # reference side resumes from ours
    o2 = FlatAdamW(model.parameters(), **kw)
    o2.load_state_dict(sd_r)                                                     # we resume from the reference's
    assert o2._t == 2
    p0 = next(p for p in model.parameters() if p.requires_grad)
    i0 = [i for i, p in enumerate(model.parameters()) if p is p0][0]
    assert torch.allclose(o2.state[p0]['exp_avg'], sd_r['state'][i0]['exp_avg'])


@pytest.mark.gpu
def test_flat_adamw_sync_group_single_rank_rccl():
    """FlatAdamW(sync_group=True): the gradient exchange is one RCCL all-reduce (mean) of the flat buffer. With one rank the mean is the
    identity: same parameters as without the group; a parameter without a local gradient is updated with zeros (DDP semantics)."""
    import os
    import torch.distributed as dist
    from maggie_amd.optim import FlatAdamW
    dev = _dev()
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        torch.manual_seed(1)
        a = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in ((40, 9), (7,), (3, 3))]
        b = [torch.nn.Parameter(p.detach().clone()) for p in a]
        oa = FlatAdamW(a, lr=1e-3, max_grad_norm=0.01, sync_group=True)
        ob = FlatAdamW(b, lr=1e-3, max_grad_norm=0.01)
        for it in range(3):
            for x, y in zip(a, b):
                g = torch.randn_like(x)
                x.grad, y.grad = g.clone(), g.clone()
            oa.step(); ob.step()
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        before = a[1].detach().clone()
        a[1].grad = None
        oa.step()
        assert oa._steps == [4, 4, 4] and not torch.equal(a[1], before)      # decayed / moment-driven update, like a zero gradient under DDP
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('M,with_res,act', [(1000, True, 2), (50000, False, 2), (40000, True, 1), (8, False, 0)])
def test_batch_norm_act_one_call_path_matches_torch(dtype, M, with_res, act):
    """functional.batch_norm_act (mg_bn_train_fwd / mg_bn_train_bwd: statistics, finalize, apply behind one call each way) against
    torch.nn.functional.batch_norm on the CPU -- both the exact two-pass branch (M <= 32768) and the replica-accumulator branch."""
    from maggie_amd import functional as MF
    dev = _dev()
    rs = np.random.RandomState(M)
    C = 32
    q = _q(dtype)
    x = q(torch.from_numpy(rs.normal(0.5, 2.0, (M, C)).astype(np.float32))).requires_grad_(True)
    res = q(torch.from_numpy(rs.normal(size=(M, C)).astype(np.float32))).requires_grad_(True) if with_res else None
    bn_ref = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn_ref.weight.copy_(torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)))
        bn_ref.bias.copy_(torch.from_numpy(rs.normal(size=C).astype(np.float32)))
    bn = torch.nn.BatchNorm1d(C).to(dev)
    bn.load_state_dict(bn_ref.state_dict())
    y_ref = bn_ref(x) if res is None else bn_ref(x) + res
    y_ref = [lambda t: t, F.relu, lambda t: F.leaky_relu(t, 0.2)][act](y_ref)
    gy = q(torch.from_numpy(rs.normal(size=(M, C)).astype(np.float32)))
    y_ref.backward(gy)
    xd = x.detach().to(dev, dtype).requires_grad_(True)
    rd = None if res is None else res.detach().to(dev, dtype).requires_grad_(True)
    MF.ARENA.reset(dev)
    y = MF.batch_norm_act(xd, bn, act, res=rd)
    y.backward(gy.to(dev, dtype))
    tol = _tol(dtype)
    assert (y.detach().float().cpu() - y_ref.detach()).abs().max() <= tol * y_ref.abs().max() + 1e-6
    assert torch.allclose(bn.running_mean.cpu(), bn_ref.running_mean, atol=1e-4) and torch.allclose(bn.running_var.cpu(), bn_ref.running_var, rtol=1e-3, atol=1e-4)
    assert int(bn.num_batches_tracked) == 1
    assert (xd.grad.float().cpu() - x.grad).abs().max() <= max(tol, 2e-4) * x.grad.abs().max() * 2
    if res is not None:
        assert (rd.grad.float().cpu() - res.grad).abs().max() <= tol * res.grad.abs().max() + 1e-6
    for got, ref in ((bn.bias.grad, bn_ref.bias.grad), (bn.weight.grad, bn_ref.weight.grad)):
        assert torch.allclose(got.cpu(), ref, rtol=2e-2, atol=2e-2 * ref.abs().max().item())


@pytest.mark.gpu
def test_bits_select_is_the_fuse_blend():
    """mg_bits_select{,_bwd}: out = a*w + b*(1-w) for the 0/1 weight plane a bit plane encodes (fuse(),
    maggie/network/decoder/resnet_inst_matt_spconv.py:272-290), forward and both gradients, bit-exact; ragged width (W % 64 != 0)."""
    from maggie_amd import functional as MF, kernels as K
    dev = _dev()
    torch.manual_seed(2)
    for P, H, W in ((6, 40, 200), (3, 17, 64), (2, 9, 70)):
        a = torch.randn(P, H, W, device=dev, requires_grad=True)
        b = torch.randn(P, H, W, device=dev, requires_grad=True)
        w8 = (torch.rand(P, H, W, device=dev) < 0.4).to(torch.uint8)
        bits = K.bits_pack(w8, mode=1)
        assert torch.equal(K.bits_unpack_u8(bits, W), w8)
        w = w8.float()
        ref = a * w + b * (1 - w)
        out = MF.bits_select(bits, a, b, W)
        assert torch.equal(out, ref)
        g = torch.randn_like(ref)
        ga, gb = torch.autograd.grad(out, (a, b), g)
        ra, rb = torch.autograd.grad(ref, (a, b), g)
        assert torch.equal(ga, ra) and torch.equal(gb, rb)


@pytest.mark.gpu
@pytest.mark.parametrize('reweight', [True, False])
def test_os8_weight_matches_reference_statements(reweight):
    """mg_os8_weight == the weight_os8 statements of compute_loss (maggie/network/arch/maggie.py:271-281), bit-exact, incl. an empty plane
    and values exactly on the 1/255 and 254/255 thresholds."""
    from maggie_amd import functional as MF
    dev = _dev()
    torch.manual_seed(4)
    alphas = torch.rand(3, 10, 64, 72, device=dev)
    alphas[alphas < 0.3] = 0
    alphas[alphas > 0.8] = 1
    alphas[:, 4] = 0                                              # an empty instance slot
    alphas[0, 0, 0, :4] = torch.tensor([1 / 255.0, 254 / 255.0, 0.0039, 0.9962], device=dev)
    a8 = torch.rand(3, 10, 64, 72, device=dev)
    a8[a8 < 0.2] = 0
    a8[0, 1, 0, :2] = torch.tensor([1 / 255.0, 254 / 255.0], device=dev)
    w = torch.ones_like(a8) * (alphas.sum((2, 3), keepdim=True) > 0)
    if reweight:
        w = (((alphas <= 254.0 / 255.0) & (alphas >= 1.0 / 255.0)) | ((a8 <= 254.0 / 255.0) & (a8 >= 1.0 / 255.0))).type(w.dtype) + w
    assert torch.equal(MF.os8_weight(alphas, a8, reweight), w)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_batched_weight_pipeline_mixes_spectral_norm_and_plain_convs(dtype):
    """functional.spectral_norm_prepare over a list that mixes SpectralNorm wrappers with ordinary conv holders (`plain` descriptors):
    the plain weights come out as the exact per-tensor conversion (KRSC + dgrad twin) and their gradient returns unchanged in OIHW; the
    SpectralNorm entries equal the per-conv kernel path (weight, u/v update, gradient)."""
    from maggie_amd import functional as MF
    from maggie_amd.network.module import SpectralNorm, ConvWeight
    dev = _dev()
    torch.manual_seed(7)
    mk_sn = lambda ci, co, k: SpectralNorm(ConvWeight(ci, co, k, 1, k // 2, 1)).to(dev)          # noqa: E731
    mods = [mk_sn(40, 64, 3), ConvWeight(136, 72, 3, 1, 1, 1).to(dev), mk_sn(64, 32, 1), ConvWeight(512, 256, 1).to(dev), ConvWeight(8, 24, 3, 1, 2, 2).to(dev)]
    twins = [SpectralNorm(ConvWeight(40, 64, 3, 1, 1, 1)).to(dev), SpectralNorm(ConvWeight(64, 32, 1)).to(dev)]
    for a, b in zip((mods[0], mods[2]), twins):
        b.load_state_dict(a.state_dict())
    MF.ARENA.reset(dev)
    cache = {}
    MF.spectral_norm_prepare(mods, dtype, cache)
    outs = []
    for m in mods:
        w = m.__dict__['_prepared']
        outs.append(w)
        if isinstance(m, SpectralNorm):
            continue
        co, ci, k, _ = m.weight.shape
        ref = MF.weight_oihw_to_krsc(m.weight.detach(), dtype)
        assert w.shape == ref.shape and torch.equal(w, ref)
        assert torch.equal(w._mg_wt, ref.permute(2, 1, 0).contiguous())
        assert MF.plain_krsc(m, dtype) is w and MF.plain_krsc(m, dtype) is not w          # handed out once per step
    # SpectralNorm entries: same weight and the same advanced u, v as the per-conv path
    for m, t, w in ((mods[0], twins[0], outs[0]), (mods[2], twins[1], outs[2])):
        ref = t.krsc(dtype)
        tol = 1e-6 if dtype == torch.float32 else 8e-3
        assert (w.float() - ref.float()).abs().max() <= tol * ref.float().abs().max()
        assert torch.allclose(m.module.weight_u, t.module.weight_u, atol=1e-6) and torch.allclose(m.module.weight_v, t.module.weight_v, atol=1e-6)
    # gradients: plain = identity back to OIHW; SpectralNorm = the per-conv backward
    grads = [torch.randn_like(o) for o in outs]
    params = [m.module.weight_bar if isinstance(m, SpectralNorm) else m.weight for m in mods]
    got = torch.autograd.grad(outs, params, grads)
    for m, g_out, g in zip(mods, grads, got):
        if isinstance(m, SpectralNorm):
            continue
        co, ci, k, _ = m.weight.shape
        ref = g_out.float()[:, :, :ci].reshape(co, k, k, ci).permute(0, 3, 1, 2)
        assert g.dtype == torch.float32 and torch.equal(g, ref)
    for idx in (0, 2):
        # u, v have advanced: compare against autograd of the reference formula W / sigma with the advanced vectors held fixed
        wbar = mods[idx].module.weight_bar
        u, v = mods[idx].module.weight_u.detach(), mods[idx].module.weight_v.detach()
        wb = wbar.detach().clone().requires_grad_(True)
        sigma = u.dot(wb.reshape(wb.shape[0], -1).mv(v))
        wn = (wb / sigma).permute(0, 2, 3, 1).reshape(wb.shape[0], -1, wb.shape[1])
        gk = grads[idx].float()[:, :, :wb.shape[1]]
        (ref,) = torch.autograd.grad(wn, wb, gk)
        tol = 2e-5 if dtype == torch.float32 else 2e-2
        assert (got[idx] - ref).abs().max() <= tol * ref.abs().max() + 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_batched_weight_pipeline_transposed_twin_gradient_layout_and_destinations(dtype):
    """Round 5, the ConvTranspose entries of the batched weight pipeline (decoder up-convolutions, maggie/network/decoder/resnet.py:20-45):
    (a) the dgrad twin (Cin_pad, taps, Cout) is emitted by the kernel and equals the permuted weight; (b) a gradient handed back in the twin's
    layout (a permuted view -- what ConvRaw.backward returns for a transposed convolution) gives the bits of the same gradient in the
    (Cout, taps, Cin_pad) layout; (c) with functional.GRAD_DEST the parameter gradients are written into the caller's tensors and those very
    tensors come back."""
    from maggie_amd import functional as MF
    from maggie_amd.network.module import SpectralNorm, ConvWeight
    dev = _dev()
    torch.manual_seed(11)
    mods = [SpectralNorm(ConvWeight(64, 40, 4, 2, 1, 1, transposed=True)).to(dev), ConvWeight(24, 48, 3, 1, 1, 1).to(dev),
            SpectralNorm(ConvWeight(20, 32, 4, 2, 1, 1, transposed=True)).to(dev)]
    state = [{k: v.clone() for k, v in m.state_dict().items()} for m in mods]
    params = [m.module.weight_bar if isinstance(m, SpectralNorm) else m.weight for m in mods]

    def run(twin_layout, dest):
        for m, sd in zip(mods, state):
            m.load_state_dict(sd)
        MF.ARENA.reset(dev)
        MF.spectral_norm_prepare(mods, dtype, {})
        outs = [m.__dict__['_prepared'] for m in mods]
        gen = torch.Generator(device=dev).manual_seed(5)
        grads = [torch.randn(o.shape, device=dev, generator=gen).to(dtype) for o in outs]
        fed = [g.permute(2, 1, 0).contiguous().permute(2, 1, 0) if (twin_layout and isinstance(m, SpectralNorm)) else g for m, g in zip(mods, grads)]
        if dest is not None:
            MF.GRAD_DEST.update({id(p): d for p, d in zip(params, dest)})
        try:
            got = torch.autograd.grad(outs, params, fed)
        finally:
            MF.GRAD_DEST.clear()
        return outs, got

    outs, ref = run(False, None)
    for m, w in zip(mods, outs):
        assert torch.equal(w._mg_wt, w.permute(2, 1, 0).contiguous())              # the twin, transposed entries included
    _, got = run(True, None)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    flat = torch.full((sum(p.numel() for p in params) + 64,), float('nan'), device=dev)
    dest, o = [], 0
    for p_ in params:
        dest.append(flat[o:o + p_.numel()].view(p_.shape))
        o += p_.numel() + 16
    _, got = run(True, dest)
    for a, b, d in zip(got, ref, dest):
        assert a.data_ptr() == d.data_ptr() and torch.equal(a, b) and torch.equal(d, b)
    assert torch.isnan(flat[params[0].numel():params[0].numel() + 16]).all()        # nothing written between the destinations


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_spatial_mean_and_k_way_sum(dtype):
    """mg_spatial_mean (AdaptiveAvgPool2d(1) of ASPP's pooled branch, maggie/network/module/aspp.py:24-27,50-52) forward / backward against torch in
    fp64, ragged channel count; mg_sum_k_t: k same-shaped tensors added in fp32 in list order and rounded once (functional.FanOut for 16-bit
    activations), ragged length."""
    from maggie_amd import functional as MF, kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(13)
    for N, H, W, C in ((4, 16, 16, 512), (2, 7, 9, 40)):
        x = (torch.randn((N, H, W, C), generator=g) + 0.5).to(dev, dtype).requires_grad_(True)
        y = MF.spatial_mean(x)
        ref = x.detach().double().mean((1, 2), keepdim=True)
        assert y.shape == (N, 1, 1, C) and y.dtype == dtype
        tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8
        assert float((y.double() - ref).abs().max()) <= tol * float(ref.abs().max())
        dy = torch.randn((N, 1, 1, C), generator=g).to(dev, dtype)
        (dx,) = torch.autograd.grad(y, x, dy)
        want = (dy.double() / (H * W)).expand(N, H, W, C)
        assert float((dx.double() - want).abs().max()) <= tol * float(want.abs().max())
        # the broadcast back over the map: a view forward; backward = the sum over the map, read in place from a channel slice of a wider gradient
        p_ = torch.randn((N, 1, 1, C), generator=g).to(dev, dtype).requires_grad_(True)
        wide = torch.cat([MF.spatial_broadcast(p_, H, W), x.detach()], -1)
        assert torch.equal(wide[..., :C], p_.detach().expand(N, H, W, C))
        gw = torch.randn((N, H, W, 2 * C), generator=g).to(dev, dtype)
        (gp,) = torch.autograd.grad(wide, p_, gw)
        want = gw[..., :C].double().sum((1, 2), keepdim=True)
        assert float((gp.double() - want).abs().max()) <= tol * float(want.abs().max())
    for n in (4 * 16 * 16 * 512, 1003):
        ts = [torch.randn(n, generator=g).to(dev, dtype) for _ in range(5)]
        got = K.sum_k(ts)
        acc = ts[0].float()
        for t in ts[1:]:
            acc = acc + t.float()
        assert got.dtype == dtype and torch.equal(got, acc.to(dtype))
    # through autograd: a 16-bit tensor with five consumers
    x = torch.randn((2, 8, 8, 64), generator=g).to(dev, dtype).requires_grad_(True)
    fan = MF.Fan(x, 5)
    ws = [float(i + 1) for i in range(5)]
    out = sum((fan() * w_).float().sum() for w_ in ws)
    (gx,) = torch.autograd.grad(out, x)
    assert torch.equal(gx, torch.full_like(x, 15.0))


@pytest.mark.gpu
def test_packed_parameter_slices_collect_their_gradients_in_one_buffer():
    """functional.split_packed (nn.MultiheadAttention.in_proj_weight / in_proj_bias -> q, k, v slices, maggie/network/module/mask_attention.py): the
    token-side linears write the slices' gradients into ONE buffer that comes back as the packed gradient -- same bits as unbind + stack; a slice
    with two consumers goes through functional.Fan and is summed into its slot; a slice consumed outside the slot-aware kernels falls back to
    stacking."""
    from maggie_amd import functional as MF
    dev = _dev()
    torch.manual_seed(3)
    d = 128
    P = torch.randn(3 * d, d, device=dev, requires_grad=True)
    Pb = torch.randn(3 * d, device=dev, requires_grad=True)
    x = torch.randn(4, 10, d, device=dev, requires_grad=True)

    def net(split, fan, torch_tail=False):
        (wq, wk, wv), (bq, bk, bv) = split(P), split(Pb)
        wk2 = fan(wk)
        q, k = MF.token_linear_multi([dict(x=x, W=wq, b=bq), dict(x=x, W=MF.take(wk2), b=bk)])
        k2 = MF.token_linear(q, MF.take(wk2), None, wt=True)
        v = (x @ wv.t() + bv) if torch_tail else MF.token_linear(x, wv, bv)
        return (q * 0.5 + k * 0.25 + k2 * 0.125 + v).sum()

    plain = lambda t: t.view(3, d, *t.shape[1:]).unbind(0)            # noqa: E731
    ref = torch.autograd.grad(net(plain, lambda t: t), (P, Pb, x))
    seen = []
    orig = MF.SplitPacked.backward

    def spy(ctx, *gs):
        out = orig(ctx, *gs)
        seen.append(ctx.holder.k == 3 and out[0].data_ptr() == gs[0].data_ptr())      # the buffer itself came back (slice 0 starts it)
        return out
    MF.SplitPacked.backward = staticmethod(spy)
    try:
        got = torch.autograd.grad(net(lambda t: MF.split_packed(t, 3), lambda t: MF.Fan(t, 2)), (P, Pb, x))
        assert seen == [True, True]
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
        del seen[:]
        got = torch.autograd.grad(net(lambda t: MF.split_packed(t, 3), lambda t: MF.Fan(t, 2), torch_tail=True), (P, Pb, x))
        assert seen == [False, False]                                  # the v slice's gradient came from torch: stacked
        ref2 = torch.autograd.grad(net(plain, lambda t: t, torch_tail=True), (P, Pb, x))
        for a, b in zip(got, ref2):
            assert torch.equal(a, b)
    finally:
        MF.SplitPacked.backward = orig


@pytest.mark.gpu
@pytest.mark.parametrize('T', [2, 3, 5])
def test_bidirectional_fusion_kernel_matches_torch_restatement(T):
    """mg_bifuse_fwd / _bwd against the statement-by-statement torch form of bidirectional_fusion's blend
    (maggie/network/decoder/resnet_inst_matt_spconv_temp.py:53-78) for given difference logits: fused alphas, the zero-padded difference
    stacks, their sigmoids, and the gradients w.r.t. predictions and logits."""
    from maggie_amd import functional as MF
    dev = _dev()
    rs = np.random.RandomState(T)
    B, NI, H, W = 2, 3, 8, 24
    preds = torch.from_numpy(rs.uniform(size=(B, T, NI, H, W)).astype(np.float32)).requires_grad_(True)
    diffs = torch.from_numpy(rs.normal(size=(2 * (T - 1), B, 1, H, W)).astype(np.float32)).requires_grad_(True)
    fw, bw = [preds[:, 0]], [preds[:, T - 1]]
    fd, bd = [], []
    for i in range(1, T):
        d = diffs[i - 1]; fd.append(d)
        fw.append(fw[-1] * (1 - d.sigmoid()) + preds[:, i] * d.sigmoid())
    for j, i in enumerate(range(T - 1, 0, -1)):
        d = diffs[T - 1 + j]; bd.append(d)
        bw.append(bw[-1] * (1 - d.sigmoid()) + preds[:, i - 1] * d.sigmoid())
    bw, bd = bw[::-1], bd[::-1]
    fused_ref = torch.stack([fw[i] if i == 0 else bw[i] if i == T - 1 else (fw[i] + bw[i]) / 2 for i in range(T)], 1)
    fd_ref = torch.stack([torch.zeros_like(fd[0])] + fd, 1)
    bd_ref = torch.stack(bd + [torch.zeros_like(bd[-1])], 1)
    wgt = torch.from_numpy(rs.normal(size=fused_ref.shape).astype(np.float32))
    (fused_ref * wgt).sum().backward()
    p2 = preds.detach().to(dev).requires_grad_(True)
    d2 = diffs.detach().to(dev).requires_grad_(True)
    fused, aux = MF.BiFuse.apply(p2, d2)
    (fused * wgt.to(dev)).sum().backward()
    assert torch.allclose(fused.cpu(), fused_ref.detach(), atol=1e-6)
    assert torch.allclose(aux[0].cpu(), fd_ref.detach(), atol=0) and torch.allclose(aux[1].cpu(), bd_ref.detach(), atol=0)
    assert torch.allclose(aux[2].cpu(), fd_ref.detach().sigmoid(), atol=1e-6) and torch.allclose(aux[3].cpu(), bd_ref.detach().sigmoid(), atol=1e-6)
    assert torch.allclose(p2.grad.cpu(), preds.grad, atol=1e-5), float((p2.grad.cpu() - preds.grad).abs().max())
    dref = diffs.grad if diffs.grad is not None else torch.zeros_like(diffs)       # T = 2: the fused frames are the predictions themselves
    assert torch.allclose(d2.grad.cpu(), dref, atol=1e-5), float((d2.grad.cpu() - dref).abs().max())


@pytest.mark.gpu
def test_dtssd_and_bce_loss_kernels_match_torch():
    """mg_dtssd_* (loss_dtSSD, maggie/network/loss.py:7-16, pinned value in dense_pinned.npz) and mg_bce_logits_* against torch: full
    tensors, frame slices used in place (loss_temporal_sparsity's [:, 1:] / [:, :-1]), sigmoid-of-logits form, absent mask; values + gradients."""
    from maggie_amd import functional as MF
    from helpers import load_golden
    dev = _dev()
    rs = np.random.RandomState(9)
    # the pinned value of the reference's own loss_dtSSD (same draws as tests/golden/make_golden.py:dense_fixture)
    _ = rs.normal(size=(1, 3, 128, 8, 8)); _ = rs.normal(size=(1, 3, 64, 8, 8)); _ = rs.uniform(size=(1, 3, 2, 64, 64))
    a = torch.from_numpy(rs.uniform(size=(2, 3, 32, 32)).astype(np.float32))
    g = torch.from_numpy(rs.uniform(size=(2, 3, 32, 32)).astype(np.float32))
    wgt = torch.from_numpy((rs.uniform(size=(2, 3, 32, 32)) > 0.5).astype(np.float32))
    r5 = lambda t: t.reshape(1, 2, 3, 32, 32).to(dev)       # noqa: E731
    assert abs(float(MF.dtssd_loss(r5(a), r5(g), r5(wgt))) - float(load_golden('dense_pinned.npz')['loss/dtssd'])) < 1e-6

    def ref_dtssd(pred, gt, mask):
        dadt, dgdt = pred[:, 1:] - pred[:, :-1], gt[:, 1:] - gt[:, :-1]
        return torch.sum((dadt - dgdt) ** 2 * mask[:, 1:]) / torch.sum(mask[:, 1:] + 1e-6)

    B, T, NI, H, W = 2, 4, 3, 8, 16
    x = torch.from_numpy(rs.normal(size=(B, T, NI, H, W)).astype(np.float32))
    y = torch.from_numpy(rs.uniform(size=(B, T, NI, H, W)).astype(np.float32))
    m = torch.from_numpy((rs.uniform(size=(B, T, NI, H, W)) > 0.3).astype(np.float32))
    for sl, sig, use_m in ((slice(None), False, True), (slice(1, None), True, False), (slice(None, -1), True, False), (slice(1, None), False, True)):
        xr = x.clone().requires_grad_(True)
        pr = xr[:, sl].sigmoid() if sig else xr[:, sl]
        lr = ref_dtssd(pr, y[:, 1:] if sl != slice(None) else y, (m[:, 1:] if sl != slice(None) else m) if use_m else torch.ones_like(pr))
        lr.backward()
        xg = x.clone().to(dev).requires_grad_(True)
        yy = (y[:, 1:] if sl != slice(None) else y).to(dev)
        mm = ((m[:, 1:] if sl != slice(None) else m).to(dev)) if use_m else None
        lg = MF.dtssd_loss(xg[:, sl], yy, mm, sig=sig)
        lg.backward()
        assert abs(float(lg) - float(lr)) <= 1e-5 * max(1.0, abs(float(lr))), (sl, sig)
        assert torch.allclose(xg.grad.cpu(), xr.grad, atol=1e-6, rtol=1e-4), (sl, sig, float((xg.grad.cpu() - xr.grad).abs().max()))
        # binary cross entropy with logits on the same slices
        xr2 = x.clone().requires_grad_(True)
        tgt = ((y[:, 1:] if sl != slice(None) else y) > 0.5).float()
        lb = F.binary_cross_entropy_with_logits(xr2[:, sl], tgt, reduction='mean')
        lb.backward()
        xg2 = x.clone().to(dev).requires_grad_(True)
        lh = MF.bce_logits_mean(xg2[:, sl], tgt.to(dev))
        lh.backward()
        assert abs(float(lh) - float(lb)) <= 1e-5 and torch.allclose(xg2.grad.cpu(), xr2.grad, atol=1e-7, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('R,Kd,N,xadd,bias,res,relu,ln', [
    (40, 128, 128, True, True, True, False, True),      # projection + residual + LayerNorm (attention out-projection, FFN second layer)
    (40, 128, 128, False, True, False, True, False),    # linear + ReLU (FFN first layer)
    (40, 128, 11, False, False, False, False, False),   # score-bias table (N not a multiple of 4)
    (11, 128, 128, False, True, False, False, False),   # ID-embedding rows through a projection
    (10, 128, 64, False, True, False, False, True),     # final MLP + decoder_norm (batch 1), LayerNorm without residual
    (80, 128, 1, True, False, False, False, False),     # batch 8, one output column
])
def test_token_linear_matches_torch(R, Kd, N, xadd, bias, res, relu, ln):
    """mg_token_linear_fwd / _bwd: y = LN(res + act((x + xadd) W^T + b)) and every gradient against torch autograd."""
    from maggie_amd import functional as MF
    dev = _dev()
    g = torch.Generator().manual_seed(R * 1000 + N)
    mk = lambda *s: torch.randn(*s, generator=g)                      # noqa: E731
    x, W = mk(R, Kd), mk(N, Kd) / Kd ** 0.5
    xa = mk(R, Kd) if xadd else None
    b = mk(N) if bias else None
    r = mk(R, N) if res else None
    norm = torch.nn.LayerNorm(N) if ln else None
    if ln:
        with torch.no_grad():
            norm.weight.copy_(mk(N)); norm.bias.copy_(mk(N))
    leaves = [t for t in (x, xa, W, b, r) if t is not None]
    for t in leaves:
        t.requires_grad_(True)
    h = F.linear(x + xa if xadd else x, W, b)
    if relu:
        h = F.relu(h)
    if res:
        h = h + r
    y_ref = norm(h) if ln else h
    wgt = mk(R, N)
    (y_ref * wgt).sum().backward()
    dl = [None if t is None else t.detach().clone().to(dev).requires_grad_(True) for t in (x, xa, W, b, r)]
    nd = None
    if ln:
        nd = torch.nn.LayerNorm(N).to(dev)
        nd.load_state_dict(norm.state_dict())
    y = MF.token_linear(dl[0], dl[2], dl[3], xadd=dl[1], res=dl[4], relu=relu, ln=nd)
    (y * wgt.to(dev)).sum().backward()
    tol = dict(atol=2e-5, rtol=2e-5)
    assert torch.allclose(y.detach().cpu(), y_ref.detach(), **tol), float((y.detach().cpu() - y_ref.detach()).abs().max())
    for name, a, c in zip(('x', 'xadd', 'W', 'b', 'res'), dl, (x, xa, W, b, r)):
        if c is not None:
            assert torch.allclose(a.grad.cpu(), c.grad, atol=5e-5, rtol=5e-5), (name, float((a.grad.cpu() - c.grad).abs().max()))
    if ln:
        assert torch.allclose(nd.weight.grad.cpu(), norm.weight.grad, atol=5e-5, rtol=5e-5) and torch.allclose(nd.bias.grad.cpu(), norm.bias.grad, atol=5e-5, rtol=5e-5)


@pytest.mark.gpu
def test_token_self_attention_matches_torch():
    """mg_token_sa_fwd / _bwd: softmax(q k^T / sqrt(d), key padding mask) v for 10 tokens per batch element, against torch autograd."""
    from maggie_amd import functional as MF
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    B, T, D = 4, 10, 128
    q, k, v = (torch.randn(B, T, D, generator=g).requires_grad_(True) for _ in range(3))
    pad = torch.zeros(B, T, dtype=torch.bool)
    pad[:, 2:] = True
    pad[1, :5] = False
    s = torch.matmul(q, k.transpose(1, 2)) / D ** 0.5
    s = s.masked_fill(pad[:, None, :], float('-inf'))
    out_ref = torch.matmul(torch.softmax(s, -1), v)
    wgt = torch.randn(B, T, D, generator=g)
    (out_ref * wgt).sum().backward()
    qd, kd, vd = (t.detach().clone().to(dev).requires_grad_(True) for t in (q, k, v))
    out = MF.token_self_attention(qd, kd, vd, pad.to(dev))
    (out * wgt.to(dev)).sum().backward()
    assert torch.allclose(out.detach().cpu(), out_ref.detach(), atol=2e-5, rtol=2e-5)
    for a, c in ((qd, q), (kd, k), (vd, v)):
        assert torch.allclose(a.grad.cpu(), c.grad, atol=5e-5, rtol=5e-5), float((a.grad.cpu() - c.grad).abs().max())


@pytest.mark.gpu
def test_atten_guidance_loss_and_scalar_lincomb_match_torch():
    """compute_atten_loss (instance_matte_decoder.py) and the loss sums of arch/maggie.py:283-300 as fused HIP launches: value and gradients
    against the torch expressions they replace."""
    from maggie_amd import functional as MF
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    b, n_i, L, n_f = 3, 10, 777, 1
    gm = (torch.rand(b, n_i, L, generator=g) > 0.7).float()
    gm[1, 4] = 0                                                     # an instance slot without guidance: contributes 0 - 0
    att = torch.rand(b, n_i, L, generator=g).softmax(-1)
    a_ref = att.clone().requires_grad_(True)
    ref = ((gm.sum(2) != 0).float() - (gm * a_ref).sum(2)).sum() / (n_f * b)
    (ref * 1.7).backward()
    a = att.to(dev).requires_grad_(True)
    out = MF.atten_guidance_loss(gm.to(dev), a, 1.0 / (n_f * b))
    (out * 1.7).backward()
    assert abs(float(out) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert (a.grad.cpu() - a_ref.grad).abs().max().item() <= 1e-6

    ts = [torch.randn((), generator=g) for _ in range(9)]
    cs = [2.0, 1.0, 1.0, 0.5, 0.25, 0.25, 3.0, 1.5, 1.5]
    tr = [t.clone().requires_grad_(True) for t in ts]
    ref = sum(c * t for c, t in zip(cs, tr)) + 0.125 * 2.0
    ref.backward()
    td = [t.to(dev).requires_grad_(True) for t in ts]
    out = MF.scalar_lincomb(td + [0.125], cs + [2.0])               # a python number among the terms is folded on the host
    out.backward()
    assert abs(float(out) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
    for t, r, c in zip(td, tr, cs):
        assert abs(float(t.grad) - float(r.grad)) <= 1e-7 and abs(float(t.grad) - c) <= 1e-7


@pytest.mark.gpu
def test_temporal_crop_matches_oracle_restatement():
    """Eval bounding-box crop of the video decoder (resnet_inst_matt_spconv_temp.py:115-142): smoothing with the reference's kernel quirk, crop,
    bilinear resize, threshold 0.1, box +-30 px, applied to the alpha planes and to the detail bit planes -- against the oracle's restatement
    (oracle/refmodel.py:gaussian_smoothing + the box loop) on planes with separated blobs, an empty plane and a plane touching the border."""
    from maggie_amd import functional as MF, kernels as K
    from oracle import refmodel as rm
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    N, n_i, H, W = 2, 3, 96, 160
    a = torch.zeros(N, n_i, H, W)
    a[0, 0, 20:40, 30:70] = torch.rand(20, 40, generator=g) * 0.8 + 0.2
    a[0, 1, 0:12, 100:160] = 0.9                                     # touches the top / right border
    a[1, 0, 60:90, 5:25] = 0.5
    a[1, 0, 10:14, 120:126] = 0.05                                   # faint: stays below the 0.1 threshold after smoothing
    a[1, 2, 40:44, 80:84] = 1.0                                      # tiny blob
    unk = (torch.rand(N, n_i, H, W, generator=g) > 0.5)
    sm = rm.gaussian_smoothing(a, 3)
    ref_a, ref_u = a.clone(), unk.clone()
    for i in range(N):
        for j in range(n_i):
            ys, xs = torch.nonzero(sm[i, j] > 0.1, as_tuple=True)
            if len(ys) == 0:
                continue
            y0, y1 = max(0, int(ys.min()) - 30), min(int(ys.max()) + 30, H)
            x0, x1 = max(0, int(xs.min()) - 30), min(int(xs.max()) + 30, W)
            tm = torch.zeros(H, W, dtype=torch.bool)
            tm[y0:y1, x0:x1] = True
            ref_a[i, j] = ref_a[i, j] * tm
            ref_u[i, j] = ref_u[i, j] & tm
    ad = a.to(dev).contiguous()
    bits = K.bits_pack(unk.reshape(N * n_i, H, W).to(torch.uint8).to(dev), mode=1)
    MF.temporal_crop_(ad, bits)
    assert torch.equal(ad.cpu(), ref_a)
    got_u = K.bits_unpack_u8(bits, W, (N * n_i, H, W)).cpu().bool().reshape(N, n_i, H, W)
    assert torch.equal(got_u, ref_u)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_rows_dropout_mask_statistics_and_backward_reuse(dtype):
    """inst_spec_layer's dropout (mask_attention.py:170-182, p = 0.1; resnet_inst_matt_spconv.py:226-232) as the counter-based row kernel:
    keep rate within 3 sigma of 1 - p, kept values scaled by exactly 1 / (1 - p), the backward call with the same (state, salt) re-creates
    the forward's mask (checked against torch given the extracted mask), another salt / step gives another mask, p = 0 is the identity, and
    only the first `rows` rows of a capacity-sized buffer are touched."""
    from maggie_amd import kernels as K
    from maggie_amd.sparse_head import DeviceRng
    dev = _dev()
    torch.manual_seed(11)
    M, C, p = 40000, 32, 0.1
    rng = DeviceRng(dev)
    state = rng.snapshot()
    x = (torch.rand(M, C, device=dev) + 0.5).to(dtype)                 # strictly positive: a zero in the output IS a dropped element
    live = torch.tensor([M - 1000], dtype=torch.int32, device=dev)
    y = K.rows_dropout(x, p, state, 1, rows=live)
    yl, xl = y[:M - 1000].float(), x[:M - 1000].float()
    keep = yl != 0
    n = keep.numel()
    rate = keep.float().mean().item()
    sigma = (p * (1 - p) / n) ** 0.5
    assert abs(rate - (1 - p)) <= 3 * sigma, (rate, sigma)
    # per-channel and per-row keep rates are unbiased too (a hash that correlates with the row or channel index would fail here)
    assert (keep.float().mean(0) - (1 - p)).abs().max().item() <= 5 * (p * (1 - p) / keep.shape[0]) ** 0.5
    assert abs(keep.float().mean(1).std().item() - (p * (1 - p) / C) ** 0.5) <= 0.1 * (p * (1 - p) / C) ** 0.5
    tol = 0 if dtype == torch.float32 else 2 ** -8
    ref = xl * (1.0 / (1.0 - p))
    assert ((yl - ref * keep).abs() <= tol * ref.abs() + (1e-6 if dtype == torch.float32 else 0)).all()
    # backward: same (state, salt) on the upstream gradient = torch's dropout backward with the forward's mask
    g = torch.randn(M, C, device=dev).to(dtype)
    gx = K.rows_dropout(g, p, state, 1, rows=live)
    gref = g[:M - 1000].float() * keep * (1.0 / (1.0 - p))
    assert ((gx[:M - 1000].float() - gref).abs() <= tol * gref.abs() + 1e-6).all()
    # a second dropout site of the same forward (salt 2) and the next step draw different masks
    k2 = K.rows_dropout(x, p, state, 2, rows=live)[:M - 1000] != 0
    k3 = K.rows_dropout(x, p, rng.snapshot(), 1, rows=live)[:M - 1000] != 0
    for other in (k2, k3):
        agree = (other == keep).float().mean().item()                 # independent masks agree on (1-p)^2 + p^2 = 0.82 of the elements
        assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 0.01, agree
    assert torch.equal(K.rows_dropout(x, 0.0, state, 1, rows=live)[:M - 1000], x[:M - 1000])


@pytest.mark.gpu
@pytest.mark.parametrize('R,Kd,N,xadd', [(40, 128, 128, True), (7, 128, 64, False), (130, 64, 128, False)])
def test_token_linear_with_untransposed_weight_matches_torch(R, Kd, N, xadd):
    """mg_token_linear_fwd_ex / _bwd_ex with wt = 1: y = (x + xadd) W for W given as (K, N) -- `q @ wk` of the cross-attention folding
    (module/mask_attention.py) without the transposed copy of the projection weight every step; dW comes back as (K, N)."""
    from maggie_amd import functional as MF
    dev = _dev()
    g = torch.Generator().manual_seed(R + N)
    x, W = torch.randn(R, Kd, generator=g).requires_grad_(True), (torch.randn(Kd, N, generator=g) / Kd ** 0.5).requires_grad_(True)
    xa = torch.randn(R, Kd, generator=g).requires_grad_(True) if xadd else None
    y_ref = (x + xa if xadd else x) @ W
    wgt = torch.randn(R, N, generator=g)
    (y_ref * wgt).sum().backward()
    xd, Wd = x.detach().clone().to(dev).requires_grad_(True), W.detach().clone().to(dev).requires_grad_(True)
    xad = xa.detach().clone().to(dev).requires_grad_(True) if xadd else None
    y = MF.token_linear(xd, Wd, xadd=xad, wt=True)
    (y * wgt.to(dev)).sum().backward()
    assert torch.allclose(y.detach().cpu(), y_ref.detach(), atol=2e-5, rtol=2e-5)
    assert Wd.grad.shape == W.grad.shape
    for a, c in ((xd, x), (Wd, W)) + (((xad, xa),) if xadd else ()):
        assert torch.allclose(a.grad.cpu(), c.grad, atol=5e-5, rtol=5e-5), float((a.grad.cpu() - c.grad).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,L,Q,C', [(4, 4096, 10, 64), (1, 777, 10, 32), (2, 300, 16, 64)])
def test_token_einsum_matches_torch(dtype, B, L, Q, C):
    """mg_token_einsum_fwd / _bwd: einsum('bqc,blc->blq') of instance_matte_decoder.py:296-299 (pixel logits against the instance tokens of the
    pixel's batch element), padded to 16 output columns, against torch.einsum on the CPU (inputs rounded to the compute dtype first)."""
    from maggie_amd import functional as MF
    dev = _dev()
    q_ = _q(dtype)
    g = torch.Generator().manual_seed(B * L)
    feat = q_(torch.randn(B, L, C, generator=g)).requires_grad_(True)
    tok = q_(torch.randn(B, Q, C, generator=g)).requires_grad_(True)
    ref = torch.einsum('bqc,blc->blq', tok, feat)
    wgt = q_(torch.randn(B, L, Q, generator=g))
    (ref * wgt).sum().backward()
    fd = feat.detach().to(dev, dtype).requires_grad_(True)
    td = tok.detach().to(dev).requires_grad_(True)
    out = MF.token_einsum(fd, td)
    assert out.shape == (B, L, 16) and out.dtype == dtype
    assert float(out[..., Q:].float().abs().max()) == 0.0 if Q < 16 else True
    w16 = torch.zeros(B, L, 16)
    w16[..., :Q] = wgt
    (out.float() * w16.to(dev)).sum().backward()
    tol = _tol(dtype)
    assert (out[..., :Q].detach().float().cpu() - ref.detach()).abs().max() <= tol * ref.abs().max()
    assert (fd.grad.float().cpu() - feat.grad).abs().max() <= tol * feat.grad.abs().max()
    assert (td.grad.float().cpu() - tok.grad).abs().max() <= max(tol, 1e-4) * tok.grad.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize('cap', [0, 1, 777, 5000, 10 ** 9])
def test_bits_truncate_keeps_the_first_cap_sites(cap):
    """mg_bits_truncate (bounded sparse-head capacity): the first `cap` active sites in torch.nonzero order survive, the count word is clamped
    and the sticky overflow flag says whether anything was dropped -- against numpy."""
    from maggie_amd import kernels as K
    dev = _dev()
    rs = np.random.RandomState(5)
    P, H, W = 3, 37, 150
    dense = (rs.rand(P, H, W) < 0.3)
    a = torch.from_numpy(dense.astype(np.float32)).to(dev)
    bits = K.bits_pack(a, mode=1)
    rowoff, wordoff = K.bits_rank(bits, W)
    count = rowoff[-1:]
    total = int(dense.sum())
    assert int(count) == total
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    K.bits_truncate_(bits, wordoff, W, cap, count, ovf)
    got = K.bits_unpack_u8(bits, W).cpu().numpy().astype(bool)
    want = np.zeros_like(dense)
    idx = np.argwhere(dense)[:cap]
    want[idx[:, 0], idx[:, 1], idx[:, 2]] = True
    assert np.array_equal(got, want)
    assert int(count) == min(total, cap) and int(ovf) == int(total > cap)


@pytest.mark.gpu
def test_token_linear_multi_equals_separate_layers():
    """mg_token_linear_multi_fwd / _bwd: several INDEPENDENT token layers per launch (the q / k / v projections of the token self-attention; a
    projection next to the batch-independent ID-table product; a transposed-weight product next to a narrow table product) must give exactly what
    the one-layer-per-launch path gives -- outputs and every gradient, including a tensor shared by several layers (autograd sums)."""
    from maggie_amd import functional as MF
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)                      # noqa: E731
    ln = torch.nn.LayerNorm(128).to(dev)
    with torch.no_grad():
        ln.weight.copy_(mk(128)); ln.bias.copy_(mk(128))

    def leaves():
        torch.manual_seed(1)
        t = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        return t

    base = dict(tgt=mk(4, 10, 128), pos=mk(4, 10, 128), wq=mk(128, 128) / 11, wk=mk(128, 128) / 11, wv=mk(128, 128) / 11, bq=mk(128), bk=mk(128), bv=mk(128),
                table=mk(11, 128), kp=mk(11, 128) / 11, res=mk(4, 10, 128), w1=mk(64, 128) / 11, b1=mk(64))
    wts = [mk(4, 10, 128), mk(4, 10, 128), mk(4, 10, 128), mk(11, 128), mk(11, 128), mk(4, 10, 64), mk(4, 10, 128), mk(4, 10, 11), mk(4, 10, 128), mk(4, 10, 64)]

    def run(multi):
        t = leaves()
        prev, MF.TOKEN_MULTI = MF.TOKEN_MULTI, multi
        try:
            a = MF.token_linear_multi([dict(x=t['tgt'], W=t['wq'], b=t['bq'], xadd=t['pos']), dict(x=t['tgt'], W=t['wk'], b=t['bk'], xadd=t['pos']),
                                       dict(x=t['tgt'], W=t['wv'], b=t['bv']), dict(x=t['table'], W=t['wk'], b=t['bk']),
                                       dict(x=t['table'], W=t['wq'], b=t['bq']), dict(x=t['tgt'], W=t['w1'], b=t['b1'])])      # six layers: one launch
            b = MF.token_linear_multi([dict(x=a[0], W=t['wk'], wt=True), dict(x=a[0], W=t['kp']),
                                       dict(x=a[2], W=t['wq'], res=t['res'], ln=ln), dict(x=a[1], W=t['w1'], b=t['b1'], relu=True)])
        finally:
            MF.TOKEN_MULTI = prev
        outs = list(a) + list(b)
        ln.zero_grad()
        sum((o * w).sum() for o, w in zip(outs, wts)).backward()
        return [o.detach() for o in outs], {k: v.grad.clone() for k, v in t.items()}, (ln.weight.grad.clone(), ln.bias.grad.clone())

    o1, g1, l1 = run(True)
    o0, g0, l0 = run(False)
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    for k in g0:
        assert torch.allclose(g1[k], g0[k], atol=2e-5, rtol=1e-5), (k, float((g1[k] - g0[k]).abs().max()))      # (a tensor shared by several layers: autograd's sum order)
    assert torch.allclose(l1[0], l0[0], atol=2e-5, rtol=1e-5) and torch.allclose(l1[1], l0[1], atol=2e-5, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('with_valid', [False, True])
def test_matting_losses_of_three_scales_in_one_pipeline(with_valid):
    """mg_matting_losses_fwd / _bwd: the loss pipelines of alpha_os1 / os4 / os8 (arch/maggie.py:283-300: same shape, same target, their own weights)
    as ONE batched set of launches must equal three single-scale pipelines (values and gradients), also with the per-plane validity of
    `pred * valid_masks` (:112-118) folded in -- which itself must equal multiplying the predictions first."""
    from maggie_amd import functional as MF
    dev = _dev()
    rs = np.random.RandomState(21)
    P, H, W = (2, 5), 48, 72
    mk = lambda: torch.from_numpy(rs.uniform(size=P + (H, W)).astype(np.float32)).to(dev)       # noqa: E731
    preds = [mk() for _ in range(3)]
    tgt = mk()
    wts = [(mk() > t_).float() * s_ for t_, s_ in ((0.4, 1.0), (0.7, 1.0), (0.2, 2.0))]
    wts[0][:, 1] = 0
    wts[1][:, 1] = 0
    wts[2][:, 3] = 0
    pvalid = torch.tensor([1, 1, 0, 1, 1, 1, 1, 1, 0, 1], dtype=torch.int32, device=dev) if with_valid else None
    coefs = torch.tensor([[1.3, 0.7, 2.1], [0.4, 1.9, 0.6], [1.0, 1.0, 1.0]], device=dev)

    def run(multi):
        prev, MF.LOSS_MULTI = MF.LOSS_MULTI, multi
        try:
            ps = [p_.clone().requires_grad_(True) for p_ in preds]
            out = torch.stack([torch.stack(r) for r in MF.matting_losses_multi(ps, tgt, wts, pvalid)])
            (out * coefs).sum().backward()
            return out.detach(), [p_.grad.clone() for p_ in ps]
        finally:
            MF.LOSS_MULTI = prev

    o1, g1 = run(True)
    o0, g0 = run(False)
    assert torch.allclose(o1, o0, rtol=2e-6, atol=1e-7), (o1, o0)
    for a, b in zip(g1, g0):
        assert (a - b).abs().max().item() <= 1e-6 * max(b.abs().max().item(), 1e-12) + 1e-9
    if with_valid:
        # the folded validity == multiplying the predictions by the 0 / 1 plane factor before the loss
        vm = pvalid.float().view(2, 5, 1, 1)
        ps = [(p_ * vm).requires_grad_(True) for p_ in preds]
        ref = torch.stack([torch.stack(r) for r in MF.matting_losses_multi(ps, tgt, wts, None)])
        assert torch.allclose(o1, ref, rtol=2e-6, atol=1e-7)
        for g_ in g1:
            assert float(g_.view(10, H, W)[2].abs().max()) == 0.0 and float(g_.view(10, H, W)[8].abs().max()) == 0.0


def test_fan_out_adds_the_consumer_gradients_in_one_ordered_launch():
    """functional.Fan / FanOut: k aliases of a tensor, one per consumer; the k gradients are added by mg_sum_k as ((g0 + g1) + g2) + ... in alias
    order. Against plain autograd (pairwise adds) and against the same order written out; 20 consumers (> one table of 16); an unused alias."""
    from maggie_amd import functional as MF, kernels as K
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn(4, 10, 128, generator=g).to(dev) for _ in range(20)]
    for k in (3, 5, 20):
        t = torch.randn(4, 10, 128, generator=g).to(dev).requires_grad_(True)
        f = MF.Fan(t, k + 1)                                       # one alias stays unused
        assert f.outs is not None
        used = [f() for _ in range(k)]
        sum((a * w).sum() for a, w in zip(used, ws)).backward()
        got = t.grad.clone()
        t.grad = None
        sum((t * w).sum() for w in ws[:k]).backward()
        assert torch.allclose(got, t.grad, rtol=1e-6, atol=1e-6)
    a, b, c = ws[:3]
    assert torch.equal(K.sum_k([a, b, c]), (a + b) + c)
    plain = torch.randn(3, 4, device=dev)                          # no gradient wanted: the tensor itself is handed out
    assert MF.Fan(plain, 4)() is plain and MF.take(plain) is plain and MF.take(None) is None


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('act', [0, 1, 2])
@pytest.mark.parametrize('M,C', [(2048, 64), (5000, 32), (40000, 128)])
def test_batchnorm_backward_with_the_activation_mask_reformed_from_the_raw_input(M, C, act, dtype):
    """Operand-path BatchNorm (round 5): the normalised activation z = act(x * scale + shift) is never stored, so mg_bn_train_bwd gets y = NULL and
    re-forms the sign of z from the raw input. Same bits as the stored form (dx and both sums), for every activation, with and without the
    ReLU-before-BatchNorm input mask."""
    from maggie_amd import kernels as K
    dev = _dev()
    rs = np.random.RandomState(M + C + act)
    x = torch.from_numpy(rs.normal(0.2, 1.3, (M, C)).astype(np.float32)).to(dev, dtype)
    dz = torch.from_numpy(rs.normal(size=(M, C)).astype(np.float32)).to(dev, dtype)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32) * rs.choice([-1.0, 1.0], C).astype(np.float32)).to(dev)
    beta = torch.from_numpy(rs.normal(size=C).astype(np.float32)).to(dev)
    stats = K.colstats(x)
    sc, sh, mean, invstd = K.bn_finalize(stats, M, gamma, beta, None, None, 0.1, 1e-5)
    pack = sc._base.view(-1)
    z = K.affine_act(x, sc, sh, act=act, slope=0.2)
    for mask_x_pos in (False, True):
        dx_a, _, s_a = K.bn_train_bwd(dz, z, x, pack, act, 0.2, False, mask_x_pos)
        dx_b, _, s_b = K.bn_train_bwd(dz, None, x, pack, act, 0.2, False, mask_x_pos)
        assert torch.equal(dx_a, dx_b) and torch.equal(s_a, s_b), (int((dx_a != dx_b).sum()), float((s_a - s_b).abs().max()))


def test_copy_k_many_buffers_in_one_launch():
    """mg_copy_k (round 5): up to 16 device copies per launch -- odd byte counts, unaligned addresses, empty and aliased jobs, more than 16 jobs,
    mixed dtypes; pairs that are not plain byte copies fall back to torch's copy."""
    from maggie_amd import kernels as K
    dev = _dev()
    g = torch.Generator(device='cpu').manual_seed(3)
    sizes = [1, 7, 4096, 4097, 65536 + 3, 10 * 512 * 512, 0, 33, 16, 15, 1 << 20, 5, 1000003, 64, 2, 9, 4095, 8192 + 1, 12345]
    srcs = [torch.randn(max(n, 1), generator=g)[:n].to(dev) for n in sizes]
    srcs[3] = torch.randn(4098, generator=g).to(dev)[1:]                       # a 4-byte-aligned (not 16-byte-aligned) source
    srcs.append((torch.randn(777, generator=g) * 100).to(dev).to(torch.bfloat16))
    srcs.append(torch.randint(0, 255, (4, 10, 64, 64), generator=g).to(dev).to(torch.uint8))
    dsts = [torch.full_like(s, 7) for s in srcs]
    alias = torch.arange(100, device=dev, dtype=torch.float32)
    srcs.append(alias); dsts.append(alias)                                     # same address: skipped
    srcs.append(torch.randn(6, 5, generator=g).to(dev).t()); dsts.append(torch.zeros(5, 6, device=dev))        # non-contiguous: torch's copy
    srcs.append(torch.randn(32, generator=g).to(dev)); dsts.append(torch.zeros(32, device=dev, dtype=torch.bfloat16))   # dtype change: torch's copy
    K.copy_k(dsts, srcs)
    torch.cuda.synchronize()
    for d, s_ in zip(dsts, srcs):
        assert torch.equal(d.float(), s_.to(d.dtype).float()), (d.shape, d.dtype)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('act,mask_x_pos,lazy', [(1, False, False), (2, False, True), (0, True, False), (1, False, True)])
@pytest.mark.parametrize('M,C', [(2048, 64), (16384, 128), (65536, 64), (4096, 256), (16001, 128), (5000, 32), (1500, 512)])
def test_batchnorm_backward_single_launch_form(M, C, act, mask_x_pos, lazy, dtype):
    """mg_bn_train_bwd's one-launch form (round 5: bn_bwd_coop_kernel -- reduce, ordered sum and apply in ONE kernel, rows kept in registers, flag
    hand-shake between the row blocks of a channel group) against the three-launch kernels (mg_bn_bwd_reduce + ordered sum + mg_bn_bwd_apply): same
    dx up to the order of the fp32 sums, same dgamma / dbeta to 1e-5 relative, the residual gradient g EXACTLY; repeated launches give the same bits
    (the generation words advance from launch to launch) and no workgroup ever gave up waiting (mg_coop_error)."""
    import ctypes
    from maggie_amd import kernels as K, hip
    dev = _dev()
    rs = np.random.RandomState(M % 1000 + C + act)
    x = torch.from_numpy(rs.normal(0.2, 1.3, (M, C)).astype(np.float32)).to(dev, dtype)
    dz = torch.from_numpy(rs.normal(size=(M, C)).astype(np.float32)).to(dev, dtype)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32) * rs.choice([-1.0, 1.0], C).astype(np.float32)).to(dev)
    beta = torch.from_numpy(rs.normal(size=C).astype(np.float32)).to(dev)
    sc, sh, mean, invstd = K.bn_finalize(K.colstats(x), M, gamma, beta, None, None, 0.1, 1e-5)
    pack = sc._base.view(-1)
    z = K.affine_act(x, sc, sh, act=act, slope=0.2)
    y = None if lazy else z
    was = hip.lib().mg_set_bn_coop(ctypes.c_int(1))          # (off by default: a hand-shake inside a launch costs more than two kernel boundaries here)
    try:
        outs = [K.bn_train_bwd(dz, y, x, pack, act, 0.2, True, mask_x_pos) for _ in range(4)]
    finally:
        hip.lib().mg_set_bn_coop(ctypes.c_int(was))
    for dx_, dres_, s_ in outs[1:]:
        assert torch.equal(dx_, outs[0][0]) and torch.equal(dres_, outs[0][1]) and torch.equal(s_, outs[0][2])
    dx, dres, sums = outs[0]
    err = ctypes.c_int(-1)
    assert hip.lib().mg_coop_error(ctypes.byref(err)) == 0 and err.value == 0
    # the three-launch kernels (stored activation: they know no mask_from_x): reduce -> ordered sum -> apply
    _, _, s_ref = K.bn_backward(dz, z, x, sc, mean, invstd, M, act=act, slope=0.2, reduce_only=True)
    dx_ref, dres_ref, _ = K.bn_backward(dz, z, x, sc, mean, invstd, M, act=act, slope=0.2, want_dres=True, mask_x_pos=mask_x_pos, sums=s_ref, apply_only=True)
    assert torch.equal(dres, dres_ref)
    assert torch.allclose(sums, s_ref, rtol=1e-5, atol=1e-5 * float(s_ref.abs().max()))
    tol = 8e-3 if dtype == torch.bfloat16 else 1e-3                       # one storage ulp where the sums' last bits move a rounding boundary
    assert (dx.float() - dx_ref.float()).abs().max() <= tol * dx_ref.float().abs().max()
    assert float((dx != dx_ref).float().mean()) <= 2e-3


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('M,C', [(16384, 128), (65536, 64), (4096, 256), (16001, 128), (5000, 32), (1500, 512), (262144, 32)])
def test_batchnorm_backward_sums_by_the_last_row_block(M, C, dtype):
    """mg_bn_bwd_reduce's last-arriver form (round 6, VERDICT round 5 item 4a: mg_set_bn_bwd_tail): the row block that draws the last ticket of its
    channel group adds the partial rows in row order with det_reduce_kernel's arithmetic -- the sums must equal the separate ordered-sum launch BIT FOR
    BIT, whichever block was last, launch after launch (the ticket words reset themselves)."""
    import ctypes
    from maggie_amd import kernels as K, hip
    dev = _dev()
    rs = np.random.RandomState(M % 1000 + C)
    x = torch.from_numpy(rs.normal(0.2, 1.3, (M, C)).astype(np.float32)).to(dev, dtype)
    dz = torch.from_numpy(rs.normal(size=(M, C)).astype(np.float32)).to(dev, dtype)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)).to(dev)
    beta = torch.from_numpy(rs.normal(size=C).astype(np.float32)).to(dev)
    sc, sh, mean, invstd = K.bn_finalize(K.colstats(x), M, gamma, beta, None, None, 0.1, 1e-5)
    z = K.affine_act(x, sc, sh, act=1, slope=0.2)
    _, _, s_ref = K.bn_backward(dz, z, x, sc, mean, invstd, M, act=1, slope=0.2, reduce_only=True)
    was = hip.lib().mg_set_bn_bwd_tail(ctypes.c_int(1))
    try:
        outs = [K.bn_backward(dz, z, x, sc, mean, invstd, M, act=1, slope=0.2, reduce_only=True)[2] for _ in range(5)]
    finally:
        hip.lib().mg_set_bn_bwd_tail(ctypes.c_int(was))
    torch.cuda.synchronize()
    assert float(s_ref.abs().max()) > 0
    for s_ in outs:
        assert torch.equal(s_, s_ref)
