"""Properties checked at the FULL size of BASELINE.json configs[1] (image 512x512, batch 4, 2 real instances in 10 slots),
where the CPU oracle is too slow to be the checker: size-independent invariants of the path.

  * region ops: pack/unpack round trip, dilation is extensive + monotone + idempotent-on-saturation, stride-2 active set
    contains the image of every fine site (SparseConv2d receptive-field rule);
  * convolution is linear: f(a x + b y) = a f(x) + b f(y) on a real layer geometry (fp32 exact MFMA path);
  * fusion: outside the detail region the refined alpha IS alpha_os8; everything stays in [0, 1];
  * the same step replayed from hipGraphs equals the eager step (full size)."""
import copy

import numpy as np
import pytest
import torch

from helpers import seed_all, DSEED
from test_gpu_model import _dev, _build, _to

pytestmark = pytest.mark.gpu


def test_region_ops_properties_full_size():
    from maggie_amd import kernels as K
    dev = _dev()
    rs = np.random.RandomState(0)
    P, H, W = 40, 512, 512
    a = torch.from_numpy((rs.uniform(size=(P, H, W)) < 0.002).astype(np.uint8)).to(dev)
    bits = K.bits_pack(a, mode=1)
    assert torch.equal(K.bits_unpack_u8(bits, W, (P, H, W)), a)                      # round trip
    d15 = K.bits_dilate(bits, W, width=15)
    d30 = K.bits_dilate(bits, W, width=30)
    u15, u30 = K.bits_unpack_u8(d15, W, (P, H, W)), K.bits_unpack_u8(d30, W, (P, H, W))
    assert bool((u15 >= a).all()) and bool((u30 >= u15).all())                       # extensive, monotone in the element size
    assert int(u15.sum()) > int(a.sum())
    coarse = K.bits_downsample(bits, W)[0]
    uc = K.bits_unpack_u8(coarse, W // 2, (P, H // 2, W // 2))
    fine_img = torch.nn.functional.max_pool2d(a.float(), 2, 2)                       # every fine site lies in a coarse cell
    assert bool((uc.float() >= fine_img).all())
    assert int(uc.sum()) <= 4 * int(a.sum())                                         # each fine site reaches at most 4 coarse outputs


def test_conv_linearity_real_layer_fp32():
    from maggie_amd import kernels as K
    dev = _dev()
    g = torch.Generator(device='cpu').manual_seed(3)
    N, H, W, Cin, Cout = 4, 128, 128, 64, 64                                         # encoder layer1 geometry at 512x512, batch 4
    x = torch.randn((N * H * W, Cin), generator=g).to(dev)
    y = torch.randn((N * H * W, Cin), generator=g).to(dev)
    w = (torch.randn((Cout, 9, Cin), generator=g) / 24).to(dev)
    f = lambda t: K.conv_fprop(t, w, mode=K.MODE_CONV, N=N, Hin=H, Win=W, R=3, S=3, stride=1, pad=1, dil=1)
    lhs = f(2.0 * x - 3.0 * y)
    rhs = 2.0 * f(x) - 3.0 * f(y)
    assert float((lhs - rhs).abs().max()) <= 2e-4 * float(rhs.abs().max())
    dy = torch.randn((N * H * W, Cout), generator=g).to(dev)
    dw = lambda t: K.conv_wgrad(t, dy, cout=Cout, mode=K.MODE_CONV, N=N, Hin=H, Win=W, Hout=H, Wout=W, R=3, S=3, stride=1, pad=1, dil=1)
    l2, r2 = dw(2.0 * x - 3.0 * y), 2.0 * dw(x) - 3.0 * dw(y)
    assert float((l2 - r2).abs().max()) <= 5e-4 * float(r2.abs().max())


def test_full_size_step_invariants_and_graph_replay():
    from maggie_amd.utils import synth
    dev = _dev()
    model, _ = _build('image', dev, True)
    batch = _to(synth.synthetic_batch(4, 1, 2, 512, 512, seed=DSEED, train=True, max_inst=10, it=10000), dev)
    state = copy.deepcopy(model.state_dict())
    res = []
    for graphs in (False, True, True, True):                                         # eager, first sight, capture, replay
        model.load_state_dict(state)
        model.hip_graphs = graphs
        model.zero_grad(set_to_none=True)
        seed_all(11)
        out, loss = model(batch)
        loss['total'].backward()
        res.append((out, float(loss['total'].detach()),
                    {n: p.grad.float().norm().item() for n, p in model.named_parameters() if p.grad is not None}))
    out, lv, _ = res[0]
    a8, a, m = out['alpha_os8'].float(), out['refined_masks'].float(), out['detail_mask'].float()
    for k in ('alpha_os8', 'alpha_os4', 'alpha_os1', 'refined_masks'):
        v = out[k].detach().float()
        assert v.shape == (4, 1, 10, 512, 512)                 # 2 real instances in 10 slots
        assert bool(torch.isfinite(v).all()) and float(v.min()) >= 0.0 and float(v.max()) <= 1.0
    assert bool((a[m == 0] == a8[m == 0]).all()), 'outside the detail region the refined alpha must be alpha_os8 (fusion)'
    assert 0.0 < float(m.mean()) < 1.0
    assert np.isfinite(lv)
    # the replayed step reproduces the eager one (fp32): loss within 1e-3 relative, gradient norms within 2 %
    out_r, lv_r, gn_r = res[3]
    assert abs(lv_r - lv) <= 1e-3 * abs(lv)
    gn = res[0][2]
    rel = sorted(abs(gn_r[k] - gn[k]) / max(gn[k], 1e-12) for k in gn)
    assert rel[len(rel) // 2] <= 2e-2 and rel[int(0.9 * len(rel))] <= 0.2
    assert float((out_r['alpha_os8'].detach().float() - a8.detach()).abs().mean()) <= 1e-4
