"""Bit-reproducibility of the hot path (MAGGIE_DETERMINISTIC, on by default).

The reference runs with torch.backends.cudnn.deterministic = True, benchmark = False (tools/main.py:135-136): two runs of one training step
give the same bits, and with them the same active-pixel index map -- a threshold of the coarse alpha (maggie/utils/utils.py:31). This build
forms every cross-workgroup fp32 sum as "one partial per workgroup, added in index order" (csrc/det.hip) instead of atomicAdd; these tests hold
it to EXACT equality: the same step twice (eager and replayed from hipGraphs), the ordered-sum kernel against a host sum in the same order, and
the kernels that used to end in atomics run many times on the same input."""
import copy

import numpy as np
import pytest
import torch

from helpers import seed_all, DSEED
from test_gpu_model import _dev, _build, _to

pytestmark = pytest.mark.gpu


def _host_ordered_sum(slots):
    """slots [nblk, nv] fp32 -> [nv]: the order of csrc/det.hip:det_reduce_kernel (16 chunks of ceil(nblk / 16) slots; inside a chunk four running
    sums over (index - chunk start) mod 4, combined as (a0 + a1) + (a2 + a3); the chunks in order)."""
    nblk, nv = slots.shape
    cs = (nblk + 15) // 16
    out = np.zeros(nv, np.float32)
    for k in range(16):
        b0, b1 = k * cs, min(nblk, (k + 1) * cs)
        a = [np.zeros(nv, np.float32) for _ in range(4)]
        b = b0
        while b + 3 < b1:
            for j in range(4):
                a[j] = (a[j] + slots[b + j]).astype(np.float32)
            b += 4
        while b < b1:
            a[0] = (a[0] + slots[b]).astype(np.float32)
            b += 1
        out = (out + ((a[0] + a[1]).astype(np.float32) + (a[2] + a[3]).astype(np.float32)).astype(np.float32)).astype(np.float32)
    return out


@pytest.mark.parametrize('nblk,groups,nv', [(1, 1, 5), (7, 2, 33), (64, 4, 130), (513, 1, 64), (1024, 3, 17)])
def test_ordered_slot_sum_matches_the_host_sum_in_the_same_order(nblk, groups, nv):
    import ctypes
    from maggie_amd import hip
    dev = _dev()
    rs = np.random.RandomState(nblk + nv)
    slots = (rs.normal(size=(groups, nblk, nv)) * np.exp(rs.uniform(-8, 8, size=(groups, nblk, nv)))).astype(np.float32)
    d_slots = torch.from_numpy(slots).to(dev)
    outs = []
    for _ in range(3):
        dst = torch.zeros((groups, nv), device=dev)
        hip.call('mg_det_reduce_test', hip.ptr(d_slots), ctypes.c_int(nblk), ctypes.c_int(groups), ctypes.c_int(nv), hip.ptr(dst), hip.stream())
        outs.append(dst.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    for g in range(groups):
        ref = _host_ordered_sum(slots[g])
        assert np.array_equal(outs[0][g], ref), (g, np.abs(outs[0][g] - ref).max())


def _snapshot(model, out, loss):
    snap = {'loss/' + k: v.detach().float().cpu().clone() for k, v in loss.items()}
    for k, v in out.items():
        if torch.is_tensor(v):
            snap['out/' + k] = v.detach().cpu().clone()
    for n, p in model.named_parameters():
        if p.grad is not None:
            snap['grad/' + n] = p.grad.detach().cpu().clone()
    for n, v in model.state_dict().items():
        if n.endswith(('running_mean', 'running_var', 'weight_u', 'weight_v', 'num_batches_tracked')):
            snap['state/' + n] = v.detach().cpu().clone()
    return snap


def _assert_same_bits(a, b, what):
    assert set(a) == set(b), (what, set(a) ^ set(b))
    bad = []
    for k in a:
        x, y = a[k], b[k]
        if not torch.equal(x, y):
            d = (x.double() - y.double()).abs()
            bad.append((k, int((x != y).sum()), float(d.max())))
    assert not bad, '%s: %d tensors differ, e.g. %s' % (what, len(bad), bad[:6])


@pytest.mark.parametrize('kind,bf16,it,cap', [('image', False, 10000, 1.0), ('image', True, 10000, 1.0), ('image', False, 100, 1.0), ('video', False, 10000, 1.0),
                                              ('video', True, 10000, 1.0), ('image', True, 100, 'auto'), ('video', True, 10000, 'auto')])
def test_train_step_is_bit_reproducible(kind, bf16, it, cap):
    """the same training step twice -- eagerly, from the capturing step and from pure hipGraph replays -- gives
    IDENTICAL outputs (all alphas, the detail mask), losses, gradients and updated buffers (BatchNorm running statistics, SpectralNorm u / v).
    it = 10000: the detail region is guided by the model's own coarse alpha (its threshold is then reproducible, too); it = 100: by the ground
    truth. Dropout stays ON (its counter-based mask is a function of (seed, step): restored with the state)."""
    from maggie_amd import hip
    from maggie_amd.utils import synth
    assert hip.DETERMINISTIC, 'these tests hold the default (deterministic) mode'
    dev = _dev()
    n_f = 3 if kind == 'video' else 1
    b = 2 if kind == 'image' else 1
    model, _ = _build(kind, dev, True)
    batch = _to(synth.synthetic_batch(b, n_f, 2, 64, 64, seed=DSEED, train=True, max_inst=10, it=it), dev)
    state = copy.deepcopy(model.state_dict())

    def step(graphs):
        model.load_state_dict(state)
        _reset_dropout(model)
        model.hip_graphs = graphs
        model.zero_grad(set_to_none=True)
        seed_all(5)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
            out, loss = model(batch)
        loss['total'].backward()
        return _snapshot(model, out, loss)

    model.decoder.sparse_capacity_frac = cap
    if cap == 'auto':
        # the product default: the sparse head's row capacities follow the workload (1.5x the live sites of the first step). They size the
        # persistent grids, so the bits are those of ONE capacity setting: two throw-away steps let them settle, then everything below holds
        step(False), step(False)
        st = list(model.decoder._sparse_auto.values())[0]
        assert st['tuned'] and st['caps'][0] < st['full'][0]
    e0, e1, e2 = step(False), step(False), step(False)
    _assert_same_bits(e1, e0, 'eager run 2 vs eager run 1')
    _assert_same_bits(e2, e0, 'eager run 3 vs eager run 1')
    g = [step(True) for _ in range(5)]                      # first sight (eager), capture, then replays
    _assert_same_bits(g[3], g[2], 'replay 2 vs replay 1')
    _assert_same_bits(g[4], g[2], 'replay 3 vs replay 1')
    _assert_same_bits(g[2], e0, 'replayed step vs eager step')
    _assert_same_bits(g[1], e0, 'capturing step vs eager step')
    m = e0['out/detail_mask'].float()
    assert 0.0 < float(m.mean()) < 1.0, 'the step must refine something for the index map to mean anything'


@pytest.mark.parametrize('kind', ['image', 'video'])
def test_parked_slab_reductions_leave_the_step_unchanged(kind):
    """The weight-gradient slab reductions parked until the gradients meet (functional.PARKED, one launch per backward pass) against one reduce
    launch per layer (MAGGIE_PARK_WGRAD=0): the same step, eagerly and from a replayed graph, bit for bit -- every gradient, loss and buffer.
    (The grouped form for a weight used by several convolution calls -- video: ConvGRU gates, refine_OS8 per frame -- is NOT the same arithmetic as
    per-call reductions + autograd adds: one fp32 sum, rounded once. It is switched off for this comparison and checked on its own: the replayed
    step against the eager one here, its values in test_gpu_conv.py::test_one_weight_used_by_several_identical_convolutions_reduces_all_slabs_once.)"""
    from maggie_amd import functional as MF
    from maggie_amd.utils import synth
    dev = _dev()
    n_f = 3 if kind == 'video' else 1
    model, _ = _build(kind, dev, True)
    batch = _to(synth.synthetic_batch(2 if kind == 'image' else 1, n_f, 2, 64, 64, seed=DSEED, train=True, max_inst=10, it=10000), dev)
    state = copy.deepcopy(model.state_dict())

    def step(graphs):
        model.load_state_dict(state)
        _reset_dropout(model)
        model.hip_graphs = graphs
        model.zero_grad(set_to_none=True)
        seed_all(5)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out, loss = model(batch)
        loss['total'].backward()
        assert MF.PARKED == [], 'every parked reduction must have run by the end of the backward pass'
        return _snapshot(model, out, loss)

    assert MF.PARK_WGRAD, 'parking is the default'
    snaps = {}
    grouped = MF.GROUP_WGRAD
    try:
        if kind == 'video' and grouped:
            snaps['grouped', 'eager'] = step(False)
            snaps['grouped', 'graph'] = [step(True) for _ in range(4)][-1]
            _assert_same_bits(snaps['grouped', 'graph'], snaps['grouped', 'eager'], 'grouped reductions: replay vs eager')
        MF.GROUP_WGRAD = False
        for park in (True, False):
            MF.PARK_WGRAD = park
            torch.cuda.synchronize()                                      # never destroy a graph the device may still be executing
            for store in ('_trunk_graphs', '_trunk_enc_graphs', '_detail_graphs', '_detail_names'):
                model.__dict__.get(store, {}).clear()              # graphs captured under the other setting hold its launches
            snaps[park, 'eager'] = step(False)
            snaps[park, 'graph'] = [step(True) for _ in range(4)][-1]
    finally:
        MF.PARK_WGRAD = True
        MF.GROUP_WGRAD = grouped
    if ('grouped', 'eager') in snaps:
        # the grouped weights differ from the per-call form by 16-bit roundings only; everything else has the same bits
        a, b = snaps['grouped', 'eager'], snaps[True, 'eager']
        diff = [k for k in a if not torch.equal(a[k], b[k])]
        assert diff and all(k.startswith('grad/') and ('os8_temp_module' in k or 'refine_OS8' in k) for k in diff), diff
        for k in diff:
            assert float((a[k].double() - b[k].double()).abs().max()) <= 2.0 ** -6 * float(b[k].double().abs().max()), k
    _assert_same_bits(snaps[True, 'eager'], snaps[False, 'eager'], 'eager step, parked vs per-layer reductions')
    _assert_same_bits(snaps[True, 'graph'], snaps[False, 'graph'], 'replayed step, parked vs per-layer reductions')
    _assert_same_bits(snaps[True, 'graph'], snaps[True, 'eager'], 'parked reductions: replay vs eager')
    assert any(k.startswith('grad/') for k in snaps[True, 'eager'])


def _reset_dropout(model):
    """The sparse head's dropout draws from a counter (seed, step) kept on the device (maggie_amd/sparse_head.py:DeviceRng, created at the first
    training forward under torch.initial_seed()): put it back to step 0 of seed 5 -- what the first run, made under seed_all(5), found."""
    rng = model.decoder.__dict__.get('_head_rng')
    if rng is not None:
        rng.state.copy_(torch.tensor([5, 0], dtype=torch.int64))


def test_many_steps_stay_bit_identical_between_two_runs():
    """Two training RUNS of 6 optimizer steps each from the same state (FlatAdamW with the clip folded in: the gradient norm is one of the ordered
    sums) end in bit-identical parameters -- the steps of a run feed each other, so a single differing bit anywhere would spread."""
    from maggie_amd.optim import FlatAdamW
    from maggie_amd.utils import synth
    dev = _dev()
    batch = _to(synth.synthetic_batch(2, 1, 2, 64, 64, seed=DSEED, train=True, max_inst=10, it=10000), dev)
    finals = []
    for run in range(2):
        model, _ = _build('image', dev, True)
        model.hip_graphs = True
        opt = FlatAdamW(model.parameters(), lr=1.5e-4 / 25, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=0.01)
        seed_all(7)
        losses = []
        for _ in range(6):
            opt.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                out, loss = model(batch)
            loss['total'].backward()
            opt.step()
            losses.append(float(loss['total'].detach()))
        finals.append((losses, torch.cat([p.detach().flatten().float().cpu() for p in model.parameters()]), out['detail_mask'].cpu().clone()))
    assert finals[0][0] == finals[1][0], (finals[0][0], finals[1][0])
    assert torch.equal(finals[0][1], finals[1][1]), 'parameters after 6 steps: %d differ' % int((finals[0][1] != finals[1][1]).sum())
    assert torch.equal(finals[0][2], finals[1][2])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_former_atomic_kernels_repeat_exactly(dtype):
    """The kernels that used to end in fp32 atomics, each run 8 times on one input under load (a copy stream keeps the memory system busy so
    that workgroup finishing order varies): BatchNorm forward (statistics from the column kernel and from a conv epilogue) and backward,
    bias / activation backward, the token side of both attention directions, the einsum backward, LayerNorm backward."""
    from maggie_amd import kernels as K
    dev = _dev()
    g = torch.Generator(device='cpu').manual_seed(1)
    M, C = 70000, 64
    x = torch.randn((M, C), generator=g).to(dev, dtype)
    dy = torch.randn((M, C), generator=g).to(dev, dtype)
    gamma, beta = torch.rand(C, generator=g).to(dev) + 0.5, torch.randn(C, generator=g).to(dev)
    noise_src = torch.randn(1 << 24, device=dev)
    side = torch.cuda.Stream()

    def under_load(fn):
        outs = []
        for i in range(8):
            with torch.cuda.stream(side):                         # unrelated traffic: changes which workgroup finishes first
                for _ in range(i % 3 + 1):
                    noise_src.mul_(1.0000001)
            outs.append(fn())
        torch.cuda.synchronize()
        return outs

    def same(outs, what):
        for o in outs[1:]:
            for a, b_ in zip(outs[0], o):
                assert torch.equal(a, b_), '%s: %d elements differ (max %g)' % (what, int((a != b_).sum()), float((a.float() - b_.float()).abs().max()))

    def bn_fwd_bwd():
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y, pack = K.bn_train_fwd(x, gamma, beta, rm, rv, 0.1, 1e-5, K.ACT_LRELU, 0.2)
        dx, _, sums = K.bn_train_bwd(dy, y, x, pack, K.ACT_LRELU, 0.2)
        return y, pack, rm, rv, dx, sums
    same(under_load(bn_fwd_bwd), 'BatchNorm forward / backward')

    w = (torch.randn((C, 9, C), generator=g) / 24).to(dev, dtype)
    N, H, W = 4, 128, 128
    x_big = torch.randn((N * H * W, C), generator=g).to(dev, dtype)

    def conv_stats():
        st = torch.zeros((K.conv_stat_rows(N * H * W, N, H, W), 2 * C), device=dev)
        y = K.conv_fprop(x_big, w, mode=K.MODE_CONV, N=N, Hin=H, Win=W, R=3, S=3, stride=1, pad=1, dil=1, stats=st)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        pack = K.bn_finalize(st, N * H * W, gamma, beta, rm, rv, 0.1, 1e-5)
        return (y,) + tuple(pack) + (rm, rv)
    same(under_load(conv_stats), 'conv epilogue statistics + finalize')

    same(under_load(lambda: K.bias_act_bwd(dy, x, True)), 'bias / ReLU backward')
    if dtype == torch.float32:
        B, T, L, D, NID = 2, 10, 4096, 128, 11
        qk = torch.randn((B, T, D), generator=g).to(dev)
        btab = torch.randn((B, T, NID), generator=g).to(dev)
        feat = torch.randn((B, L, D), generator=g).to(dev)
        ids = torch.randint(0, NID, (B, L), generator=g).to(dev, torch.int32)
        dctx = torch.randn((B, T, D), generator=g).to(dev)

        def tok():
            p, ctx = K.attn_tok_fwd(qk, btab, feat, ids, 0.1)
            return (p, ctx) + tuple(K.attn_tok_bwd(p, feat, qk, ids, dctx, None, 0.1, NID))
        same(under_load(tok), 'tokens <- features attention')
        kq, vp = torch.randn((B, T, D), generator=g).to(dev), torch.randn((B, T, D), generator=g).to(dev)
        b2 = torch.randn((B, NID, T), generator=g).to(dev)
        ob = torch.randn(D, generator=g).to(dev)
        dout = torch.randn((B, L, D), generator=g).to(dev)

        def feat_attn():
            out, p = K.attn_feat_fwd(feat, kq, b2, vp, ob, None, ids, 0.1)
            return (out, p) + tuple(K.attn_feat_bwd(dout, p, feat, kq, vp, ids, 0.1, NID, True))
        same(under_load(feat_attn), 'features <- tokens attention')
    r = torch.randn((M, C), generator=g).to(dev, dtype)

    def ln():
        y, rstat = K.rows_add_layernorm(x, r, gamma, beta, 1e-5)
        return (y,) + tuple(K.rows_add_layernorm_bwd(dy, x, r, gamma, rstat))
    same(under_load(ln), 'LayerNorm backward')


def test_batchnorm_statistics_rows_cover_every_conv_tile():
    """conv_stat_rows() must be at least the number of output tiles of EVERY kernel form (else two tiles add to one row and their order shows):
    for a set of real layer geometries, the statistics of the deterministic buffer equal those of a single-row-per-tile oracle sum exactly
    enough, and repeated launches are bit-identical."""
    from maggie_amd import kernels as K
    dev = _dev()
    g = torch.Generator(device='cpu').manual_seed(2)
    for (N, H, W, Cin, Cout, k, stride, transposed) in [(4, 64, 64, 128, 128, 3, 1, False), (4, 128, 128, 64, 64, 3, 1, False),
                                                         (2, 36, 52, 32, 32, 3, 1, False), (4, 32, 32, 256, 128, 1, 1, False),
                                                         (4, 64, 64, 64, 128, 3, 2, False), (2, 16, 16, 64, 64, 4, 2, True),
                                                         (4, 256, 256, 8, 32, 3, 1, False), (4, 32, 32, 512, 512, 3, 2, False)]:
        mode = K.MODE_TCONV if transposed else K.MODE_CONV
        pad = 1 if k > 1 else 0
        Ho, Wo = K.conv_out_size(mode, H, k, stride, pad, 1), K.conv_out_size(mode, W, k, stride, pad, 1)
        x = torch.randn((N * H * W, Cin), generator=g).to(dev, torch.bfloat16)
        w = (torch.randn((Cout, k * k, Cin), generator=g) / (k * Cin ** 0.5)).to(dev, torch.bfloat16)
        outs = []
        for _ in range(4):
            st = torch.zeros((K.conv_stat_rows(N * Ho * Wo, N, Ho, Wo), 2 * Cout), device=dev)
            y = K.conv_fprop(x, w, mode=mode, N=N, Hin=H, Win=W, R=k, S=k, stride=stride, pad=pad, dil=1, stats=st)
            outs.append((y, st))
        for y, st in outs[1:]:
            assert torch.equal(y, outs[0][0]) and torch.equal(st, outs[0][1]), (N, H, W, Cin, Cout, k, stride)
        yf = outs[0][0].float()
        s = outs[0][1].double().sum(0).float()
        assert torch.allclose(s[:Cout], yf.sum(0), rtol=2e-3, atol=2e-2), (N, H, W, Cin, Cout, k, stride)
        assert torch.allclose(s[Cout:], (yf * yf).sum(0), rtol=2e-3, atol=2e-2)
        # every word of the buffer received at most one addition: a row is either untouched (zero) or one tile's partial; with FEWER rows than
        # tiles the sums stay right (32-replica mode) -- that form is what MAGGIE_DETERMINISTIC=0 runs
        from maggie_amd import hip
        st32 = torch.zeros((K.STAT_REPLICAS, 2 * Cout), device=dev)
        hip.set_deterministic(False)
        try:
            K.conv_fprop(x, w, mode=mode, N=N, Hin=H, Win=W, R=k, S=k, stride=stride, pad=pad, dil=1, stats=st32)
        finally:
            hip.set_deterministic(True)
        assert torch.allclose(st32.sum(0), s, rtol=1e-4, atol=1e-2)
        # ... and in deterministic mode a buffer with fewer rows than tiles is an ERROR (two tiles on one word would be order-dependent)
        if K.conv_stat_rows(N * Ho * Wo, N, Ho, Wo) > 64 and N * Ho * Wo >= 64 * 64:
            with pytest.raises(hip.MaggieHipError):
                K.conv_fprop(x, w, mode=mode, N=N, Hin=H, Win=W, R=k, S=k, stride=stride, pad=pad, dil=1, stats=torch.zeros((K.STAT_REPLICAS, 2 * Cout), device=dev))


@pytest.mark.parametrize('kind,size', [('image', 128), ('video', 64)])
def test_operand_path_batchnorm_gives_the_bits_of_the_stored_form(kind, size):
    """Round 5: with MAGGIE_LAZY_BN (default) the conv -> BatchNorm -> activation -> conv chains of the encoder blocks,
    shortcut branches, stem and decoder blocks never store the normalised activation -- the consumer's forward and weight-gradient kernels form it
    on load, the BatchNorm backward re-forms the activation mask from the raw conv output. That must change NOTHING: the same bf16 training step
    with the operand path on and off gives identical outputs, losses, gradients and buffers (eager, and replayed from hipGraphs), and the lazy
    path really ran (the apply kernel is launched for fewer layers)."""
    from maggie_amd import functional as MF, hip, sparse_head
    from maggie_amd.utils import synth
    dev = _dev()
    n_f = 3 if kind == 'video' else 1
    model, _ = _build(kind, dev, True)
    batch = _to(synth.synthetic_batch(2 if kind == 'image' else 1, n_f, 2, size, size, seed=DSEED, train=True, max_inst=10, it=10000), dev)
    state = copy.deepcopy(model.state_dict())
    sparse_was, sparse_head.LAZY_SPARSE = sparse_head.LAZY_SPARSE, True      # the sparse head's (off-by-default) operand path is held to the same equality
    calls = {}
    orig_call = hip.call

    def counting_call(name, *a, **kw):
        calls[name] = calls.get(name, 0) + 1
        return orig_call(name, *a, **kw)

    def step(lazy, graphs):
        MF.LAZY_BN = lazy
        torch.cuda.synchronize()                                      # never destroy a graph the device may still be executing
        for store in ('_trunk_graphs', '_trunk_enc_graphs', '_detail_graphs'):
            model.__dict__.get(store, {}).clear()
        snaps = []
        for _ in range(3 if graphs else 1):
            model.load_state_dict(state)
            _reset_dropout(model)
            model.hip_graphs = graphs
            model.zero_grad(set_to_none=True)
            seed_all(5)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                out, loss = model(batch)
            loss['total'].backward()
            snaps.append(_snapshot(model, out, loss))
        return snaps[-1]

    try:
        hip.call = MF.K.hip.call = counting_call
        calls.clear(); stored = step(False, False); n_stored = dict(calls)
        calls.clear(); lazy = step(True, False); n_lazy = dict(calls)
        hip.call = MF.K.hip.call = orig_call
        _assert_same_bits(lazy, stored, 'operand-path BatchNorm vs stored BatchNorm (eager)')
        fused = n_lazy.get('mg_bn_finalize', 0) - n_stored.get('mg_bn_finalize', 0)
        print(kind, 'layers on the operand path:', fused, '| mg_bn_train_fwd calls', n_stored.get('mg_bn_train_fwd'), '->', n_lazy.get('mg_bn_train_fwd'))
        assert fused >= (6 if kind == 'image' else 3) and n_lazy.get('mg_bn_train_fwd', 0) == n_stored.get('mg_bn_train_fwd', 0) - fused
        _assert_same_bits(step(True, True), stored, 'operand-path BatchNorm replayed from hipGraphs vs stored BatchNorm (eager)')
    finally:
        hip.call = MF.K.hip.call = orig_call
        MF.LAZY_BN = True
        sparse_head.LAZY_SPARSE = sparse_was
        torch.cuda.synchronize()                                      # never destroy a graph the device may still be executing
        for store in ('_trunk_graphs', '_trunk_enc_graphs', '_detail_graphs'):
            model.__dict__.get(store, {}).clear()


def test_video_clip_length_8_trains_and_replays_like_eager():
    """The reference's own video training shape has clip_length 8 (configs/maggie_video.yaml:32; bench.py --video --frames 8): two optimizer
    steps of a T = 8 clip stay finite, and the step replayed from hipGraphs gives the bits of the eager step (the ConvGRU recurrence and the
    bidirectional fusion walk 8 frames; T = 3 and T = 5 are the fixture geometries)."""
    from maggie_amd.optim import FlatAdamW
    from maggie_amd.utils import synth
    dev = _dev()
    model, _ = _build('video', dev, True)
    batch = _to(synth.synthetic_batch(1, 8, 2, 96, 96, seed=DSEED, train=True, max_inst=10, it=10000), dev)
    state = copy.deepcopy(model.state_dict())

    def step(graphs):
        model.load_state_dict(state)
        _reset_dropout(model)
        model.hip_graphs = graphs
        model.zero_grad(set_to_none=True)
        seed_all(5)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out, loss = model(batch)
        loss['total'].backward()
        return _snapshot(model, out, loss)

    e0 = step(False)
    assert e0['out/alpha_os1'].shape == (1, 8, 10, 96, 96) and all(bool(torch.isfinite(v.float()).all()) for v in e0.values())
    for k in ('loss/loss_temp', 'loss/total'):
        assert k in e0 and bool(torch.isfinite(e0[k]).all()), k
    g = [step(True) for _ in range(4)]
    _assert_same_bits(g[3], g[2], 'T = 8: replay 2 vs replay 1')
    _assert_same_bits(g[2], e0, 'T = 8: replayed step vs eager step')
    # ... and it trains: two FlatAdamW steps with the reference's clip stay finite
    model.load_state_dict(state)
    opt = FlatAdamW([p for p in model.parameters() if p.requires_grad], lr=5e-5 / 25, weight_decay=0.01, max_grad_norm=0.01)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        seed_all(5)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            _, loss = model(batch)
        loss['total'].backward()
        opt.step()
        assert bool(torch.isfinite(loss['total']))
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    torch.cuda.synchronize()                                      # never destroy a graph the device may still be executing
    for store in ('_trunk_graphs', '_trunk_enc_graphs', '_detail_graphs'):
        model.__dict__.get(store, {}).clear()


def test_side_stream_shortcut_branches_keep_the_step_reproducible():
    """MAGGIE_SIDE_SHORTCUTS (round 5, measured slower and OFF by default: DESIGN.md 11.11): the encoder's three fine shortcut branches issued on a
    side stream next to the instance-token chain, forward and backward -- a parallel branch of the captured graphs. The library's ordered sums have a
    slot scratch per stream (mg_det_side_stream), so the step stays bit-reproducible: replay == replay == eager, and against the one-stream step only
    the accumulation order of the three tapped activations' gradients differs (autograd's add instead of the data-gradient epilogue)."""
    from maggie_amd import functional as MF
    from maggie_amd.utils import synth
    dev = _dev()
    model, _ = _build('image', dev, True)
    batch = _to(synth.synthetic_batch(2, 1, 2, 64, 64, seed=DSEED, train=True, max_inst=10, it=10000), dev)
    state = copy.deepcopy(model.state_dict())

    def step(graphs):
        model.load_state_dict(state)
        _reset_dropout(model)
        model.hip_graphs = graphs
        model.zero_grad(set_to_none=True)
        seed_all(5)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out, loss = model(batch)
        loss['total'].backward()
        torch.cuda.synchronize()
        return _snapshot(model, out, loss)

    def clear():
        torch.cuda.synchronize()                                      # never destroy a graph the device may still be executing
        for store in ('_trunk_graphs', '_trunk_enc_graphs', '_detail_graphs', '_detail_names'):
            model.__dict__.get(store, {}).clear()

    ref = step(False)
    keep = MF.SIDE_SHORTCUTS
    MF.SIDE_SHORTCUTS = True
    try:
        clear()
        e0 = step(False)
        assert MF.side_lane(dev) is not None
        g = [step(True) for _ in range(4)]
    finally:
        MF.SIDE_SHORTCUTS = keep
        clear()
    _assert_same_bits(g[3], g[2], 'side stream: replay 2 vs replay 1')
    _assert_same_bits(g[2], e0, 'side stream: replayed step vs eager step')
    for k in ('out/alpha_os1', 'out/alpha_os8', 'loss/total'):
        assert torch.equal(e0[k], ref[k]), k                       # the forward pass does the same arithmetic on either stream
    worst = max(float((e0[k].double() - ref[k].double()).abs().max()) / max(float(ref[k].double().abs().max()), 1e-12) for k in ref if k.startswith('grad/'))
    assert worst <= 0.25, worst                                    # (bf16, BatchNorm over a handful of samples: a missing or doubled gradient would be O(1))
