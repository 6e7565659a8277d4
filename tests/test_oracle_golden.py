"""CPU (-m "not gpu"): the oracle restatement against the golden fixtures produced by the reference's own modules
(tests/golden/make_golden.py). `dense_pinned.npz` entries are PINNED (reference code, stock torch ops only);
model_*.npz entries downstream of cv2/spconv pin the reference's orchestration on top of the oracle's restatement of
those third-party libraries (UNPINNED arithmetic)."""
import numpy as np
import pytest
import torch

from helpers import seed_all, load_golden, unpack_bits, reference_layout_state_dict, model_cfg, DSEED, RSEED
from maggie_amd.utils import synth
from oracle import refmodel


def _close(a, b, tol):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()) <= tol


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_dense_path_pinned(mode):
    gold = load_golden('dense_pinned.npz')
    sd = reference_layout_state_dict('image')
    training = mode == 'train'
    batch = synth.synthetic_batch(2, 1, 2, 64, 64, seed=DSEED, train=True, max_inst=10)
    b, n_f, n_i, h, w = 2, 1, 10, 64, 64
    x = batch['image'].reshape(-1, 3, h, w)
    masks = torch.nn.functional.interpolate(batch['mask'].flatten(0, 1), size=(h, w), mode='nearest')
    alphas = batch['alpha'].reshape(-1, n_i, h, w)
    with torch.no_grad():
        emb, mid = refmodel.encoder(sd, 'encoder', torch.cat([x, masks], 1), training)
        emb = refmodel.aspp(sd, 'aspp', emb, training)
        assert _close(emb, gold[mode + '/enc_embedding_aspp'], 2e-5)
        for i, f in enumerate(mid['shortcut']):
            g = gold['%s/fea%d_sum' % (mode, i + 1)]
            assert _close(f.double().sum((2, 3)), g, 1e-4 * max(1.0, np.abs(g).max()))
        assert _close(mid['shortcut'][4], gold[mode + '/fea5'], 2e-5)
        xx, m5, valid, gt_masks = refmodel._dec_prologue(sd, 'decoder', emb, mid, b, n_f, n_i, masks, alphas, training)
        assert _close(xx, gold[mode + '/os8_feat'], 5e-5)
        logits, xf, q, loss_max, _ = refmodel.imd(sd, 'decoder.refine_OS8', xx, m5, training, gt_masks)
        assert _close(logits, gold[mode + '/imd_logits'], 2e-4)
        assert _close(xf, gold[mode + '/imd_out_feat'], 5e-5)
        assert _close(q, gold[mode + '/imd_tokens'], 5e-5)
        if training:
            assert abs(float(loss_max) - float(gold[mode + '/imd_max_loss'])) < 1e-5


def test_spectral_norm_state_pinned():
    """One power iteration per call, also in eval, written back to u/v (spectral_norm.py:22-35,73-80)."""
    gold = load_golden('dense_pinned.npz')
    sd = reference_layout_state_dict('video')
    x = torch.from_numpy(np.random.RandomState(5).normal(size=(1, 64, 8, 8)).astype(np.float32))
    with torch.no_grad():
        y1 = refmodel.sn_conv(sd, 'encoder.layer1.0.conv1', x, 1, 1)
        assert _close(sd['encoder.layer1.0.conv1.module.weight_u'], gold['sn/u1'], 1e-6)
        assert _close(sd['encoder.layer1.0.conv1.module.weight_v'], gold['sn/v1'], 1e-6)
        y2 = refmodel.sn_conv(sd, 'encoder.layer1.0.conv1', x, 1, 1)
        assert _close(sd['encoder.layer1.0.conv1.module.weight_u'], gold['sn/u2'], 1e-6)
    assert _close(y1, gold['sn/y1'], 1e-5) and _close(y2, gold['sn/y2'], 1e-5)
    assert not _close(y1, y2, 1e-7)        # two identical eval calls give different outputs (SURVEY correction 7)


def test_gru_bifusion_losses_pinned():
    gold = load_golden('dense_pinned.npz')
    sd = reference_layout_state_dict('video')
    rs = np.random.RandomState(9)
    feat = torch.from_numpy(rs.normal(size=(1, 3, 128, 8, 8)).astype(np.float32))
    with torch.no_grad():
        o, hdn = refmodel.conv_gru_propagate(sd, 'decoder.os8_temp_module', feat, 3, None, 'bi')
        assert _close(o, gold['gru/out'], 1e-5) and _close(hdn, gold['gru/hidden'], 1e-5)
        f64 = torch.from_numpy(rs.normal(size=(1, 3, 64, 8, 8)).astype(np.float32))
        preds = torch.from_numpy(rs.uniform(size=(1, 3, 2, 64, 64)).astype(np.float32))
        df, db, fu = refmodel.bidirectional_fusion(sd, 'decoder', f64, preds, False)
        assert _close(df, gold['bifuse/df'], 1e-5) and _close(db, gold['bifuse/db'], 1e-5) and _close(fu, gold['bifuse/fused'], 1e-5)
    a = torch.from_numpy(rs.uniform(size=(2, 3, 32, 32)).astype(np.float32))
    g = torch.from_numpy(rs.uniform(size=(2, 3, 32, 32)).astype(np.float32))
    wgt = torch.from_numpy((rs.uniform(size=(2, 3, 32, 32)) > 0.5).astype(np.float32))
    v = lambda t: t.reshape(-1, 1, 32, 32)
    assert abs(float(refmodel.lap_loss(v(a), v(g), v(wgt))) - float(gold['loss/lap'])) < 1e-5
    assert abs(float(refmodel.grad_loss(a, g, wgt)) - float(gold['loss/grad'])) < 1e-6
    assert abs(float(refmodel.regression_loss(a, g, wgt)) - float(gold['loss/l1'])) < 1e-6
    r5 = lambda t: t.reshape(1, 2, 3, 32, 32)
    assert abs(float(refmodel.loss_dtssd(r5(a), r5(g), r5(wgt))) - float(gold['loss/dtssd'])) < 1e-6


@pytest.mark.parametrize('kind,name,b,n_f', [('image', 'model_image_eval.npz', 1, 1), ('video', 'model_video_eval.npz', 1, 3)])
def test_full_model_eval_vs_reference_glue(kind, name, b, n_f):
    gold = load_golden(name)
    sd = reference_layout_state_dict(kind)
    batch = synth.synthetic_batch(b, n_f, 2, 128, 128, seed=DSEED, train=False)
    with torch.no_grad():
        out = refmodel.maggie_forward(sd, model_cfg(kind), batch, False)
    for k in ('alpha_os8', 'alpha_os4', 'alpha_os1', 'refined_masks'):
        assert _close(out[k], gold['out/' + k], 2e-5), k
    dm = out['detail_mask'].numpy()
    assert np.array_equal(dm.reshape(-1), unpack_bits(gold['out/detail_mask'], dm.shape).reshape(-1))


def test_full_model_train_vs_reference_glue():
    gold = load_golden('model_image_train.npz')
    sd = reference_layout_state_dict('image', requires_grad=True)
    batch = synth.synthetic_batch(2, 1, 2, 128, 128, seed=DSEED, train=True, it=10000, max_inst=10)
    seed_all(RSEED)
    out, loss = refmodel.maggie_forward(sd, model_cfg('image'), batch, True)
    loss['total'].backward()
    for k, v in loss.items():
        assert abs(float(v) - float(gold['loss/' + k])) <= 2e-5 * max(1.0, abs(float(v))), k
    names = [str(n) for n in gold['grad_norm_names']]
    norms = dict(zip(names, gold['grad_norms']))
    bad = []
    for n, t in sd.items():
        if t.grad is not None and n in norms and norms[n] > 1e-6:
            if abs(float(t.grad.double().norm()) - norms[n]) > 2e-3 * norms[n]:
                bad.append(n)
    assert not bad, bad[:5]
    assert _close(sd['encoder.bn1.running_mean'].detach(), gold['bn/encoder.bn1.running_mean'], 1e-6)


NEW_GEOMETRIES = {
    # BASELINE configs[2] geometry: image, 4 instances (batch 4 per GPU in training); configs[4] geometry: video, T = 5, 3 instances
    'image_eval_4inst': ('image', False, 1, 1, 4, (128, 128), 0, None, 'model_image_eval_4inst.npz'),
    'image_train_4inst_b4': ('image', True, 4, 1, 4, (128, 128), 10000, 10, 'model_image_train_4inst_b4.npz'),
    'video_eval_t5': ('video', False, 1, 5, 3, (96, 128), 0, None, 'model_video_eval_t5.npz'),
    'video_train_t5': ('video', True, 1, 5, 3, (96, 96), 10000, 10, 'model_video_train_t5.npz'),
}


@pytest.mark.parametrize('case', sorted(NEW_GEOMETRIES))
def test_oracle_matches_reference_glue_on_config2_and_config4_geometries(case):
    """oracle/refmodel.py against the reference's own forward (+ backward) at the instance / frame counts of BASELINE configs[2] and
    configs[4] (tests/golden/make_golden.py --new-geometries): mattes, index map, every loss, gradient norms."""
    kind, train, b, n_f, n_inst, (h, w), it, max_inst, name = NEW_GEOMETRIES[case]
    gold = load_golden(name)
    sd = reference_layout_state_dict(kind, requires_grad=train)
    batch = synth.synthetic_batch(b, n_f, n_inst, h, w, seed=DSEED, train=train, it=it, max_inst=max_inst)
    seed_all(RSEED)
    if train:
        out, loss = refmodel.maggie_forward(sd, model_cfg(kind), batch, True)
        loss['total'].backward()
        for k, v in loss.items():
            assert abs(float(v) - float(gold['loss/' + k])) <= 5e-5 * max(1.0, abs(float(v))), k
        norms = dict(zip([str(n) for n in gold['grad_norm_names']], gold['grad_norms']))
        bad = [n for n, t in sd.items() if t.grad is not None and n in norms and norms[n] > 1e-6
               and abs(float(t.grad.double().norm()) - norms[n]) > 5e-3 * norms[n]]
        assert not bad, bad[:5]
    else:
        with torch.no_grad():
            out = refmodel.maggie_forward(sd, model_cfg(kind), batch, False)
    for k in ('alpha_os8', 'refined_masks') + (('temp_alpha',) if 'out/temp_alpha' in gold.files else ()):
        assert _close(out[k].detach(), gold['out/' + k], 5e-5), k
    dm = out['detail_mask'].numpy()
    assert np.array_equal(dm.reshape(-1), unpack_bits(gold['out/detail_mask'], dm.shape).reshape(-1))


def test_postprocess_oracle_matches_reference_fixture():
    """oracle/postprocess.py against tests/golden/postprocess_pinned.npz (outputs of the reference's own
    maggie/utils/postprocessing.py:reverse_transform_tensor on seeded planes; generator: tests/golden/make_golden.py)."""
    import numpy as np
    import torch
    from helpers import load_golden
    from oracle import postprocess as pp
    gold = load_golden('postprocess_pinned.npz')
    rs = np.random.RandomState(21)
    cases = {'resize_pad': ((2, 3, 40, 56), [{'name': ['resize'], 'ori_size': (torch.tensor(37), torch.tensor(61))},
                                            {'name': ['padding'], 'pad_size': (torch.tensor(5), torch.tensor(8))}]),
             'pad_resize_same': ((1, 2, 32, 48), [{'name': 'resize', 'ori_size': (29, 48)}, {'name': 'padding', 'pad_size': (3, 0)}]),
             'resize_only': ((3, 24, 24), [{'name': 'resize', 'ori_size': (50, 33)}])}
    for key, (shape, info) in cases.items():
        x = torch.from_numpy(rs.uniform(-0.05, 1.05, size=shape).astype(np.float32))
        y = pp.reverse_transform_tensor(x, info).numpy()
        assert y.shape == gold[key].shape
        assert np.abs(y - gold[key]).max() <= 1e-6, key
    a = pp.snap_alpha(np.array([0.0, 1 / 255.0, 0.004, 0.5, 254 / 255.0, 0.9999], np.float32))
    assert a.tolist() == [0.0, 0.0, np.float32(0.004), 0.5, 1.0, 1.0]


def test_preprocess_oracle_matches_reference_fixture():
    """oracle/preprocess.py against tests/golden/preprocess_pinned.npz (outputs of the reference's own ToTensor / Normalize and of
    the him.py item-assembly statements on seeded uint8 inputs): bit-exact."""
    import numpy as np
    from helpers import load_golden, preprocess_inputs, PREPROCESS_CASES
    from oracle import preprocess as pre
    gold = load_golden('preprocess_pinned.npz')
    for key in PREPROCESS_CASES:
        frames, alphas, masks, max_inst = preprocess_inputs(key)
        ids = None if max_inst is None else [int(i) for i in gold[key + '.slot_ids']]
        H, W = frames.shape[1:3]
        img = pre.normalize_frames(frames, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
        a = pre.scale_planes(alphas, max_inst, ids, None, 5)
        m = pre.scale_planes(masks, max_inst, ids, (H // 8, W // 8), 0)
        assert np.array_equal(img, gold[key + '.image']), key
        assert np.array_equal(a, gold[key + '.alpha']), key
        assert np.array_equal(m, gold[key + '.mask']), key
        assert float(a[(a > 0)].min()) >= 5 / 255.0 - 1e-7


def test_metric_oracle_matches_reference_fixture():
    """oracle/metric.py against tests/golden/metric_pinned.npz (the reference's own SAD / MSE / MAD / Grad / dtSSD classes)."""
    import numpy as np
    from helpers import load_golden, metric_inputs, METRIC_CASES
    from oracle import metric as om
    gold = load_golden('metric_pinned.npz')
    for key in METRIC_CASES:
        pred, gt, tri = metric_inputs(key)
        for name, fn in (('SAD', om.sad), ('MSE', om.mse), ('MAD', om.mad), ('Grad', om.grad), ('dtSSD', om.dtssd)):
            score, count = fn(pred, gt, tri)
            ref = gold['%s.%s' % (key, name)]                      # [update() return, score, count, average()]
            assert count == ref[2], (key, name)
            assert abs(score - ref[1]) <= 2e-5 * abs(ref[1]) + 1e-9, (key, name, score, ref[1])
