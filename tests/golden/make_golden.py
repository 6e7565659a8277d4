"""Generate the golden fixtures in this directory FROM THE REFERENCE (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's own Python modules are imported from /root/reference by oracle/ref_loader.py (nothing is
copied) and run on CPU fp32 with
  * weights  = maggie_amd.utils.synth.fill_state_dict_(state_dict, seed)   (regenerable, not stored)
  * inputs   = maggie_amd.utils.synth.synthetic_batch(...)                 (regenerable, not stored)
Only OUTPUTS are stored (npz, compressed).

Pinning status of each fixture:
  PINNED    -- produced by the reference's own code with stock torch ops only (no stand-in on the data path):
               dense_*.npz (SpectralNorm state, encoder, ASPP, os32->os8, InstanceMatteDecoder, ConvGRU,
               losses) and the `alpha_os8` / `loss_*_os8` / `loss_max_atten` entries of model_*.npz.
  UNPINNED  -- anything downstream of `cv2.dilate` or `spconv`: the reference's glue code ran on top of
               oracle/standins (our restatement of those third-party libraries), so these pin the reference's
               orchestration, not the third-party arithmetic: detail_mask, alpha_os4, alpha_os1, refined_masks ...
"""
import copy
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from oracle import ref_loader                                   # noqa: E402
from maggie_amd.utils import synth, config                      # noqa: E402

WSEED = 7
DSEED = 3
RSEED = 11


def seed_all(s):
    np.random.seed(s)
    random.seed(s)
    torch.manual_seed(s)


def build(ns, kind):
    mc = config.MODEL_IMAGE if kind == 'image' else config.MODEL_VIDEO
    cls = ns.MaGGIe if kind == 'image' else ns.MaGGIe_Temp
    m = cls(ns.CfgNode(copy.deepcopy(mc)))
    sd = m.state_dict()
    synth.fill_state_dict_(sd, WSEED)
    m.load_state_dict(sd)
    return m


def pack(t):
    t = t.detach()
    if t.dtype == torch.uint8 or t.dtype == torch.bool:
        return np.packbits(t.numpy().astype(np.uint8).reshape(-1))
    return t.float().numpy()


def model_fixture(ns, kind, train, b, n_f, n_inst, hw, it, max_inst, name, keep=None):
    """`hw`: int (square) or (h, w). `keep`: output names to store (None = all) -- the larger-geometry fixtures keep the final matte, the
    coarse matte and the index map only."""
    m = build(ns, kind)
    m.train(train)
    h_, w_ = (hw, hw) if isinstance(hw, int) else hw
    batch = synth.synthetic_batch(b, n_f, n_inst, h_, w_, seed=DSEED, train=train, it=it, max_inst=max_inst)
    seed_all(RSEED)
    out = {}
    if train:
        o, loss = m(batch)
        loss['total'].backward()
        for k, v in loss.items():
            out['loss/' + k] = np.float32(float(v))
        gn = {}
        for n, p in m.named_parameters():
            if p.grad is not None:
                gn[n] = float(p.grad.double().norm())
        keys = sorted(gn)
        out['grad_norm_names'] = np.array(keys)
        out['grad_norms'] = np.array([gn[k] for k in keys], np.float64)
        sd = m.state_dict()
        out['bn/encoder.bn1.running_mean'] = sd['encoder.bn1.running_mean'].numpy()
        out['bn/decoder.refine_OS1.1.running_var'] = sd['decoder.refine_OS1.1.running_var'].numpy()
    else:
        with torch.no_grad():
            o = m(batch)
    for k, v in o.items():
        if torch.is_tensor(v) and (keep is None or k in keep):
            out['out/' + k] = pack(v)
            out['shape/' + k] = np.array(v.shape)
    sd = m.state_dict()
    out['sn/encoder.conv1.module.weight_u'] = sd['encoder.conv1.module.weight_u'].numpy()
    out['meta'] = np.array([b, n_f, n_inst, h_, it, -1 if max_inst is None else max_inst, WSEED, DSEED, RSEED] + ([] if isinstance(hw, int) else [w_]))
    np.savez_compressed(os.path.join(HERE, name), **out)
    print('wrote', name, {k: (v.shape if hasattr(v, 'shape') else v) for k, v in list(out.items())[:4]})


def dense_fixture(ns):
    """PINNED: reference modules that need no third-party stand-in."""
    out = {}
    torch.manual_seed(0)
    # SpectralNorm state after 1 and 2 calls (spectral_norm.py:22-35,73-80)
    m = build(ns, 'video')
    m.eval()
    conv = m.encoder.layer1[0].conv1
    x = torch.from_numpy(np.random.RandomState(5).normal(size=(1, 64, 8, 8)).astype(np.float32))
    with torch.no_grad():
        y1 = conv(x)
        out['sn/u1'] = conv.module.weight_u.numpy().copy()
        out['sn/v1'] = conv.module.weight_v.numpy().copy()
        y2 = conv(x)
        out['sn/u2'] = conv.module.weight_u.numpy().copy()
        out['sn/y1'] = y1.numpy()
        out['sn/y2'] = y2.numpy()
    # encoder / aspp / decoder dense / IMD, eval and train(batch-stat) mode, 64x64, 2 frames
    for mode in ('eval', 'train'):
        m = build(ns, 'image')
        m.train(mode == 'train')
        batch = synth.synthetic_batch(2, 1, 2, 64, 64, seed=DSEED, train=True, max_inst=10)
        with torch.no_grad():
            masks, alphas, trans_gt, b, n_f, h, w, n_i, chosen, emb, mid = m.forward_encoder(batch)
            out[mode + '/enc_embedding_aspp'] = emb.numpy()
            for i, f in enumerate(mid['shortcut']):
                out['%s/fea%d_sum' % (mode, i + 1)] = f.double().sum(dim=(2, 3)).numpy()
            out[mode + '/fea5'] = mid['shortcut'][4].numpy()
            x, masks5, valid, gt_masks, f1, f2, f3, image, h, w = m.decoder.os32_to_os8(emb, mid, b, n_f, n_i, masks, alphas)
            out[mode + '/os8_feat'] = x.numpy()
            x_os8, xf, queries, loss_max, _ = m.decoder.refine_OS8(x, masks5, use_mask_atten=False, gt_mask=gt_masks)
            out[mode + '/imd_logits'] = x_os8.numpy()
            out[mode + '/imd_out_feat'] = xf.numpy()
            out[mode + '/imd_tokens'] = queries.numpy()
            out[mode + '/imd_max_loss'] = np.float32(float(loss_max))
    # ConvGRU 'bi' + bidirectional fusion (video decoder pieces)
    m = build(ns, 'video')
    m.eval()
    rs = np.random.RandomState(9)
    feat = torch.from_numpy(rs.normal(size=(1, 3, 128, 8, 8)).astype(np.float32))
    with torch.no_grad():
        o, hdn = m.decoder.os8_temp_module.propagate_features(feat.clone(), n_f=3, prev_h_state=None, temp_method='bi')
        out['gru/out'] = o.numpy()
        out['gru/hidden'] = hdn.numpy()
        f64 = torch.from_numpy(rs.normal(size=(1, 3, 64, 8, 8)).astype(np.float32))
        preds = torch.from_numpy(rs.uniform(size=(1, 3, 2, 64, 64)).astype(np.float32))
        df, db, fu = m.decoder.bidirectional_fusion(f64, preds)
        out['bifuse/df'] = df.numpy()
        out['bifuse/db'] = db.numpy()
        out['bifuse/fused'] = fu.numpy()
    # losses incl. the LapLoss channel quirk (loss.py:67-191, arch/maggie.py:237-266)
    a = torch.from_numpy(rs.uniform(size=(2, 3, 32, 32)).astype(np.float32))
    g = torch.from_numpy(rs.uniform(size=(2, 3, 32, 32)).astype(np.float32))
    wgt = torch.from_numpy((rs.uniform(size=(2, 3, 32, 32)) > 0.5).astype(np.float32))
    lap = ns.loss.LapLoss()
    grd = ns.loss.GradientLoss()
    out['loss/lap'] = np.float32(float(lap(a.view(-1, 1, 32, 32), g.view(-1, 1, 32, 32), wgt.view(-1, 1, 32, 32))))
    out['loss/grad'] = np.float32(float(grd(a, g, wgt)))
    out['loss/l1'] = np.float32(float(ns.MaGGIe.regression_loss(a, g, loss_type='l1', weight=wgt)))
    out['loss/dtssd'] = np.float32(float(ns.loss.loss_dtSSD(a.view(1, 2, 3, 32, 32), g.view(1, 2, 3, 32, 32), wgt.view(1, 2, 3, 32, 32))))
    np.savez_compressed(os.path.join(HERE, 'dense_pinned.npz'), **out)
    print('wrote dense_pinned.npz', len(out))


def layout_fixture(ns):
    """Checkpoint layout (parameter / buffer names, shapes, dtypes) of the reference models: the state_dict
    keys are an API (checkpoint/resume)."""
    import json
    for kind in ('image', 'video'):
        m = build(ns, kind)
        lay = {k: [list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in m.state_dict().items()}
        trainable = sorted(n for n, p in m.named_parameters() if p.requires_grad)
        with open(os.path.join(HERE, 'state_dict_layout_%s.json' % kind), 'w') as f:
            json.dump({'state_dict': lay, 'trainable': trainable}, f, indent=0, sort_keys=True)
        print('wrote layout', kind, len(lay))


def postprocess_fixture():
    """reverse_transform_tensor of the reference itself (maggie/utils/postprocessing.py) on seeded planes. Its module imports
    skimage / cv2 at the top (used by other functions only): empty stand-ins are enough to import it."""
    import importlib, sys, types
    for name in ('skimage', 'skimage.measure'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['skimage.measure'].label = None
    if 'maggie.utils.metric' not in sys.modules:                     # postprocessing.py only needs reshape2D from it
        m = types.ModuleType('maggie.utils.metric')
        m.reshape2D = lambda x: x.reshape(-1, *x.shape[-2:])
        sys.modules['maggie.utils.metric'] = m
    pp = importlib.import_module('maggie.utils.postprocessing')
    rs = np.random.RandomState(21)
    out = {}
    cases = {'resize_pad': ((2, 3, 40, 56), [{'name': ['resize'], 'ori_size': (torch.tensor(37), torch.tensor(61))},
                                            {'name': ['padding'], 'pad_size': (torch.tensor(5), torch.tensor(8))}]),
             'pad_resize_same': ((1, 2, 32, 48), [{'name': 'resize', 'ori_size': (29, 48)}, {'name': 'padding', 'pad_size': (3, 0)}]),
             'resize_only': ((3, 24, 24), [{'name': 'resize', 'ori_size': (50, 33)}])}
    for key, (shape, info) in cases.items():
        x = torch.from_numpy(rs.uniform(-0.05, 1.05, size=shape).astype(np.float32))
        y = pp.reverse_transform_tensor(x, info).numpy()
        out[key] = y
    np.savez_compressed(os.path.join(HERE, 'postprocess_pinned.npz'), **out)
    print('wrote postprocess_pinned.npz', {k: v.shape for k, v in out.items()})


def preprocess_fixture():
    """Input side: the reference's ToTensor / Normalize classes (maggie/dataloader/transforms.py:720-778) and the item-assembly
    statements of HIMDataset.__getitem__ (maggie/dataloader/him.py:157-173, executed here statement by statement with stock torch)
    on seeded uint8 inputs. transforms.py imports cv2 / albumentations / imgaug / skimage at the top for OTHER classes: empty
    stand-ins are enough to import it (nothing on this data path touches them)."""
    import importlib, sys, types
    import torch.nn.functional as F
    for name in ('cv2', 'albumentations', 'imgaug', 'imgaug.augmenters', 'imgaug.parameters', 'skimage', 'skimage.exposure'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['imgaug'].augmenters, sys.modules['imgaug'].parameters = sys.modules['imgaug.augmenters'], sys.modules['imgaug.parameters']
    sys.modules['skimage'].exposure = sys.modules['skimage.exposure']
    if 'maggie.dataloader' not in sys.modules:
        dl = types.ModuleType('maggie.dataloader')
        dl.__path__ = [os.path.join(ref_loader.REF_ROOT, 'maggie', 'dataloader')]
        sys.modules['maggie.dataloader'] = dl
    T = importlib.import_module('maggie.dataloader.transforms')
    out = {}
    for key, (n_f, n_i, H, W, max_inst, seed) in {'image_train': (1, 3, 64, 96, 10, 5), 'video_eval': (3, 2, 40, 56, None, 6),
                                                  'odd_size': (1, 1, 36, 52, 10, 7)}.items():
        rs = np.random.RandomState(seed)
        frames = rs.randint(0, 256, size=(n_f, H, W, 3)).astype(np.uint8)
        alphas = rs.randint(0, 256, size=(n_f * n_i, H, W)).astype(np.uint8)
        alphas[rs.rand(*alphas.shape) < 0.3] = rs.randint(0, 8)                    # exercise the `< 5 -> 0` rule
        masks = (rs.rand(n_f * n_i, H, W) < 0.5).astype(np.uint8) * 255
        d = {'frames': frames.copy(), 'alphas': alphas.copy(), 'masks': masks.copy()}
        d = T.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])(T.ToTensor()(d))
        image, alpha, mask = d['frames'], d['alphas'], d['masks']
        alpha = alpha * 1.0 / 255                                                # him.py:157
        mask = mask * 1.0 / 255                                                  # him.py:158
        if max_inst is not None:                                                 # him.py:159-167 (is_train)
            chosen_ids = rs.choice(range(max_inst), alpha.shape[1], replace=False)
            new_alpha = torch.zeros(alpha.shape[0], max_inst, *alpha.shape[2:])
            new_mask = torch.zeros(alpha.shape[0], max_inst, *mask.shape[2:])
            new_alpha[:, chosen_ids] = alpha.float()
            new_mask[:, chosen_ids] = mask.float()
            mask, alpha = new_mask, new_alpha
            out[key + '.slot_ids'] = np.asarray(chosen_ids, np.int64)
        mask = F.interpolate(mask.float(), size=(image.shape[2] // 8, image.shape[3] // 8), mode='nearest')     # him.py:172-173
        out[key + '.image'], out[key + '.alpha'], out[key + '.mask'] = image.numpy(), alpha.float().numpy(), mask.float().numpy()
    np.savez_compressed(os.path.join(HERE, 'preprocess_pinned.npz'), **out)
    print('wrote preprocess_pinned.npz', {k: v.shape for k, v in out.items()})


def metric_fixture():
    """The reference's own metric classes (maggie/utils/metric.py: SAD, MSE, MAD, Grad on CPU, dtSSD) on seeded planes. The module
    imports cv2 / skimage.measure at the top for Conn / MESSDdt only: empty stand-ins are enough to import it."""
    import importlib, sys, types
    for name in ('cv2', 'skimage', 'skimage.measure'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['skimage'].measure = sys.modules['skimage.measure']
    stub = sys.modules.get('maggie.utils.metric')
    if stub is not None and not hasattr(stub, 'SAD'):                 # postprocess_fixture's reshape2D-only stand-in
        del sys.modules['maggie.utils.metric']
    M = importlib.import_module('maggie.utils.metric')
    out = {}
    for key, (shape, seed, with_tri) in {'image': ((2, 3, 48, 40), 31, True), 'no_trimap': ((1, 2, 33, 57), 32, False),
                                         'clip': ((4, 2, 32, 32), 33, True)}.items():
        rs = np.random.RandomState(seed)
        pred = rs.rand(*shape).astype(np.float32)
        gt = np.clip(pred + rs.normal(0, 0.1, size=shape), 0, 1).astype(np.float32)
        tri = rs.randint(0, 3, size=shape).astype(np.float32) if with_tri else None
        for name in ('SAD', 'MSE', 'MAD', 'Grad'):
            m = getattr(M, name)()
            r = m.update(pred, gt, tri, device='cpu')
            out['%s.%s' % (key, name)] = np.asarray([r, m.score, m.count, m.average()], np.float64)
        m = M.dtSSD()
        r = m.update(pred, gt, tri)
        out['%s.dtSSD' % key] = np.asarray([r, m.score, m.count, m.average()], np.float64)
    np.savez_compressed(os.path.join(HERE, 'metric_pinned.npz'), **out)
    print('wrote metric_pinned.npz', {k: v.tolist() for k, v in out.items()})


def main():
    ns = ref_loader.load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == '--new-geometries':          # only the round-2 additions (the others are unchanged)
        return new_geometry_fixtures(ns)
    postprocess_fixture()
    preprocess_fixture()
    metric_fixture()
    layout_fixture(ns)
    dense_fixture(ns)
    model_fixture(ns, 'image', False, 1, 1, 2, 128, 0, None, 'model_image_eval.npz')
    model_fixture(ns, 'image', True, 2, 1, 2, 128, 10000, 10, 'model_image_train.npz')
    model_fixture(ns, 'image', True, 2, 1, 2, 128, 100, None, 'model_image_train_warmup.npz')
    model_fixture(ns, 'video', False, 1, 3, 2, 128, 0, None, 'model_video_eval.npz')
    model_fixture(ns, 'video', True, 1, 3, 2, 128, 10000, 10, 'model_video_train.npz')
    new_geometry_fixtures(ns)


def new_geometry_fixtures(ns):
    # BASELINE configs[2] geometry (image, 4 instances, batch 4 per GPU) and configs[4] geometry (video, T = 5, 3 instances), small sizes
    keep = ('refined_masks', 'alpha_os8', 'detail_mask')
    model_fixture(ns, 'image', False, 1, 1, 4, 128, 0, None, 'model_image_eval_4inst.npz', keep)
    model_fixture(ns, 'image', True, 4, 1, 4, 128, 10000, 10, 'model_image_train_4inst_b4.npz', keep)
    model_fixture(ns, 'video', False, 1, 5, 3, (96, 128), 0, None, 'model_video_eval_t5.npz', keep + ('temp_alpha',))
    model_fixture(ns, 'video', True, 1, 5, 3, 96, 10000, 10, 'model_video_train_t5.npz', keep)      # square: the reference's LapLoss upsample (loss.py:138) breaks on H != W


if __name__ == '__main__':
    main()


