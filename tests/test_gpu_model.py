"""End-to-end parity of the MI355X build (maggie_amd.network, HIP kernels through the C ABI) against the CPU oracle
(oracle/refmodel.py) and the committed golden fixtures (tests/golden/*.npz, produced by the reference's own modules).

Tolerances (north_star): fp32 alpha mattes within 1e-3 max-abs of the CPU path; the active-pixel index map
(`detail_mask`) bit-exact. bf16 (autocast) runs are checked against looser bounds and never for bit-exact masks."""
import copy

import numpy as np
import pytest
import torch

from helpers import seed_all, load_golden, unpack_bits, reference_layout_state_dict, model_cfg, record, WSEED, DSEED, RSEED

pytestmark = pytest.mark.gpu

ALPHA_TOL = 1e-3
# Train-mode fixtures at 96-128 px (batch statistics over 2-32 samples per channel in the deep layers): bars at <= 2x the distance this build
# measures against the fp32 CPU oracle -- one number per fixture, the step is bit-reproducible (profiles/r05_parity_observed.json).
# os8_max: max-abs error of the coarse alpha (no index-map dependence); frac: worst fraction of pixels of any alpha output beyond the 1e-3
# north-star tolerance (a site that flips in the index map switches its OS1 / OS4 pixel between refined and coarse); mism: fraction of index-map pixels that differ; loss_rel: worst relative loss-term error
# (incl. the reference-pinned entries); grad_med / grad_worst: per-parameter relative L2 gradient error; running: worst running-statistic error
# relative to the buffer's scale.
EVAL_IMAGE_MAX = 2.3e-5        # max-abs alpha error, image eval fixtures, fp32 (vs oracle and vs golden): measured 1.13e-5 (alpha_os8 vs golden, 4 instances)
# measured (round 5, profiles/r05_parity_observed.json): alpha_os8 vs fp64 6.7e-4 (the CPU fp32 path: 2.8e-4), one index-map pixel of 2.6 M, loss 8.1e-6,
# gradients 2.7e-3 / 8.9e-3 / 1.2e-2 (CPU fp32: 1.4e-3 / 6.8e-3 / 1.15e-2)
WELL_BARS = dict(alpha_os8_vs_fp64=1e-3, mism=8e-7, loss_rel=2e-5, grad_med=5.4e-3, grad_p90=1.8e-2, grad_worst=2.4e-2)
# measured per fixture (same file); `frac` / `mism` floors of 2e-5 / 1e-5 = a few pixels where the measurement is 0
TRAIN_BARS = {
    'model_image_train.npz': dict(os8_max=2.9e-3, frac=8.2e-4, mism=2.5e-5, loss_rel=1.9e-4, grad_med=1.1e-2, grad_worst=4e-2, running=1.1e-4),
    'model_image_train_warmup.npz': dict(os8_max=6e-4, frac=2e-5, mism=1e-5, loss_rel=1.6e-5, grad_med=3.6e-3, grad_worst=4e-2, running=1.6e-5),
    'model_video_train.npz': dict(os8_max=8.3e-4, frac=2e-4, mism=1e-5, loss_rel=3e-5, grad_med=4.6e-3, grad_worst=2.5e-2, running=6.2e-5),
    'model_image_train_4inst_b4.npz': dict(os8_max=1.1e-3, frac=8e-5, mism=1e-5, loss_rel=4e-5, grad_med=7.6e-3, grad_worst=3.4e-2, running=4e-5),
    'model_video_train_t5.npz': dict(os8_max=2.8e-4, frac=2e-5, mism=1e-5, loss_rel=1.2e-4, grad_med=3.1e-3, grad_worst=2e-2, running=1.4e-5),
}


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def _build(kind, dev, train):
    from maggie_amd.network import build_model
    from maggie_amd.utils import config
    model, _ = build_model(config.model_config(kind))
    sd = reference_layout_state_dict(kind)
    model.load_state_dict(sd)
    model.to(dev).train(train)
    return model, sd


def _to(batch, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}


@pytest.mark.parametrize('kind,b,n_f,n_inst,h,w,gname', [
    ('image', 1, 1, 2, 128, 128, 'model_image_eval.npz'),
    ('video', 1, 3, 2, 128, 128, 'model_video_eval.npz'),
    ('image', 1, 1, 4, 128, 128, 'model_image_eval_4inst.npz'),        # BASELINE configs[2] geometry: 4 instances
    ('video', 1, 5, 3, 96, 128, 'model_video_eval_t5.npz'),            # BASELINE configs[4] geometry: T = 5, 3 instances
])
def test_eval_forward_matches_oracle_and_golden(kind, b, n_f, n_inst, h, w, gname):
    from maggie_amd.utils import synth
    from oracle import refmodel
    dev = _dev()
    model, sd = _build(kind, dev, False)
    batch = synth.synthetic_batch(b, n_f, n_inst, h, w, seed=DSEED, train=False)
    with torch.no_grad():
        out = model(_to(batch, dev))
        ref = refmodel.maggie_forward({k: v.clone() for k, v in sd.items()}, model_cfg(kind), batch, False)
    gold = load_golden(gname)
    # The video decoder snaps alpha_os8 >= 0.95 to 1.0 (resnet_inst_matt_spconv_temp.py:115-117): a float discontinuity, so
    # pixels whose pre-threshold value sits within rounding distance of 0.95 may legitimately land on the other side.
    # One such pixel also moves one active site, which the stacked 3x3 sparse convs spread over its ~5x5 neighbourhood.
    # Both effects are bounded to a 5e-4 fraction of pixels; everything else must meet the 1e-3 bar.
    flip_budget = 5e-4 if kind == 'video' else 0.0
    obs = {}
    for k in ('alpha_os8', 'alpha_os4', 'alpha_os1', 'refined_masks'):
        o = out[k].float().cpu()
        assert o.shape == ref[k].shape == (b, n_f, n_inst, h, w)
        diff = (o - ref[k]).abs()
        frac_bad = float((diff > ALPHA_TOL).float().mean())
        d_gd = np.abs(o.numpy() - gold['out/' + k]) if 'out/' + k in gold.files else np.zeros(1)
        print(kind, k, 'vs oracle max %.3g (frac>tol %.2e)  vs golden max %.3g (frac>tol %.2e)' % (
            diff.max().item(), frac_bad, d_gd.max(), float((d_gd > ALPHA_TOL).mean())))
        assert frac_bad <= flip_budget and float((d_gd > ALPHA_TOL).mean()) <= flip_budget, k
        obs['max_' + k] = diff.max().item()
        obs['max_vs_golden_' + k] = float(d_gd.max())
    record('eval/' + gname, **obs)
    if kind == 'image':
        # image eval: no float discontinuity on the path -> one reproducible distance per fixture, bar at <= 2x of it (far inside the 1e-3 north star)
        assert max(obs.values()) <= EVAL_IMAGE_MAX, obs
    dm = out['detail_mask'].cpu().numpy()
    mism = float((dm != ref['detail_mask'].numpy()).mean())
    print(kind, 'detail_mask mismatch fraction', mism)
    if kind == 'image':
        assert np.array_equal(dm, ref['detail_mask'].numpy()), 'active-pixel index map must be bit-exact'
        assert np.array_equal(dm.reshape(-1), unpack_bits(gold['out/detail_mask'], dm.shape).reshape(-1))
    else:
        assert mism <= 1e-3
    if kind == 'video':
        for k in ('diff_pred_forward', 'diff_pred_backward', 'temp_alpha'):
            assert float(((out[k].float().cpu() - ref[k]).abs() > ALPHA_TOL).float().mean()) <= flip_budget, k
    # SpectralNorm state advanced exactly one power iteration (spectral_norm.py:73-80)
    u = model.state_dict()['encoder.conv1.module.weight_u'].cpu().numpy()
    assert np.abs(u - gold['sn/encoder.conv1.module.weight_u']).max() < 1e-5


@pytest.mark.parametrize('kind,b,n_f,n_inst,hw,it,max_inst,gname', [
    ('image', 2, 1, 2, 128, 10000, 10, 'model_image_train.npz'),
    ('image', 2, 1, 2, 128, 100, None, 'model_image_train_warmup.npz'),
    ('video', 1, 3, 2, 128, 10000, 10, 'model_video_train.npz'),
    ('image', 4, 1, 4, 128, 10000, 10, 'model_image_train_4inst_b4.npz'),       # BASELINE configs[2] geometry: 4 instances, batch 4
    ('video', 1, 5, 3, 96, 10000, 10, 'model_video_train_t5.npz'),              # BASELINE configs[4] geometry: T = 5, 3 instances
])
def test_train_step_matches_oracle_and_golden(kind, b, n_f, n_inst, hw, it, max_inst, gname):
    from maggie_amd.utils import synth
    from oracle import refmodel
    dev = _dev()
    model, _ = _build(kind, dev, True)
    model.decoder.inst_spec_layer.dropout.p = 0.0            # dropout masks are device-RNG dependent
    sd = reference_layout_state_dict(kind, requires_grad=True)
    batch = synth.synthetic_batch(b, n_f, n_inst, hw, hw, seed=DSEED, train=True, it=it, max_inst=max_inst)
    seed_all(RSEED)
    out, loss = model(_to(batch, dev))
    loss['total'].backward()
    seed_all(RSEED)
    # oracle with dropout disabled the same way
    import oracle.refmodel as rm
    orig = rm.predict_details
    rm.predict_details = lambda *a, **kw: orig(*a, **{**kw, 'drop_p': 0.0})
    try:
        ref, rloss = refmodel.maggie_forward(sd, model_cfg(kind), batch, True)
    finally:
        rm.predict_details = orig
    rloss['total'].backward()
    # Train mode normalises with BATCH statistics; at this test size the deepest layers see 8-32 samples per channel and the ASPP pooled branch 2,
    # which amplifies fp32 rounding differences between GPU and CPU (ill-conditioned 1/std). The step is bit-reproducible (MAGGIE_DETERMINISTIC,
    # tests/test_gpu_determinism.py), so each fixture has ONE HIP-vs-oracle distance: the bars below are <= 2x what this build measures
    # (TRAIN_BARS; observed values are written to gpurun_out/parity_observed.json, copied to profiles/ per round). The north-star 1e-3 bar is
    # enforced in eval mode and by the well-conditioned train test below.
    bars = TRAIN_BARS[gname]
    obs = {}
    for k in ('alpha_os8', 'alpha_os4', 'alpha_os1', 'refined_masks'):
        diff = (out[k].float().cpu() - ref[k].detach()).abs()
        obs['max_' + k] = diff.max().item()
        obs['frac_gt_1e-3_' + k] = float((diff > ALPHA_TOL).float().mean())
        print(kind, it, k, 'vs oracle max %.3g frac>1e-3 %.2e' % (obs['max_' + k], obs['frac_gt_1e-3_' + k]))
    mism = float((out['detail_mask'].cpu() != ref['detail_mask']).float().mean())
    obs['detail_mask_mismatch'] = mism
    print(kind, it, 'detail_mask mismatch fraction %.2e' % mism)
    gold = load_golden(gname)
    obs['loss_rel'] = 0.0
    for k, v in rloss.items():
        a, r = float(loss[k]), float(v)
        print('  loss', k, a, r)
        obs['loss_rel'] = max(obs['loss_rel'], abs(a - r) / max(1.0, abs(r)))
        if it >= 3000 and k in ('loss_rec_os8', 'loss_lap_os8', 'loss_grad_os8', 'loss_max_atten'):      # pinned entries
            obs['loss_rel'] = max(obs['loss_rel'], abs(a - float(gold['loss/' + k])) / max(1.0, abs(r)))
    # gradients: relative L2 error per parameter against the oracle's autograd
    worst = (0.0, None)
    errs = []
    n_checked = 0
    for n, p in model.named_parameters():
        g_ref = sd[n].grad
        if p.grad is None:
            assert g_ref is None or float(g_ref.abs().max()) == 0.0, n
            continue
        if g_ref is None:
            continue
        g = p.grad.float().cpu()
        scale = float(g_ref.norm()) + 1e-8
        err = float((g - g_ref).norm()) / scale
        n_checked += 1
        errs.append((err, n, scale))
        if scale > 1e-5 and err > worst[0]:
            worst = (err, n)
    errs.sort(reverse=True)
    print('  params checked', n_checked, 'worst rel grad errs', [(round(e, 4), n, '%.2e' % sc) for e, n, sc in errs[:8]])
    assert n_checked >= 290
    med = sorted(e for e, _, sc in errs if sc > 1e-5)[len(errs) // 2]
    print('  median rel grad err', med, 'worst', worst)
    obs['grad_rel_median'], obs['grad_rel_worst'] = med, worst[0]
    # running statistics were updated like the reference's BatchNorm
    msd = model.state_dict()
    assert np.abs(msd['encoder.bn1.running_mean'].cpu().numpy() - gold['bn/encoder.bn1.running_mean']).max() < 1e-4
    # ... and EVERY buffer the step leaves behind agrees with the oracle's (which updated `sd` in place like the reference's modules do): all
    # running_mean / running_var / num_batches_tracked (incl. the 9 sparse BatchNorm1d over active rows and the decoder's skip BatchNorm, which
    # this build evaluates before the up-sampling: same mean / biased variance, 4 samples per row in the unbiased factor), all SpectralNorm u / v.
    # Bars: relative to the buffer's own scale -- batch statistics over 2-32 samples per channel carry the same conditioning as the outputs above.
    checked = {'running_mean': 0, 'running_var': 0, 'num_batches_tracked': 0, 'weight_u': 0, 'weight_v': 0}
    worst_b = {}
    for n, v in msd.items():
        kind_ = n.rsplit('.', 1)[-1]
        if kind_ not in checked or n.startswith('decoder.dummy_downscale'):
            continue
        r = sd[n].detach()
        a = v.detach().cpu()
        checked[kind_] += 1
        if kind_ == 'num_batches_tracked':
            assert int(a) == int(r), (n, int(a), int(r))
            continue
        scale_ = max(float(r.abs().max()), 1e-3)
        err = float((a.float() - r.float()).abs().max()) / scale_
        if err > worst_b.get(kind_, (0.0, None))[0]:
            worst_b[kind_] = (err, n)
    print('  buffers checked', checked, 'worst', {k: (round(e, 6), n) for k, (e, n) in worst_b.items()})
    assert checked['running_mean'] >= 71 and checked['running_var'] >= 71 and checked['weight_u'] >= 54 and checked['weight_v'] >= 54, checked
    assert worst_b.get('weight_u', (0.0, None))[0] <= 1e-5 and worst_b.get('weight_v', (0.0, None))[0] <= 1e-5, worst_b
    obs['running_mean_rel'], obs['running_var_rel'] = worst_b.get('running_mean', (0.0, None))[0], worst_b.get('running_var', (0.0, None))[0]
    record('train_step/' + gname, **obs)
    # every bar: <= 2x the (single, reproducible) distance this build measures on this fixture
    frac = max(obs['frac_gt_1e-3_' + k] for k in ('alpha_os8', 'alpha_os4', 'alpha_os1', 'refined_masks'))
    assert obs['max_alpha_os8'] <= bars['os8_max'] and frac <= bars['frac'], (obs, bars)
    assert mism <= bars['mism'], (mism, bars)
    assert obs['loss_rel'] <= bars['loss_rel'], (obs['loss_rel'], bars)
    assert med <= bars['grad_med'] and worst[0] <= bars['grad_worst'], (med, worst, bars)
    assert obs['running_mean_rel'] <= bars['running'] and obs['running_var_rel'] <= bars['running'], (worst_b, bars)


def _oracle_train_step(kind, batch, dtype):
    """One oracle train step (dropout off) with every floating-point tensor in `dtype`: fp32 = the reference's CPU path, fp64 = the
    yardstick (same algorithm, rounding removed)."""
    import oracle.refmodel as rm
    torch.set_default_dtype(dtype)
    try:
        sd = reference_layout_state_dict(kind)
        sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(('_u', '_v', 'running_mean', 'running_var')):
                v.requires_grad_(True)
        bt = {k: (v.to(dtype) if torch.is_tensor(v) else v) for k, v in batch.items()}
        orig = rm.predict_details
        rm.predict_details = lambda *a, **kw: orig(*a, **{**kw, 'drop_p': 0.0})
        try:
            seed_all(RSEED)
            out, loss = rm.maggie_forward(sd, model_cfg(kind), bt, True)
            loss['total'].backward()
        finally:
            rm.predict_details = orig
    finally:
        torch.set_default_dtype(torch.float32)
    return out, loss, sd


def _grad_errs(get_grad, sd_true):
    errs = []
    for n, t in sd_true.items():
        g_true = t.grad
        g = get_grad(n)
        if g is None or g_true is None or float(g_true.norm()) <= 1e-5:
            continue
        errs.append(float((g.double() - g_true.double()).norm() / g_true.double().norm()))
    errs.sort()
    return errs


def test_train_step_well_conditioned_batch_at_the_fp32_noise_floor():
    """Train-mode parity at a batch where batch-statistic BatchNorm is well conditioned: 16 frames of 128x128 -> the deepest layers
    (OS32, 4x4 per frame) see 256 samples per channel, the ASPP pooled branch 16. The yardstick is the oracle run in FLOAT64: the
    reference's own fp32 CPU path sits 3e-4 (alpha_os8) / 2.7e-3 median, 8e-3 p90 (per-parameter gradients) away from it -- batch
    statistics through ~70 normalisation layers amplify fp32 rounding, and the OS8 loss weights are thresholded predictions. The HIP
    fp32 path must sit at the same noise floor as the reference's fp32 path (its run-to-run spread is bounded by 6x the CPU median, 2.5x on
    the p90 / worst gradient quantiles, 3x on the alpha max-abs; plus: alpha within the 1e-3
    north-star bar, losses within 1e-3 relative of the fp32 oracle). Round 5: the HIP step is bit-reproducible, so the "spread" is one number;
    the bars are now absolute, at <= 2x what this build measures (WELL_BARS)."""
    from maggie_amd.utils import synth
    dev = _dev()
    model, _ = _build('image', dev, True)
    model.decoder.inst_spec_layer.dropout.p = 0.0
    batch = synth.synthetic_batch(16, 1, 2, 128, 128, seed=DSEED, train=True, it=10000, max_inst=10)
    seed_all(RSEED)
    out, loss = model(_to(batch, dev))
    loss['total'].backward()
    ref, rloss, sd32 = _oracle_train_step('image', batch, torch.float32)
    tru, tloss, sd64 = _oracle_train_step('image', batch, torch.float64)
    for k in ('alpha_os8', 'alpha_os4', 'alpha_os1', 'refined_masks'):
        diff = (out[k].float().cpu() - ref[k].detach()).abs()
        frac = float((diff > ALPHA_TOL).float().mean())
        print(k, 'vs fp32 oracle max %.3g frac>1e-3 %.2e' % (diff.max().item(), frac))
        assert frac <= (0.0 if k == 'alpha_os8' else 5e-4), k            # a threshold flip of the index map moves a handful of sites
    e_gpu = float((out['alpha_os8'].double().cpu() - tru['alpha_os8'].detach()).abs().max())
    e_cpu = float((ref['alpha_os8'].detach().double() - tru['alpha_os8'].detach()).abs().max())
    print('alpha_os8 max-abs vs fp64: HIP fp32 %.3g, CPU fp32 %.3g' % (e_gpu, e_cpu))
    mism = float((out['detail_mask'].cpu() != ref['detail_mask']).float().mean())
    print('detail_mask mismatch fraction %.2e' % mism)
    loss_rel = max(abs(float(loss[k]) - float(v)) / max(1.0, abs(float(v))) for k, v in rloss.items())
    grads = {n: p.grad for n, p in model.named_parameters()}
    g_gpu = _grad_errs(lambda n: None if grads.get(n) is None else grads[n].cpu(), sd64)
    g_cpu = _grad_errs(lambda n: sd32[n].grad, sd64)
    q = lambda e, f: e[min(int(f * len(e)), len(e) - 1)]          # noqa: E731
    print('per-parameter gradient error vs fp64 (median / p90 / worst): HIP fp32 %.3g / %.3g / %.3g   CPU fp32 %.3g / %.3g / %.3g' % (
        q(g_gpu, .5), q(g_gpu, .9), g_gpu[-1], q(g_cpu, .5), q(g_cpu, .9), g_cpu[-1]))
    record('train_step_well_conditioned', alpha_os8_vs_fp64_hip=e_gpu, alpha_os8_vs_fp64_cpu=e_cpu, detail_mask_mismatch=mism, loss_rel=loss_rel,
           grad_hip_median=q(g_gpu, .5), grad_hip_p90=q(g_gpu, .9), grad_hip_worst=g_gpu[-1],
           grad_cpu_median=q(g_cpu, .5), grad_cpu_p90=q(g_cpu, .9), grad_cpu_worst=g_cpu[-1])
    assert len(g_gpu) >= 280
    # Both paths are fp32 rounding noise amplified by ~70 batch-statistic normalisations and thresholded loss weights; the CPU path is one
    # deterministic draw of it, and -- since round 4 -- so is the HIP path (bit-reproducible steps): the bars are <= 2x the single values this
    # build measures (WELL_BARS), and the coarse alpha must stay inside the 1e-3 north-star tolerance of the EXACT (fp64) answer.
    assert e_gpu <= WELL_BARS['alpha_os8_vs_fp64'] and e_gpu <= ALPHA_TOL, e_gpu
    assert mism <= WELL_BARS['mism'], mism
    assert loss_rel <= WELL_BARS['loss_rel'], loss_rel
    assert q(g_gpu, .5) <= WELL_BARS['grad_med'] and q(g_gpu, .9) <= WELL_BARS['grad_p90'] and g_gpu[-1] <= WELL_BARS['grad_worst'], (
        q(g_gpu, .5), q(g_gpu, .9), g_gpu[-1])


@pytest.mark.parametrize('fmt', ['pth_ddp_prefix_old_spconv_layout', 'safetensors', 'hub_snapshot_dir'])
def test_checkpoint_bridge_reference_format_file_to_hip_eval_matches_oracle(tmp_path, fmt):
    """SURVEY 8(f)3 on the GPU (the bridge had host tests only): a reference-format checkpoint FILE -- `.pth` as
    maggie/engine/train.py:324 writes it from a DistributedDataParallel model (`module.` prefix) with the sparse-conv weights in spconv < 2.2's
    (kh, kw, Cin, Cout) layout, `.safetensors`, and a hub snapshot directory (maggie/network/__init__.py:9) -- is loaded through
    maggie_amd.utils.checkpoint into a freshly built model, and the HIP eval forward equals the oracle run on the tensors that were written:
    alpha within 1e-3, index map bit-exact. The weights are NOT the ones every other test uses (seed WSEED + 5): a load that silently kept the
    model's own initialisation, or mis-laid one sparse weight, cannot pass."""
    from maggie_amd.network import build_model
    from maggie_amd.utils import checkpoint, config, synth
    from oracle import refmodel
    dev = _dev()
    sd = reference_layout_state_dict('image', seed=WSEED + 5)
    model, _ = build_model(config.model_config('image'))
    sparse = checkpoint._sparse_weight_names(model)
    assert len(sparse) >= 15
    if fmt == 'pth_ddp_prefix_old_spconv_layout':
        path = str(tmp_path / 'best_model.pth')
        torch.save({'module.' + k: (v.permute(1, 2, 3, 0).contiguous() if k in sparse else v.clone()) for k, v in sd.items()}, path)
    elif fmt == 'safetensors':
        from safetensors.torch import save_file
        path = str(tmp_path / 'model.safetensors')
        save_file({k: v.contiguous() for k, v in sd.items()}, path)
    else:
        from safetensors.torch import save_file
        (tmp_path / 'snapshot').mkdir()
        save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / 'snapshot' / 'model.safetensors'))
        path = str(tmp_path / 'snapshot')
    missing, unexpected, mismatch = checkpoint.load_pretrained(model, path, strict=True)
    assert not missing and not unexpected and not mismatch
    model.to(dev).eval()
    batch = synth.synthetic_batch(1, 1, 3, 128, 96, seed=DSEED + 2, train=False)
    with torch.no_grad():
        out = model(_to(batch, dev))
        ref = refmodel.maggie_forward({k: v.clone() for k, v in sd.items()}, model_cfg('image'), batch, False)
        ref_default = refmodel.maggie_forward(reference_layout_state_dict('image'), model_cfg('image'), batch, False)
    worst = 0.0
    for k in ('alpha_os8', 'alpha_os4', 'alpha_os1', 'refined_masks'):
        worst = max(worst, float((out[k].float().cpu() - ref[k]).abs().max()))
    record('checkpoint_bridge/' + fmt, alpha_max=worst)
    assert worst <= ALPHA_TOL, worst
    assert np.array_equal(out['detail_mask'].cpu().numpy(), ref['detail_mask'].numpy())
    assert float((ref['alpha_os8'] - ref_default['alpha_os8']).abs().max()) > 1e-2            # these weights really are different ones
    # ... and back: what this build saves from the device model is what the reference's loader reads (same keys, shapes, values)
    back = str(tmp_path / 'saved.pth')
    checkpoint.save_model(model, back)
    sd_back = torch.load(back, map_location='cpu', weights_only=True)
    assert set(sd_back) == set(sd)
    for k, v in sd.items():
        if k.endswith(('weight_u', 'weight_v')):
            continue                                           # SpectralNorm advanced one power iteration during the forward (spectral_norm.py:73-80)
        assert sd_back[k].shape == v.shape and torch.equal(sd_back[k], v), k


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_dense_path_matches_reference_pinned(mode):
    """PINNED parity: encoder + ASPP + decoder OS32->OS8 + InstanceMatteDecoder against outputs of the REFERENCE's own
    modules (tests/golden/dense_pinned.npz, no third-party stand-in on that path), eval and train (batch-stat BN)."""
    from maggie_amd.utils import synth
    dev = _dev()
    gold = load_golden('dense_pinned.npz')
    model, _ = _build('image', dev, mode == 'train')
    batch = _to(synth.synthetic_batch(2, 1, 2, 64, 64, seed=DSEED, train=True, max_inst=10), dev)
    nchw = lambda t: t.float().permute(0, 3, 1, 2).cpu().numpy()
    with torch.no_grad():
        masks, alphas, trans_gt, b, n_f, h, w, n_i, chosen, emb, mid = model.forward_encoder(batch)
        rep = {}
        rep['enc_embedding_aspp'] = (nchw(emb), gold[mode + '/enc_embedding_aspp'])
        for i, f in enumerate(mid['shortcut']):
            rep['fea%d_sum' % (i + 1)] = (f.double().sum((1, 2)).cpu().numpy(), gold['%s/fea%d_sum' % (mode, i + 1)])
        rep['fea5'] = (nchw(mid['shortcut'][4]), gold[mode + '/fea5'])
        x, masks5, valid, gt_masks, f1, f2, f3, image, h, w = model.decoder.os32_to_os8(emb, mid, b, n_f, n_i, masks, alphas)
        rep['os8_feat'] = (nchw(x), gold[mode + '/os8_feat'])
        logits, xf, queries, loss_max, _ = model.decoder.refine_OS8(x, masks5, use_mask_atten=False, gt_mask=gt_masks)
        rep['imd_logits'] = (nchw(logits)[:, :10], gold[mode + '/imd_logits'])
        rep['imd_out_feat'] = (nchw(xf), gold[mode + '/imd_out_feat'])
        rep['imd_tokens'] = (queries.float().cpu().numpy(), gold[mode + '/imd_tokens'])
        if mode == 'train':
            rep['imd_max_loss'] = (np.array(float(loss_max)), gold[mode + '/imd_max_loss'])
    worst = 0
    for k, (a, r) in rep.items():
        rel = np.abs(a - r).max() / (np.abs(r).max() + 1e-12)
        print(mode, k, 'max abs %.3g rel-to-max %.3g' % (np.abs(a - r).max(), rel))
        worst = max(worst, rel)
    assert worst < (2e-5 if mode == 'eval' else 5e-4)


@pytest.mark.parametrize('kind,b,n_f,n_inst,h,w,mask_scale', [
    ('image', 1, 1, 1, 96, 160, 8),        # one instance, non-square, W not a multiple of 64 (ragged bit-plane words)
    ('image', 2, 1, 3, 64, 96, 8),         # batch of two different images, three instances
    ('image', 1, 1, 2, 128, 64, 1),        # guidance masks given at full resolution
    ('video', 1, 3, 1, 96, 128, 8),        # video window, one instance, non-square
])
def test_eval_edge_geometries_match_oracle(kind, b, n_f, n_inst, h, w, mask_scale):
    """Geometries the reference handles (arch/maggie.py:170-198: masks at 1/8 or full resolution, any instance count, any H x W that
    is a multiple of 32) against the CPU oracle: alpha within 1e-3, detail mask bit-exact (image)."""
    from maggie_amd.utils import synth
    from oracle import refmodel
    dev = _dev()
    model, sd = _build(kind, dev, False)
    batch = synth.synthetic_batch(b, n_f, n_inst, h, w, seed=DSEED + 1, train=False, mask_scale=mask_scale)
    with torch.no_grad():
        out = model(_to(batch, dev))
        ref = refmodel.maggie_forward({k: v.clone() for k, v in sd.items()}, model_cfg(kind), batch, False)
    budget = 1e-3 if kind == 'video' else 0.0          # the video decoder's >= 0.95 snap is a float discontinuity (see above)
    for k in ('alpha_os8', 'alpha_os4', 'alpha_os1', 'refined_masks'):
        o = out[k].float().cpu()
        assert o.shape == ref[k].shape == (b, n_f, n_inst, h, w)
        bad = float(((o - ref[k]).abs() > ALPHA_TOL).float().mean())
        assert bad <= budget, '%s: %.2e of pixels beyond %.0e (max %.3g)' % (k, bad, ALPHA_TOL, float((o - ref[k]).abs().max()))
    dm, rm = out['detail_mask'].cpu().numpy(), ref['detail_mask'].numpy()
    if kind == 'image':
        assert np.array_equal(dm, rm), 'active-pixel index map must be bit-exact'
    else:
        assert float((dm != rm).mean()) <= 2e-3


def test_empty_guidance_mask_raises_like_the_reference():
    """An instance whose guidance mask is empty poisons the attention with NaNs; the reference raises ValueError("Mask is empty")
    (mask_attention.py:95-98) -- so do the oracle and this build (eager and graphed)."""
    from maggie_amd.utils import synth
    from oracle import refmodel
    dev = _dev()
    model, sd = _build('image', dev, False)
    batch = synth.synthetic_batch(1, 1, 2, 64, 64, seed=DSEED, train=False)
    batch['mask'][:] = 0
    with torch.no_grad():
        with pytest.raises(ValueError, match='Mask is empty'):
            refmodel.maggie_forward({k: v.clone() for k, v in sd.items()}, model_cfg('image'), batch, False)
        for _ in range(3):                               # eager, capture step, replay
            with pytest.raises(ValueError, match='Mask is empty'):
                model(_to(batch, dev))


@pytest.mark.parametrize('n_f,n_inst,h,w', [(3, 2, 128, 128), (5, 3, 96, 128)])
def test_video_region_ops_bit_exact_given_the_same_coarse_alpha(n_f, n_inst, h, w):
    """end to end the video model's index map may differ from the oracle's at the `x_os8 >= 0.95 -> 1` float
    discontinuity (a 1e-6 difference in the coarse alpha flips a pixel). Here the region ops themselves are pinned: the product's OWN
    coarse alpha is handed to the oracle's restatement of resnet_inst_matt_spconv_temp.py:115-142 (snap, unknown band, smoothed bounding-box
    crop) and the product's `detail_mask` and cropped `alpha_os8` must then be identical bit for bit."""
    from maggie_amd.utils import synth
    from oracle import refmodel
    dev = _dev()
    model, _ = _build('video', dev, False)
    model.hip_graphs = False
    seen = {}
    orig = model.decoder.detail_stage

    def spy(dense, *a, **k):
        seen['x_os8'] = dense[0].detach().float().cpu().clone()
        return orig(dense, *a, **k)

    model.decoder.detail_stage = spy
    batch = synth.synthetic_batch(1, n_f, n_inst, h, w, seed=DSEED, train=False)
    with torch.no_grad():
        out = model(_to(batch, dev))
    x8 = seen['x_os8'][:, :n_inst]
    assert float(((x8 > 0.9) & (x8 < 1.0)).float().mean()) > 0 or float((x8 >= 0.95).float().mean()) > 0, 'the snap must have something to do'
    ref_a8, ref_mask = refmodel.video_eval_region(x8, n_inst, h, w)
    dm = out['detail_mask'].cpu().reshape(-1, n_inst, h, w)
    assert dm.float().sum() > 0
    assert torch.equal(dm.float(), ref_mask.float()), 'detail mask: %d pixels differ' % int((dm.float() != ref_mask.float()).sum())
    a8 = out['alpha_os8'].float().cpu().reshape(-1, n_inst, h, w)
    assert torch.equal(a8, ref_a8), 'cropped coarse alpha: max diff %g' % float((a8 - ref_a8).abs().max())


def test_video_consecutive_windows_with_prev_pred_match_oracle():
    """engine/test.py:219-224: the video model runs on overlapping 3-frame windows and hands frame t of each window's fused output to the
    next call as `prev_pred` (arch/maggie_temp.py:43-46). Two consecutive windows of a 4-frame clip through the product (HIP post-fusion,
    mg_temporal_fuse) and the oracle, each with its OWN previous output, then stitched by VideoWindow / the oracle's window bookkeeping."""
    from maggie_amd.utils import synth
    from maggie_amd.utils.video_window import VideoWindow
    from oracle import refmodel
    from oracle.video_window import Window
    dev = _dev()
    model, sd = _build('video', dev, False)
    n_i, h, w = 2, 128, 128
    clip = synth.synthetic_batch(1, 4, n_i, h, w, seed=DSEED + 1, train=True, motion=0.03)
    frames = lambda i: {k: v[:, i:i + 3].contiguous() for k, v in clip.items() if k in ('image', 'mask')}        # noqa: E731
    gts = [clip['alpha'][:, i:i + 3] for i in range(2)]
    tri = [clip['transition'][:, i:i + 3] for i in range(2)]
    names = [['f%d' % (i + j) for j in range(3)] for i in range(2)]
    ref_sd = {k: v.clone() for k, v in sd.items()}                     # ONE state for both oracle calls: SpectralNorm's u / v advance per forward
    ours, theirs = VideoWindow(), Window()
    prev = ref_prev = None
    stitched, ref_stitched = [], []
    for i in range(2):
        with torch.no_grad():
            out = model(_to(frames(i), dev), mem_feat=None, prev_pred=prev)
            ref = refmodel.maggie_forward(ref_sd, model_cfg('video'), frames(i), False, prev_pred=ref_prev)
        alpha, ref_alpha = out['refined_masks'].float().cpu(), ref['refined_masks']
        bad = float(((alpha - ref_alpha).abs() > ALPHA_TOL).float().mean())
        print('window', i, 'max diff %.3g, frac > tol %.2e' % (float((alpha - ref_alpha).abs().max()), bad))
        assert bad <= 5e-4, 'window %d' % i
        if i == 1:
            # the second window's frame 1 depends on prev_pred: it must differ from what the same window gives without it
            with torch.no_grad():
                ref_no_prev = refmodel.maggie_forward({k: v.clone() for k, v in ref_sd.items()}, model_cfg('video'), frames(i), False)
            assert float((ref_no_prev['refined_masks'][:, 1] - ref_alpha[:, 1]).abs().max()) > 0 or \
                float((ref_prev - ref_alpha[:, 0]).abs().max()) == 0, 'prev_pred had no effect in the oracle: the test would be vacuous'
        prev, ref_prev = out['refined_masks'][:, 1].cpu(), ref_alpha[:, 1].clone()       # engine/test.py:222
        o = ours.push(out['refined_masks'], gts[i].to(dev), tri[i].to(dev), names[i], i == 0, i == 1)
        r = theirs.push(ref_alpha.numpy(), gts[i].numpy(), tri[i].numpy(), names[i], i == 0, i == 1)
        assert list(o['save'][0]) == list(r['save'][0])
        stitched.append(o['save'][1].float().cpu())
        ref_stitched.append(torch.from_numpy(np.asarray(r['save'][1])))
    got, want = torch.cat(stitched, 1), torch.cat(ref_stitched, 1)
    assert got.shape == want.shape and got.shape[1] >= 4, got.shape                     # every frame of the clip reaches the saving callback
    assert float(((got - want).abs() > ALPHA_TOL).float().mean()) <= 5e-4


def test_fp16_autocast_runs_the_unchanged_harness_recipe_on_the_fp16_kernels():
    """VERDICT round 2 missing #2 / next #7: the reference's `--precision 16` recipe (engine/train.py:208,227-229,265-281: fp16 autocast,
    GradScaler.scale(loss).backward(), unscale_, clip 0.01, scaler.step, scaler.update) runs unchanged on the fp16 instantiations of the HIP
    kernels (IEEE half storage, v_mfma_f32_16x16x32_f16, fp32 accumulate). Held against an fp32 run of the same steps; fp16 has three mantissa
    bits more than bf16, so it must sit at least as close to fp32 as the bf16 family does. MAGGIE_FP16_AUTOCAST=bf16 still maps fp16 autocast
    onto the bf16 kernels (with a warning)."""
    import warnings
    from maggie_amd import functional as MF
    from maggie_amd.utils import synth
    dev = _dev()
    batch = _to(synth.synthetic_batch(2, 1, 2, 64, 64, seed=DSEED, train=True, max_inst=10, it=100), dev)

    def steps(dtype, scaler_on):
        model, _ = _build('image', dev, True)
        model.decoder.inst_spec_layer.dropout.p = 0.0
        model.hip_graphs = False
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-5)
        # init_scale 2^7 instead of the default 2^16: this random-init problem has weight gradients up to ~140 (fp32 run), so any scale beyond ~470
        # overflows the fp16 weight gradient of the stem conv -- as it does under torch's own fp16 autocast -- and the scaler spends its first steps
        # backing off (inf gradients -> skipped updates, engine/train.py:277-279); three steps would then compare nothing but forwards
        scaler = torch.amp.GradScaler('cuda', init_scale=128.0, enabled=scaler_on)
        losses, seen = [], set()
        for i in range(3):
            seed_all(50 + i)
            opt.zero_grad()
            with torch.autocast('cuda', dtype=dtype, enabled=dtype != torch.float32):
                seen.add(MF.compute_dtype())
                out, loss = model(batch)
            scaler.scale(loss['total']).backward()
            scaler.unscale_(opt)
            torch.nn.utils.clip_grad_norm_(model.parameters(), 0.01)
            scaler.step(opt)
            scaler.update()
            losses.append(float(loss['total']))
        return losses, scaler.get_scale() if scaler_on else None, seen

    prev = MF.FP16_AUTOCAST_AS_BF16
    try:
        MF.FP16_AUTOCAST_AS_BF16 = False
        l16, scale, seen = steps(torch.float16, True)
        assert seen == {torch.float16}
        l32, _, _ = steps(torch.float32, False)
        lbf, _, _ = steps(torch.bfloat16, False)
        MF.FP16_AUTOCAST_AS_BF16 = True
        MF._FP16_WARNED = False
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            lmap, mscale, mseen = steps(torch.float16, True)
        assert any('bf16 kernels' in str(x.message) for x in w) and mseen == {torch.bfloat16} and mscale == 128.0
    finally:
        MF.FP16_AUTOCAST_AS_BF16 = prev
    assert all(np.isfinite(v) for v in l16) and scale == 128.0, (l16, scale)       # no step was skipped
    # the first step is a pure forward comparison (same weights). On this tiny problem (train-mode BatchNorm over 2-32 samples per channel) repeated
    # runs of ONE dtype already differ by ~1 % (atomics order), fp16 measures 6.34-6.41 and bf16 6.48-6.67 against fp32's 6.478: bound 4 %.
    # (The kernel-level fp16 bars are in test_gpu_conv.py / test_gpu_kernels.py: 4e-3 against bf16's 2.5e-2.)
    d16 = abs(l16[0] - l32[0]) / abs(l32[0])
    assert d16 <= 0.04, (l16, lbf, l32)
    # later steps: after ONE AdamW step of lr 1e-5 the loss of this problem lands anywhere between 4.8 and 6.5 in repeated bf16 runs with identical
    # seeds (atomics order x batch statistics over a handful of samples), so values are not comparable across runs -- what must hold for every
    # family is that no update was skipped (scale unchanged, asserted above), everything stays finite and the loss FALLS like fp32's does
    for ls in (l16, l32, lbf, lmap):
        assert all(np.isfinite(v) for v in ls) and ls[2] < ls[0] - 0.3, ls
    assert abs(lmap[0] - lbf[0]) <= 0.06 * abs(lbf[0]), (lmap, lbf)          # the mapped run IS the bf16 kernels (bf16 first-step spread: 6.48-6.80)


def test_fp16_eval_forward_sits_closer_to_fp32_than_bf16():
    """Eval forward (running-statistics BatchNorm) of the image model under fp16 and bf16 autocast against fp32: the fp16
    kernel family must be the closer one (11 against 8 mantissa bits) and keep the detail region essentially unchanged."""
    from maggie_amd.utils import synth
    dev = _dev()
    batch = _to(synth.synthetic_batch(1, 1, 3, 128, 128, seed=DSEED, train=False, max_inst=10), dev)
    model, _ = _build('image', dev, False)
    model.hip_graphs = False
    outs = {}
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        with torch.no_grad(), torch.autocast('cuda', dtype=dt, enabled=dt != torch.float32):
            o = model(batch)
        outs[dt] = (o['refined_masks'].float().clone(), o['detail_mask'].clone())
    ref, refm = outs[torch.float32]
    e16 = (outs[torch.float16][0] - ref).abs().mean().item()
    ebf = (outs[torch.bfloat16][0] - ref).abs().mean().item()
    flip16 = (outs[torch.float16][1] != refm).float().mean().item()
    assert torch.isfinite(outs[torch.float16][0]).all()
    # (random-init weights with identity running statistics: the eval network amplifies rounding far more than a trained one -- bf16 sits ~0.15
    # mean-abs from fp32 here; the claim under test is the ORDER of the two families, and a bound well inside bf16's distance)
    assert e16 <= 0.75 * ebf, (e16, ebf)
    assert flip16 <= 2e-2, flip16


def test_bounded_sparse_capacity_raises_instead_of_refining_fewer_sites():
    """VERDICT round 2, weak #8 / ADVICE: the sparse head's row buffers may be sized for a FRACTION of "every site active"
    (decoder.sparse_capacity_frac / MAGGIE_SPARSE_CAPACITY). A step whose detail region exceeds it must not read or write past a buffer, and the
    model must say so (MaggieHipError at its next flag read) rather than silently refine fewer sites; with enough capacity results are unchanged.
    Training-mode forward with the ground-truth-guided detail region (iter 100): a deterministic region of a few percent of the 10 x H x W sites."""
    from maggie_amd.hip import MaggieHipError
    from maggie_amd.utils import synth
    dev = _dev()
    batch = _to(synth.synthetic_batch(2, 1, 2, 128, 128, seed=DSEED, train=True, max_inst=10, it=100), dev)
    model, _ = _build('image', dev, True)
    model.decoder.inst_spec_layer.dropout.p = 0.0
    model.hip_graphs = False
    state = copy.deepcopy(model.state_dict())

    def run():
        model.load_state_dict(state)
        seed_all(5)
        with torch.no_grad():
            return model(batch)[0]

    ref = run()
    active = float(ref['detail_mask'].float().mean())
    assert 0.005 < active < 0.2, active
    model.decoder.sparse_capacity_frac = active * 3                                     # roomy (OS1 fraction; coarser levels get 2x per level): same result
    out = run()
    assert torch.equal(out['detail_mask'], ref['detail_mask'])
    assert float((out['refined_masks'] - ref['refined_masks']).abs().max()) <= 2e-3       # (train-mode BatchNorm: fp32 atomics order between two runs)
    run()                                                                               # and no complaint on the next read
    model.decoder.sparse_capacity_frac = active / 8                                     # too small: the step runs (safely) ...
    small = run()
    assert torch.isfinite(small['refined_masks']).all()
    assert 0 < float(small['detail_mask'].float().mean()) < active                      # ... on fewer sites
    with pytest.raises(MaggieHipError, match='sparse_capacity'):
        run()                                                                           # ... and the next forward refuses to go on
    model.decoder.sparse_capacity_frac = 1.0
    again = run()
    assert torch.equal(again['detail_mask'], ref['detail_mask'])


def test_auto_sparse_capacity_follows_the_workload(caplog):
    """the default capacity of the sparse head's row buffers follows the workload ('auto': 1.5x the high-water mark of
    live sites per level, maggie_amd/network/decoder/resnet_inst_matt_spconv.py sparse_capacity) instead of "all sites of all 10 padded slots"
    (18 GB at the headline shape, 400 GB at the reference's video shape). Same results as the unbounded setting, step after step, eager and
    replayed; capacities end up between the live count and a fraction of the full size; a step that exceeds them all the same is reported, the
    capacities grow and the NEXT step is complete again."""
    import logging
    from maggie_amd.utils import synth
    dev = _dev()
    batch = _to(synth.synthetic_batch(2, 1, 2, 128, 128, seed=DSEED, train=True, max_inst=10, it=100), dev)
    outs = {}
    for mode in (1.0, 'auto'):
        model, _ = _build('image', dev, True)
        model.decoder.inst_spec_layer.dropout.p = 0.0
        model.decoder.sparse_capacity_frac = mode
        seq = []
        for i in range(6):                                       # eager, eager (capacities tuned: new graph key), capture, replays
            seed_all(5)
            with torch.no_grad():
                o = model(batch)[0]
            seq.append({k: o[k].clone() for k in ('detail_mask', 'refined_masks', 'alpha_os1', 'alpha_os4')})
        outs[mode] = (seq, model)
    # same index map, exactly; the alphas to rounding: the capacity sizes the persistent grids of the head's kernels, and the order in which the
    # BatchNorm1d partial sums over the live rows are added is a function of the launch geometry (bit-reproducible for ONE capacity, csrc/det.hip)
    worst = 0.0
    for a, b_ in zip(outs[1.0][0], outs['auto'][0]):
        assert torch.equal(a['detail_mask'], b_['detail_mask'])
        for k in ('refined_masks', 'alpha_os1', 'alpha_os4'):
            worst = max(worst, float((a[k] - b_[k]).abs().max()))
    record('auto_sparse_capacity', max_alpha_diff_vs_unbounded=worst)
    assert worst <= 1e-4, worst
    dec = outs['auto'][1].decoder
    st = dec._sparse_auto[(2 * 10, 128, 128, True)]
    live = int(outs['auto'][0][-1]['detail_mask'].sum())
    assert st['tuned'] and st['hwm'][0] == live, (st, live)
    assert live <= st['caps'][0] <= max(4096, 2 * live) and st['caps'][0] < st['full'][0] // 4, st
    assert all(h <= c <= f for h, c, f in zip(st['hwm'], st['caps'], st['full'])), st
    # a step that exceeds its capacity (same geometry, 5 instances with wide soft edges instead of 2 with narrow ones): the dropped sites are REPORTED
    # at the next flag read, the capacities grow, and the step after that is complete again
    big = _to(synth.synthetic_batch(2, 1, 5, 128, 128, seed=DSEED + 1, train=True, max_inst=10, it=100, edge=24.0), dev)
    ref_model = outs[1.0][1]
    seed_all(5)
    with torch.no_grad():
        want_big = ref_model(big)[0]
    live_big = int(want_big['detail_mask'].sum())
    assert live_big > st['caps'][0], (live_big, st)              # (otherwise this part tests nothing)
    model = outs['auto'][1]
    seed_all(5)
    with torch.no_grad():
        small = model(big)[0]
    assert 0 < int(small['detail_mask'].sum()) < live_big
    with caplog.at_level(logging.WARNING):
        for _ in range(4):
            seed_all(5)
            with torch.no_grad():
                got = model(big)[0]
    assert any('more active sites than the sparse head was sized for' in r.message for r in caplog.records)
    for _ in range(4):                                           # (every forward advances the SpectralNorm power iteration: same number of calls on both sides)
        seed_all(5)
        with torch.no_grad():
            want_big = ref_model(big)[0]
    assert torch.equal(got['detail_mask'], want_big['detail_mask']) and st['caps'][0] >= live_big
    assert float((got['refined_masks'] - want_big['refined_masks']).abs().max()) <= 1e-4


@pytest.mark.parametrize('kind,b,n_f', [('video', 2, 3), ('image', 4, 1)])
def test_fp16_train_step_of_both_models_close_to_fp32(kind, b, n_f):
    """The fp16 kernel family at model level for both architectures (the video model adds the ConvGRU gate kernels, the frame-difference module and
    the bidirectional fusion in fp16): one training forward + backward under fp16 autocast against the same step in fp32 -- every loss term finite
    and the total within 3 % (bf16 measures 0.1-1.3 % at full size, fp16 0.05 %; this small train-mode-BatchNorm problem is noisier), finite
    gradients for every parameter that has one in fp32."""
    from maggie_amd.utils import synth
    dev = _dev()
    batch = _to(synth.synthetic_batch(b, n_f, 2, 128, 128, seed=DSEED, train=True, max_inst=10, it=100), dev)
    model, _ = _build(kind, dev, True)
    model.decoder.inst_spec_layer.dropout.p = 0.0
    model.hip_graphs = False
    state = copy.deepcopy(model.state_dict())
    res = {}
    for dt in (torch.float32, torch.float16):
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        seed_all(31)
        with torch.autocast('cuda', dtype=dt, enabled=dt != torch.float32):
            out, loss = model(batch)
        loss['total'].backward()
        res[dt] = ({k: float(v.detach()) for k, v in loss.items()}, {n: p.grad for n, p in model.named_parameters() if p.grad is not None})
    l32, g32 = res[torch.float32]
    l16, g16 = res[torch.float16]
    assert all(np.isfinite(v) for v in l16.values()), l16
    assert abs(l16['total'] - l32['total']) <= 3e-2 * abs(l32['total']), (l16['total'], l32['total'])
    assert set(g16) == set(g32)
    assert all(bool(torch.isfinite(g).all()) for g in g16.values())
