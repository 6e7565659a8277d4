"""MaGGIe image decoder: dense OS32->OS8, instance matte decoder, sparse progressive refinement OS8->OS4->OS2->OS1 --
mirrors maggie/network/decoder/resnet_inst_matt_spconv.py:14-391 (same module tree / state_dict keys).

The spconv calls of the reference are replaced by: bit-plane region kernels (active-site pyramid, gather tables) and the
MG_MODE_GATHER implicit-GEMM kernel. `dummy_downscale` (reference :61-66) existed only to make spconv build rule books;
its parameters are kept (frozen) for checkpoint compatibility and never executed."""
import random

import torch
from torch import nn
from torch.nn import functional as F

from ... import functional as MF
from ... import kernels as K
from .resnet import BasicBlock
from ..module import SpectralNorm, conv1x1, InstanceMatteDecoder, Marker
from ..module.mask_attention import FFNLayer
from ...sparse_head import DevicePyramid, DeviceRng, SparseHead


class SparseConvWeight(nn.Module):
    """Parameter holder with spconv's layout: weight (Cout, k, k, Cin) [+ bias]."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, kind='subm', indice_key=None, **kw):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.kind, self.indice_key = in_channels, out_channels, kernel_size, kind, indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, kernel_size, kernel_size, in_channels))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter('bias', None)

    def krsc(self, dtype):
        return MF.weight_krsc_param(self.weight, dtype, None, MF.pad8(self.out_channels))

    def bias32(self):
        return None if self.bias is None else MF.pad_vec(self.bias.float(), MF.pad8(self.out_channels))


def SubMConv2d(i, o, kernel_size, padding=0, bias=True, indice_key=None, stride=1):
    return SparseConvWeight(i, o, kernel_size, bias, 'subm', indice_key)


def SparseConv2d(i, o, kernel_size, stride=2, padding=1, bias=True, indice_key=None):
    return SparseConvWeight(i, o, kernel_size, bias, 'down', indice_key)


def SparseInverseConv2d(i, o, kernel_size, bias=True, indice_key=None):
    return SparseConvWeight(i, o, kernel_size, bias, 'inverse', indice_key)


class ResShortCut_InstMattSpconv_Dec(nn.Module):
    def __init__(self, block, layers, norm_layer=None, large_kernel=False, late_downsample=False, atten_stride=1, atten_dim=128,
                 atten_block=2, atten_head=1, final_channel=32, max_inst=10, use_id_pe=True, warmup_mask_atten_iter=4000,
                 warmup_detail_iter=3000, use_query_temp=False, use_detail_temp=False, detail_mask_dropout=0.2, **kwargs):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self._norm_layer = norm_layer
        assert not large_kernel
        self.kernel_size = 3
        self.max_inst = max_inst
        self.inplanes = 512 if layers[0] > 0 else 256
        self.midplanes = 64 if late_downsample else 32
        self.warmup_mask_atten_iter = warmup_mask_atten_iter
        self.warmup_detail_iter = warmup_detail_iter
        self.leaky_relu = Marker('LeakyReLU(0.2)')
        self.inst_spec_layer = FFNLayer(final_channel, final_channel, 0.1)
        self.layer1 = self._make_layer(block, 256, layers[0], stride=2)
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.refine_OS8 = InstanceMatteDecoder(input_dim=128, atten_stride=atten_stride, attention_dim=atten_dim, n_block=atten_block,
                                               n_head=atten_head, output_dim=final_channel, max_inst=max_inst, return_feat=True,
                                               use_temp_pe=False, use_id_pe=use_id_pe)
        self.dummy_downscale = nn.Sequential(
            SubMConv2d(3, 32, 3, padding=1, bias=False, indice_key="subminp"),
            SparseConv2d(32, 32, 3, stride=2, padding=1, bias=False, indice_key="subm1.2"),
            SparseConv2d(32, 64, 3, stride=2, padding=1, bias=False, indice_key="subm2.4"),
            SparseConv2d(64, 64, 3, stride=2, padding=1, bias=False, indice_key="subm4.8"))
        for p in self.dummy_downscale.parameters():      # never receives a gradient in the reference either (no_grad, :217-218)
            p.requires_grad_(False)
        lr = self.leaky_relu
        self.layer3 = nn.Sequential(SparseInverseConv2d(final_channel, 64, 3, bias=False, indice_key="subm4.8"), nn.BatchNorm1d(64), lr,
                                    SubMConv2d(64, 64, 3, padding=1, bias=False, indice_key="subm4.4"))
        self.guidance_layer = nn.Sequential(SubMConv2d(128, 64, 1, padding=0, bias=False, indice_key="subm_inst.0"), nn.BatchNorm1d(64), lr,
                                            SubMConv2d(64, 64, 3, padding=1, bias=True, indice_key="subm_inst.1"), Marker('Sigmoid'))
        self.layer3_smooth = nn.Sequential(SubMConv2d(64, 64, 1, padding=0, bias=True, indice_key="subm4.smooth"), Marker('ReLU'),
                                           nn.BatchNorm1d(64))
        self.layer4 = nn.Sequential(SparseInverseConv2d(64, 32, 3, bias=False, indice_key="subm2.4"), nn.BatchNorm1d(32), lr,
                                    SubMConv2d(32, 32, 1, padding=1, bias=False, indice_key="subm2.2"))
        self.layer4_smooth = nn.Sequential(SubMConv2d(64, 32, 1, padding=0, bias=True, indice_key="subm2.smooth"), Marker('ReLU'),
                                           nn.BatchNorm1d(32))
        self.layer5 = nn.Sequential(SparseInverseConv2d(32, 32, 3, bias=False, indice_key="subm1.2"), nn.BatchNorm1d(32), lr,
                                    SubMConv2d(32, 32, 3, padding=1, bias=False, indice_key="subm1.1"))
        self.layer5_smooth = nn.Sequential(SubMConv2d(64, 32, 1, padding=0, bias=True, indice_key="subm1.smooth"), Marker('ReLU'),
                                           nn.BatchNorm1d(32))
        self.refine_OS4 = nn.Sequential(SubMConv2d(64, 32, 3, stride=1, padding=1, bias=False), nn.BatchNorm1d(32), lr,
                                        SubMConv2d(32, 1, 3, stride=1, padding=1))
        self.refine_OS1 = nn.Sequential(SubMConv2d(32, 32, 3, stride=1, padding=1, bias=False), nn.BatchNorm1d(32), lr,
                                        SubMConv2d(32, 1, 3, stride=1, padding=1))
        self.fea_dropout = nn.Dropout2d(detail_mask_dropout)

    def _make_layer(self, block, planes, blocks, stride=1):
        if blocks == 0:
            return nn.Sequential(nn.Identity())
        norm_layer = self._norm_layer
        upsample = None
        if stride != 1:
            upsample = nn.Sequential(Marker('UpsamplingNearest2d(2)'), SpectralNorm(conv1x1(self.inplanes, planes * block.expansion)),
                                     norm_layer(planes * block.expansion))
        elif self.inplanes != planes * block.expansion:
            upsample = nn.Sequential(SpectralNorm(conv1x1(self.inplanes, planes * block.expansion)), norm_layer(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, upsample, norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, norm_layer=norm_layer))
        return nn.Sequential(*layers)

    # ------------------------------------------------------------------ sparse refinement head
    _HEAD_BNS = ('layer3.1', 'guidance_layer.1', 'layer3_smooth.2', 'refine_OS4.1', 'layer4.1', 'layer4_smooth.2', 'layer5.1',
                 'layer5_smooth.2', 'refine_OS1.1')

    def _head_env(self, pyr, n_i, dtype):
        """Everything SparseHead needs besides the dense inputs: the pyramid, the converted weights (ONE weight-bank launch each way) and
        the BatchNorm / LayerNorm parameters, as one flat `params` list with name -> position tables."""
        plan, slots = self._weight_bank_plan()
        names = self.__dict__.get('_head_names')
        if names is None:
            by_id = {id(m): n for n, m in self.named_modules()}
            ffn = self.inst_spec_layer
            by_id[id(ffn.linear1)], by_id[id(ffn.linear2)] = 'ffn1', 'ffn2'
            names = self.__dict__['_head_names'] = [(by_id[id(m)], what) for m, what in slots]
        outs = MF.weight_bank(plan, dtype, [m.weight if what == 'w' else m.bias for m, what in slots])     # live tensors (capture aliases)
        env = type('HeadEnv', (), {})()
        env.dec, env.pyr, env.n_i = self, pyr, n_i
        env.w_index = {n: i for i, (n, what) in enumerate(names) if what == 'w'}
        env.b_index = {n: i for i, (n, what) in enumerate(names) if what == 'b'}
        params = list(outs)
        env.bn_index = {}
        for n in self._HEAD_BNS:
            mod, idx = n.split('.')
            bn = getattr(self, mod)[int(idx)]
            env.bn_index[n] = len(params)
            params += [bn.weight, bn.bias]
        env.ln_index = len(params)
        params += [self.inst_spec_layer.norm.weight, self.inst_spec_layer.norm.bias]
        env.n_params, env.params = len(params), params
        env.rng_state = None
        if self.inst_spec_layer.training and self.inst_spec_layer.dropout.p > 0:
            rng = self.__dict__.get('_head_rng')
            if rng is None or rng.state.device != outs[0].device:
                rng = self.__dict__['_head_rng'] = DeviceRng(outs[0].device)
            env.rng_state = rng.snapshot()                        # device ops only: same sequence eagerly and from a replayed graph
        return env

    def predict_details(self, os8_feat, roi_bits, n_i, inst_guidance_os8, dense_features, H, W):
        """os8_feat (N,h8,w8,64) NHWC; roi_bits (N*n_i, H, Ww) bit planes (patched IN PLACE when empty in training, :347-348);
        inst_guidance_os8 (N,10,64); dense_features = fea1 (N,H,W,32), fea2 (N,H/2,W/2,32), fea3 (N,H/4,W/4,64).
        Returns fp32 planes (N*n_i, H/4, W/4) and (N*n_i, H, W) with -99 outside the active sites, and the pyramid.
        No site count ever reaches the host (maggie_amd/sparse_head.py)."""
        fea1, fea2, fea3 = dense_features
        # "dummy code to prevent all zeros" (:347-348): `unknown_os8[:, :, 200:250, 200:250] = 1` -- a slice assignment, so the square
        # is clipped to the plane (and is a no-op on planes of 200 pixels or less)
        patch = (200, min(250, H), 200, min(250, W)) if (self.training and H > 200 and W > 200) else None
        cap = self.sparse_capacity()
        if cap == 'auto':
            pyr = DevicePyramid(roi_bits, H, W, patch, 1.0, self.sparse_overflow_flag(roi_bits.device), caps=self.sparse_auto_caps(roi_bits.shape[0], H, W, self.training))
            # live site counts of the four levels into the persistent flag words: the NEXT step's one host read picks them up (no extra sync)
            torch.cat([l.count for l in pyr.levels], out=self.sparse_flag_words(roi_bits.device)[4:8])
        else:
            pyr = DevicePyramid(roi_bits, H, W, patch, cap, self.sparse_overflow_flag(roi_bits.device))
        env = self._head_env(pyr, n_i, os8_feat.dtype)
        x_os4, x_os1 = SparseHead.apply(env, os8_feat.contiguous(), inst_guidance_os8, fea1.contiguous(), fea2.contiguous(), fea3.contiguous(),
                                        *env.params)
        return x_os4, x_os1, pyr

    def sparse_capacity(self):
        """How the sparse head's row buffers are sized: attribute `sparse_capacity_frac`, env MAGGIE_SPARSE_CAPACITY.
        * 'auto' (default, one process): capacities follow the WORKLOAD -- 1.5x the high-water mark of live sites per level seen so far (first
          step of a geometry: every site, capped at 8 M OS1 rows), grown and the detail graphs re-captured when a step comes within 20 % of a
          capacity. Training only: inference keeps the initial sizing (its planes are the real instances only). The reference sizes nothing in advance (spconv allocates per call, resnet_inst_matt_spconv.py:170-177); the
          round-3 default -- all sites of all 10 padded instance slots -- needed 18 GB at batch 4 and 400 GB at the reference's own video shape
          (maggie_video.yaml:32,89). A step that exceeds its capacity all the same drops the sites beyond it ON THE DEVICE (nothing reads or
          writes past a buffer), is reported with a warning at the next step, and the capacity is raised 4x.
        * a fraction < 1.0: fixed fraction of all sites; a step that exceeds it raises MaggieHipError (after the fact: at the next forward's flag read).
        * 1.0: every site may be active, nothing can overflow. Also what 'auto' means in a multi-rank job: the ranks must replay the same graphs."""
        v = self.__dict__.get('sparse_capacity_frac')
        if v is None:
            import os
            v = os.environ.get('MAGGIE_SPARSE_CAPACITY', 'auto')
        if isinstance(v, str):
            if v.strip().lower() == 'auto':
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    return 1.0
                return 'auto'
            v = float(v)
        return float(v)

    def sparse_bounded(self):
        cap = self.sparse_capacity()
        return cap == 'auto' or cap < 1.0

    # ---- 'auto' capacity: per (planes, H, W) geometry the four level capacities, their high-water marks and whether they were tuned yet ----
    def sparse_auto_caps(self, P, H, W, training=True):
        auto = self.__dict__.setdefault('_sparse_auto', {})
        st = auto.get((P, H, W, bool(training)))
        if st is None:
            full = [P * max(1, H >> k) * max(1, W >> k) for k in range(4)]
            # before anything has been observed: every site (cannot overflow) up to 8 M OS1 rows (~14 GB of head buffers), half per coarser level
            caps = [min(f, (8 << 20) >> k) for k, f in enumerate(full)]
            st = auto[(P, H, W, bool(training))] = {'full': full, 'caps': caps, 'hwm': [0, 0, 0, 0], 'tuned': False}
        return list(st['caps'])

    def sparse_caps_version(self):
        return self.__dict__.get('_sparse_caps_version', 0)

    def sparse_note_counts(self, counts, overflowed):
        """Host side of 'auto', at the step's one flag read: `counts` = live sites per level of the PREVIOUS detail stage (clamped to its capacities),
        `overflowed` = its sticky overflow flag. Raises the capacities when needed (new version -> the detail graphs are captured again)."""
        key = self.__dict__.get('_sparse_last_key')
        st = self.__dict__.get('_sparse_auto', {}).get(key)
        if st is None:
            return
        st['hwm'] = [max(h, int(c)) for h, c in zip(st['hwm'], counts)]
        want = [min(f, max(4096, -(-int(h * 1.5) // 4096) * 4096)) for f, h in zip(st['full'], st['hwm'])]
        new = list(st['caps'])
        if overflowed:
            import logging
            logging.warning('MaGGIe (MI355X build): the detail region of the previous step had more active sites than the sparse head was sized for '
                            '(capacities %s, auto mode); the sites beyond were dropped for that step. Capacities raised 4x, detail graphs re-captured.', st['caps'])
            new = [min(f, max(w, 4 * c)) for f, w, c in zip(st['full'], want, st['caps'])]
        elif not key[3]:
            pass                                                  # inference: never below the initial sizing (only the real instances' planes exist there,
            #                                                       and a dropped site is a wrong output, not a perturbed gradient) -- growth on overflow only
        elif not st['tuned']:
            new = want                                            # first observation: from the initial guess to the workload (either direction)
        else:
            new = [max(c, w) if h * 1.2 > c else c for c, w, h in zip(st['caps'], want, st['hwm'])]
        st['tuned'] = True
        if new != st['caps']:
            st['caps'] = new
            self.__dict__['_sparse_caps_version'] = self.sparse_caps_version() + 1

    def sparse_flag_words(self, device):
        """Persistent int32 [8]: words 0-3 = the step flags (mg_step_flags), 4-7 = live sites per level of the last detail stage ('auto' capacity).
        Created once: its address is baked into the captured graphs."""
        f = self.__dict__.get('_sparse_flag_words')
        if f is None or f.device != device:
            f = self.__dict__['_sparse_flag_words'] = torch.zeros(8, dtype=torch.int32, device=device)
        return f

    def sparse_overflow_flag(self, device):
        """Sticky int32 [1] device flag: 1 once any level of any step dropped sites (created once: its address is baked into the captured graphs)."""
        f = self.__dict__.get('_sparse_overflow')
        if f is None or f.device != device:
            f = self.__dict__['_sparse_overflow'] = torch.zeros(1, dtype=torch.int32, device=device)
        return f

    def fuse(self, pred, detail_bits, widths=None, want_bits=False):
        """Progressive refinement (:272-290) with the two compute_unknown calls on device bit planes. `widths` (2, P) device int32: the
        train-mode dilation widths for k = 27 and k = 15 when the caller already drew them (detail_plan). `want_bits`: return the two
        weight planes as bit planes (the caller selects between them and the ground-truth-guided ones on the device, then unpacks once)."""
        a1, a4, a8 = pred['alpha_os1'], pred['alpha_os4'], pred['alpha_os8']
        H, W = a8.shape[-2:]
        alpha = a8
        w27, w15 = (widths[0], widths[1]) if widths is not None else (None, None)
        # a*w + alpha*(1-w) with a 0/1 weight plane is a per-pixel select: done straight from the bit planes (one pass each way)
        bits4 = MF.unknown_bits(alpha, 27, self.training, andmask=detail_bits, widths=w27)
        alpha = MF.bits_select(bits4, a4, alpha, W)
        bits1 = MF.unknown_bits(alpha, 15, self.training, andmask=detail_bits, widths=w15)
        alpha = MF.bits_select(bits1, a1, alpha, W)
        if want_bits:
            return alpha, bits4, bits1
        w4 = K.bits_unpack_u8(bits4, W, a8.shape).to(alpha.dtype)
        w1 = K.bits_unpack_u8(bits1, W, a8.shape).to(alpha.dtype)
        return alpha, w4, w1

    def os32_to_os8(self, x, mid_fea, b, n_f, n_i, masks, gt_alphas):
        masks = masks.reshape(b, n_f, n_i, masks.shape[2], masks.shape[3])
        m2 = masks.flatten(0, 1)
        if m2.is_cuda and m2.dtype == torch.float32 and m2.is_contiguous():
            # `sum > 0` of a non-negative mask plane == "some element > 0": one launch, already the 0.0 / 1.0 scale the up-sampling kernel multiplies with
            valid_masks = K.plane_flags(m2, as_float=True).view(m2.shape[0], m2.shape[1], 1, 1)
        else:
            valid_masks = m2.sum((2, 3), keepdim=True) > 0
        gt_masks = None
        if self.training:
            # (gt_alphas > 0) of :322 is applied AFTER the decoder's max-pooling to OS8 (max > 0 <=> any > 0): no full-resolution compare pass
            gt_masks = gt_alphas.reshape(b, n_f, n_i, gt_alphas.shape[2], gt_alphas.shape[3])
        fea1, fea2, fea3, fea4, fea5 = mid_fea['shortcut']
        image = mid_fea['image']
        x = self.layer1[0](x, link_out=True)
        x = self.layer1[1](x, post_add=fea5)
        x = self.layer2[0](x, link_out=True)
        x = self.layer2[1](x, link_out=True)
        x = self.layer2[2](x, post_add=fea4)
        h, w = image.shape[-2:]
        return x, masks, valid_masks, gt_masks, fea1, fea2, fea3, image, h, w

    def process_os4_os1(self, x, b, n_f, fea1, fea2, fea3, hw, x_os8, queries, n_i, detail_bits):
        """:346-366. Returns alpha_os4, alpha_os1 (N, n_i, H, W) fp32 and the (possibly patched) detail bit planes."""
        H, W = hw
        N = b * n_f
        queries = queries[:, None].expand(-1, n_f, -1, -1).reshape(N, *queries.shape[1:]).contiguous()
        # an empty region needs no host decision: in training the device patches it (predict_details), in eval every plane stays at
        # -99 and (tanh(-99) + 1) / 2 is exactly 0 -- the zeros of :350-352
        x_os4, x_os1, pyr = self.predict_details(x, detail_bits, n_i, queries, [fea1, fea2, fea3], H, W)
        x_os4 = MF.upsample_tanh(x_os4.view(N, n_i, H // 4, W // 4), n_i, 4, False)
        x_os1 = MF.upsample_tanh(x_os1.view(N, n_i, H, W), n_i, 1, False)
        return x_os4, x_os1, detail_bits

    # The forward pass is split at the point where shapes stop being a function of the batch geometry alone:
    #   dense_stage  -- OS32 -> OS8 decoder + instance matte decoder (static shapes: capturable in a hipGraph, graphs.py)
    #   detail_stage -- detail region, sparse refinement, fusion (data-dependent row counts)
    def _weight_bank_plan(self):
        """Every parameter the sparse head converts per step: the spconv-layout weights / biases and inst_spec_layer's linears."""
        plan = self.__dict__.get('_wb_plan')
        if plan is None:
            skip = {id(m) for m in self.dummy_downscale.modules()}
            convs = [m for m in self.modules() if isinstance(m, SparseConvWeight) and id(m) not in skip]
            items, slots = [], []
            for m in convs:
                co, k, ci = m.out_channels, m.kernel_size, m.in_channels
                items.append((m.weight, (co, k * k, ci), MF.pad8(co), MF.pad8(ci), int(m.kind == 'subm' and k > 1), False))
                slots.append((m, 'w'))
                if m.bias is not None:
                    items.append((m.bias, (1, 1, co), 1, MF.pad8(co), 0, True))
                    slots.append((m, 'b'))
            ffn = self.inst_spec_layer
            for lin in (ffn.linear1, ffn.linear2):
                co, ci = lin.weight.shape
                items.append((lin.weight, (co, 1, ci), co, MF.pad8(ci), 0, False))
                slots.append((lin, 'w'))
                items.append((lin.bias, (1, 1, co), 1, co, 0, True))
                slots.append((lin, 'b'))
            plan = self.__dict__['_wb_plan'] = (MF.WeightBankPlan(items), slots)
        return plan

    def head_state(self):
        """Device state the detail stage mutates besides module buffers (rolled back after a capture's warm-up runs): the dropout counter."""
        rng = self.__dict__.get('_head_rng')
        return ([rng.state] if rng is not None else []) + ([self.__dict__['_sparse_overflow']] if '_sparse_overflow' in self.__dict__ else []) + \
               ([self.__dict__['_sparse_flag_words']] if '_sparse_flag_words' in self.__dict__ else [])

    def dense_modules(self):
        """Sub-modules whose parameters are touched by dense_stage only."""
        return [self.layer1, self.layer2, self.refine_OS8]

    def plain_trunk_convs(self):
        """Ordinary (not spectrally normalised) conv holders of the dense stage, converted with the batched weight pipeline."""
        return self.refine_OS8.plain_convs()

    def _refine_os8(self, x, masks, gt_masks, n_f, mem_feat):
        return self.refine_OS8(x, masks, use_mask_atten=False, gt_mask=gt_masks)

    def dense_stage(self, x, mid_fea, b, n_f, n_i, masks, gt_alphas, mem_feat=None):
        """:318-339. Returns (alpha_os8 (N, 10, H, W) fp32 [train: times valid_masks], OS8 features, queries, loss_max_atten,
        hidden_state | None, fea1, fea2, fea3) and, in training, as LAST element an int32 [1] device flag: 1 when alpha_os8 has a non-zero
        element (split_dense_flag() takes it off again)."""
        x, masks, valid_masks, gt_masks, fea1, fea2, fea3, image, h, w = self.os32_to_os8(x, mid_fea, b, n_f, n_i, masks, gt_alphas)
        if isinstance(fea1, MF.Deferred):
            # the three fine shortcut branches of the encoder, deferred to here (MAGGIE_SIDE_SHORTCUTS): issued on the side stream, next to the
            # instance-token chain below; the trunk joins the streams before it returns
            fea1, fea2, fea3 = fea1.run(), fea2.run(), fea3.run()
        x_os8, x, queries, loss_max_atten, hidden_state = self._refine_os8(x, masks, gt_masks, n_f, mem_feat)
        if self.training:
            # `x_os8 * valid_masks` (:331) and the `x_os8.sum() == 0` test of :314 ride on the up-sampling kernel: a 0 / 1 scale per plane and
            # a "some element is non-zero" flag, instead of a multiply and a reduction over the (N, 10, H, W) planes (and a multiply in backward)
            x_os8, nonzero = MF.upsample_tanh(x_os8, self.max_inst, h // x_os8.shape[1], True, pscale=valid_masks, want_flag=True)
        else:
            x_os8, nonzero = MF.upsample_tanh(x_os8, self.max_inst, h // x_os8.shape[1], True), None          # (N, 10, H, W) fp32
        if not torch.is_tensor(loss_max_atten):
            loss_max_atten = x_os8.new_zeros(())
        out = (x_os8, x, queries, loss_max_atten, hidden_state, fea1, fea2, fea3)
        return out + (nonzero,) if self.training else out

    def split_dense_flag(self, dense):
        """dense_stage()'s tuple -> (tuple without the trailing flag, flag | None)."""
        return (tuple(dense[:-1]), dense[-1]) if self.training else (tuple(dense), None)

    def forward(self, x, mid_fea, b, n_f, n_i, masks, iter, gt_alphas, **kwargs):
        dense, nonzero = self.split_dense_flag(self.dense_stage(x, mid_fea, b, n_f, n_i, masks, gt_alphas, kwargs.get('mem_feat')))
        x_os8 = dense[0]
        flags = torch.stack([torch.isnan(dense[2]).any(), (nonzero[0] == 0) if nonzero is not None else (x_os8.sum() == 0)]).tolist()
        if flags[0]:
            raise ValueError("Mask is empty")
        P = x_os8.shape[0] * (x_os8.shape[1] if self.training else n_i)
        plan = self.detail_plan(iter, bool(flags[1]), P, x_os8.device)
        return self.detail_stage(dense, mid_fea['image'].shape[-2:], b, n_f, n_i, plan, gt_alphas, spar_gt=kwargs.get('spar_gt'))

    def detail_plan(self, iter, coarse_is_zero, P, device):
        """Everything the detail stage needs from the HOST, consumed in the reference's order (:312-316 `random.random()`, then the per-slice
        `np.random.randint` widths of compute_unknown(is_train=True): fuse k = 27, k = 15 (:276,282), then -- only when the ground truth
        guides the region -- k = 30, k = 15 (:326-327)). -> dict(use_gt, with_atten, widths (4, P) device int32 | None)."""
        use_gt = bool(self.training and (iter < self.warmup_detail_iter or coarse_is_zero
                                         or (iter < self.warmup_detail_iter * 3 and random.random() < 0.5)))
        widths = None
        if self.inst_spec_layer.training and self.inst_spec_layer.dropout.p > 0:          # created here, never inside a capture
            rng = self.__dict__.get('_head_rng')
            if rng is None or rng.state.device != device:
                self.__dict__['_head_rng'] = DeviceRng(device)
        if self.training:
            import numpy as np
            rows = [MF.draw_widths(P, 27), MF.draw_widths(P, 15)]
            rows += [MF.draw_widths(P, 30), MF.draw_widths(P, 15)] if use_gt else [np.ones(P, np.int32)] * 2
            widths = torch.from_numpy(np.stack(rows)).to(device, non_blocking=True)
        return {'use_gt': use_gt, 'with_atten': bool(self.training and iter >= self.warmup_mask_atten_iter), 'widths': widths}

    def detail_stage(self, dense, hw, b, n_f, n_i, plan, gt_alphas, spar_gt=None):
        """Tensors in, tensors out, static shapes, no host read (graph-capturable): detail region -> sparse refinement -> fusion."""
        x_os8, x, queries, loss_max_atten, _, fea1, fea2, fea3 = dense[:8]
        h, w = hw
        if not self.training:
            x_os8 = x_os8[:, :n_i].contiguous()
        use_gt, widths = plan['use_gt'], plan['widths']
        gt_dev = plan.get('use_gt_dev')
        if gt_dev is not None:
            # Data-parallel runs (arch/maggie.py:_rank_safe_graphs): whether the ground truth guides the detail region is a PER-RANK decision
            # (random.random(), `x_os8.sum() == 0`), and it must not select a different captured graph on different ranks -- the graphs carry
            # the gradient collectives. Both guidance sources have the same static shape, so both are evaluated (bit planes: 1/64 of the
            # alpha planes) and the device flag selects between them word by word.
            sel = gt_dev.bool()
            detail_bits = torch.where(sel, MF.unknown_bits(gt_alphas, 30, False), MF.unknown_bits(x_os8, 30, False))
            n_cur = x_os8.shape[1]
        else:
            guided = gt_alphas if use_gt else x_os8
            n_cur = guided.shape[1]
            detail_bits = MF.unknown_bits(guided, 30, False)                               # (N*n_cur, H, Ww)
        x_os4, x_os1, detail_bits = self.process_os4_os1(x, b, n_f, fea1, fea2, fea3, hw, x_os8, queries, n_cur, detail_bits)
        ret = {'alpha_os1': x_os1, 'alpha_os4': x_os4, 'alpha_os8': x_os8}
        # the two loss-weight planes leave as BIT planes (1/64 of the bytes): the caller selects between them and the detail mask word by word
        # and unpacks once, straight to fp32 (was: unpack to uint8, cast to fp32, torch.where over 42 MB planes -- per weight plane)
        if gt_dev is not None:
            alpha_pred, bits4, bits1 = self.fuse(ret, detail_bits, widths, want_bits=True)
            g4 = MF.unknown_bits(gt_alphas, 30, self.training, andmask=detail_bits, widths=widths[2])
            g1 = MF.unknown_bits(gt_alphas, 15, self.training, andmask=detail_bits, widths=widths[3])
            wbits4, wbits1 = torch.where(sel, g4, bits4), torch.where(sel, g1, bits1)
        else:
            alpha_pred, wbits4, wbits1 = self.fuse(ret, detail_bits, widths, want_bits=True)
        ret['refined_masks'] = alpha_pred
        unknown_os8 = K.bits_unpack_u8(detail_bits, w, x_os8.shape)
        if use_gt and gt_dev is None:
            wbits4 = MF.unknown_bits(gt_alphas, 30, self.training, andmask=detail_bits, widths=widths[2])
            wbits1 = MF.unknown_bits(gt_alphas, 15, self.training, andmask=detail_bits, widths=widths[3])
        ret['weight_os4_bits'] = wbits4
        ret['weight_os1_bits'] = wbits1
        ret['detail_bits'] = detail_bits
        ret['detail_mask'] = unknown_os8
        if plan['with_atten']:
            ret['loss_max_atten'] = loss_max_atten
        return ret


def res_shortcut_inst_matt_spconv_22(**kwargs):
    return ResShortCut_InstMattSpconv_Dec(BasicBlock, [2, 3, 3, 2], **kwargs)
