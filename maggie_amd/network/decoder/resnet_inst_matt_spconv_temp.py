"""MaGGIe video decoder: adds ConvGRU feature propagation at OS8, the feature-difference module and bidirectional alpha
fusion -- mirrors maggie/network/decoder/resnet_inst_matt_spconv_temp.py:14-206."""
from functools import partial
import torch
from torch import nn
from torch.nn import functional as F

from ... import functional as MF
from ... import kernels as K
from .resnet import BasicBlock
from .resnet_inst_matt_spconv import ResShortCut_InstMattSpconv_Dec
from ..module import SpectralNorm, conv1x1, conv3x3, ConvGRU, Marker


class ResShortCut_InstMattSpconv_BiTempSpar_Dec(ResShortCut_InstMattSpconv_Dec):
    def __init__(self, temp_method='bi', **kwargs):
        super().__init__(use_temp=True, **kwargs)
        self.temp_method = temp_method.split("_")[0]
        self.use_fusion = 'fusion' in temp_method
        self.use_temp = temp_method != 'none'
        self.os8_temp_module = ConvGRU(128, dilation=1, padding=1)
        self.diff_module = nn.Sequential(
            SpectralNorm(conv1x1(128, 64)), self._norm_layer(64), Marker('ReLU'),
            SpectralNorm(conv3x3(64, 32)), self._norm_layer(32), Marker('ReLU'),
            conv3x3(32, 1))

    def _diff(self, x, w_last=None):
        """x: (b, h, w, 128) NHWC -> fp32 logits upsampled x8: (b, 1, H, W). `w_last`: the last (plain) conv's weight, already in the kernels'
        layout (frame_diffs converts it once for all the pairs)."""
        m = self.diff_module
        dt = x.dtype
        x = MF.conv_bn_act(x, m[0].krsc(dt, x.shape[-1]), m[1], MF.ACT_RELU, 1, 1, 1, 0, 1, link_out=True)
        x = MF.conv_bn_act(x, m[3].krsc(dt, x.shape[-1]), m[4], MF.ACT_RELU, 3, 3, 1, 1, 1, link_out=True)
        x = MF.conv2d(x, MF.weight_oihw_to_krsc(m[6].weight, dt, None, 8) if w_last is None else w_last, None, 3, 3, 1, 1, 1)
        return MF.upsample_tanh(x, 1, 8, True, apply_tanh=False)

    def frame_diffs(self, feat):
        """The 2*(n_f-1) difference maps of bidirectional_fusion (:35-52), in the reference's call order (forward pairs, then
        backward pairs -- BatchNorm running statistics and SpectralNorm iterations advance per call). They depend only on the
        (detached) OS8 features, i.e. they are static-shape work: dense_stage() computes them so that they are part of the captured
        trunk graphs. feat (b, n_f, h, w, C) NHWC -> (2*(n_f-1), b, 1, H, W) fp32 logits."""
        n_f = feat.shape[1]
        pairs = [(i - 1, i) for i in range(1, n_f)] + [(i, i - 1) for i in range(n_f - 1, 0, -1)]
        # the plain last conv's weight is the same for every pair (the two SpectralNorm convs advance per call): converted once
        w_last = MF.weight_oihw_to_krsc(self.diff_module[6].weight, feat.dtype, None, 8)
        return torch.stack([self._diff(torch.cat([feat[:, a], feat[:, b]], dim=-1), w_last) for a, b in pairs], 0)

    def dense_stage(self, x, mid_fea, b, n_f, n_i, masks, gt_alphas, mem_feat=None):
        out = super().dense_stage(x, mid_fea, b, n_f, n_i, masks, gt_alphas, mem_feat)
        if n_f < 2:
            return out
        core, flag = self.split_dense_flag(out)                  # the training flag stays the LAST element
        feat = core[1]
        return core + (self.frame_diffs(feat.view(b, n_f, *feat.shape[1:]).detach()),) + ((flag,) if flag is not None else ())

    def bidirectional_fusion(self, feat, preds, diffs=None):
        """feat (b, n_f, h, w, 64) NHWC (detached); preds (b, n_f, n_i, H, W) fp32; diffs: frame_diffs(feat) when the trunk already
        computed them. The two recursions pred = prev * (1 - sigmoid(d)) + cur * sigmoid(d) (:53-67), their average (:69-78) and the
        zero-padded difference stacks are ONE HIP kernel each way (mg_bifuse_fwd / _bwd).
        -> forward_diffs, backward_diffs (b, n_f, 1, H, W) logits, fused (b, n_f, n_i, H, W), sigmoid(forward_diffs), sigmoid(backward_diffs)."""
        if diffs is None:
            diffs = self.frame_diffs(feat)
        fused, aux = MF.BiFuse.apply(preds, diffs)
        # aux = [forward logits, backward logits, their sigmoids]: the logits are `diffs` re-arranged, so gradients (the temporal losses)
        # flow through a differentiable re-arrangement of `diffs` rather than through the kernel's copy
        n_f = preds.shape[1]
        zero = torch.zeros_like(diffs[0])
        fwd = torch.stack([zero] + [diffs[i - 1] for i in range(1, n_f)], dim=1)
        bwd = torch.stack([diffs[2 * n_f - 3 - k] for k in range(n_f - 1)] + [zero], dim=1)
        return fwd, bwd, fused, aux[2], aux[3]

    def dense_modules(self):
        return super().dense_modules() + [self.os8_temp_module, self.diff_module]

    def plain_trunk_convs(self):
        return super().plain_trunk_convs() + self.os8_temp_module.plain_convs()

    def _refine_os8(self, x, masks, gt_masks, n_f, mem_feat):
        prop = partial(self.os8_temp_module.propagate_features, n_f=n_f, prev_h_state=mem_feat, temp_method=self.temp_method)
        return self.refine_OS8(x, masks, use_mask_atten=False, gt_mask=gt_masks, aggregate_mem_fn=prop)

    def detail_stage(self, dense, hw, b, n_f, n_i, plan, gt_alphas, spar_gt=None):
        """Tensors in, tensors out, static shapes, no host read (see the image decoder)."""
        x_os8, x, queries, loss_max_atten, hidden_state, fea1, fea2, fea3 = dense[:8]
        diffs = dense[8] if len(dense) > 8 else None
        h, w = hw
        mem_feat = hidden_state
        feat_os8 = x.view(b, n_f, *x.shape[1:]).detach()
        if not self.training:
            x_os8 = x_os8[:, :n_i].contiguous()
        use_gt, widths = plan['use_gt'], plan['widths']
        gt_dev = plan.get('use_gt_dev') if self.training else None      # data parallel: the guidance source is a device-side select (image decoder)
        if gt_dev is not None:
            sel = gt_dev.bool()
            detail_bits = torch.where(sel, MF.unknown_bits(gt_alphas, 30, False), MF.unknown_bits(x_os8, 30, False))
            n_cur = x_os8.shape[1]
        else:
            guided = gt_alphas if use_gt else x_os8
            if not self.training:
                x_os8 = torch.where(x_os8 >= 0.95, torch.ones_like(x_os8), x_os8)
                guided = x_os8
            n_cur = guided.shape[1]
            detail_bits = MF.unknown_bits(guided, 30, False)
        if not self.training:
            # ignore everything outside each instance's padded bounding box (:121-142): smoothing, resize, threshold, per-plane bounding box
            # and its application to the coarse alpha and the detail bit planes in four small HIP kernels (mg_temporal_crop)
            x_os8 = x_os8.contiguous()
            MF.temporal_crop_(x_os8, detail_bits, sigma=3, thr=0.1, pad=30)
            guided = x_os8
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
        ret['detail_mask'] = K.bits_unpack_u8(detail_bits, w, x_os8.shape)
        if self.use_temp:
            ret['mem_feat'] = mem_feat
        if use_gt and gt_dev is None:
            wbits4 = MF.unknown_bits(gt_alphas, 30, self.training, andmask=detail_bits, widths=widths[2])
            wbits1 = MF.unknown_bits(gt_alphas, 15, self.training, andmask=detail_bits, widths=widths[3])
        ret['weight_os4_bits'] = wbits4
        ret['weight_os1_bits'] = wbits1
        ret['detail_bits'] = detail_bits
        temp_alpha = alpha_pred.view(b, n_f, *alpha_pred.shape[1:])
        diff_forward, diff_backward, temp_fused_alpha, sig_f, sig_b = self.bidirectional_fusion(feat_os8, temp_alpha, diffs)
        if (not self.training and self.use_fusion) or self.training:
            ret['temp_alpha'] = temp_fused_alpha
            ret['diff_forward'] = sig_f
            ret['diff_backward'] = sig_b
        if self.training:
            ret['loss_max_atten'] = loss_max_atten
            ret.update(self.loss_temporal_sparsity(diff_forward, diff_backward, spar_gt))
        return ret

    def loss_temporal_sparsity(self, diff_forward, diff_backward, spar_gt):
        """:183-203 -- BCE-with-logits of the difference maps against the transition maps of frames 1.. plus the temporal-derivative loss of
        their sigmoids, x 0.25; every term a fused HIP reduction with an exact backward kernel (mg_bce_logits_*, mg_dtssd_*), on frame
        slices in place (no copies)."""
        loss = {}
        spar_gt = spar_gt.view(diff_forward.shape[0], -1, *spar_gt.shape[1:])
        tgt = spar_gt[:, 1:, 0:1]                                             # (b, n_f - 1, 1, H, W)
        bce_f = MF.bce_logits_mean(diff_forward[:, 1:], tgt)
        bce_b = MF.bce_logits_mean(diff_backward[:, :-1], tgt)
        loss['loss_temp_bce'] = bce_f + bce_b
        dt_f = MF.dtssd_loss(diff_forward[:, 1:], tgt, None, sig=True)
        dt_b = MF.dtssd_loss(diff_backward[:, :-1], tgt, None, sig=True)
        loss['loss_temp_dtssd'] = dt_f + dt_b
        loss['loss_temp'] = (loss['loss_temp_bce'] + dt_f + dt_b) * 0.25
        return loss


def res_shortcut_inst_matt_spconv_temp_22(**kwargs):
    return ResShortCut_InstMattSpconv_BiTempSpar_Dec(block=BasicBlock, layers=[2, 3, 3, 2], **kwargs)
