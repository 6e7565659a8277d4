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
from ..loss import loss_dtSSD
from ..module import SpectralNorm, conv1x1, conv3x3, ConvGRU, Marker


def gaussian_smoothing(x, sigma):
    """maggie/utils/utils.py:61-83 on fp32 planes (N, C, H, W), including its kernel quirk (g*g broadcast over rows)."""
    ks = sigma * 2 + 1
    pad = ks // 2
    xp = F.pad(x, (pad, pad, pad, pad), mode='constant', value=0)
    grid = torch.arange(ks, device=x.device).float() - ks // 2
    g = torch.exp(-grid ** 2 / (2 * sigma ** 2))
    g = g / g.sum()
    k = (g.view(1, 1, -1) * g.view(1, 1, -1)).expand(x.shape[1], 1, ks, ks).type_as(x)
    sm = F.conv2d(xp, k, stride=1, padding=0, groups=x.shape[1])
    sm = sm[:, :, pad:-pad, pad:-pad]
    return F.interpolate(sm, size=x.shape[-2:], mode='bilinear', align_corners=False)


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

    def _diff(self, x):
        """x: (b, h, w, 128) NHWC -> fp32 logits upsampled x8: (b, 1, H, W)."""
        m = self.diff_module
        dt = x.dtype
        x = MF.conv_bn_act(x, m[0].krsc(dt, x.shape[-1]), m[1], MF.ACT_RELU, 1, 1, 1, 0, 1)
        x = MF.conv_bn_act(x, m[3].krsc(dt, x.shape[-1]), m[4], MF.ACT_RELU, 3, 3, 1, 1, 1)
        x = MF.conv2d(x, MF.weight_oihw_to_krsc(m[6].weight, dt, None, 8), None, 3, 3, 1, 1, 1)
        return MF.upsample_tanh(x, 1, 8, True, apply_tanh=False)

    def frame_diffs(self, feat):
        """The 2*(n_f-1) difference maps of bidirectional_fusion (:35-52), in the reference's call order (forward pairs, then
        backward pairs -- BatchNorm running statistics and SpectralNorm iterations advance per call). They depend only on the
        (detached) OS8 features, i.e. they are static-shape work: dense_stage() computes them so that they are part of the captured
        trunk graphs. feat (b, n_f, h, w, C) NHWC -> (2*(n_f-1), b, 1, H, W) fp32 logits."""
        n_f = feat.shape[1]
        pairs = [(i - 1, i) for i in range(1, n_f)] + [(i, i - 1) for i in range(n_f - 1, 0, -1)]
        return torch.stack([self._diff(torch.cat([feat[:, a], feat[:, b]], dim=-1)) for a, b in pairs], 0)

    def dense_stage(self, x, mid_fea, b, n_f, n_i, masks, gt_alphas, mem_feat=None):
        out = super().dense_stage(x, mid_fea, b, n_f, n_i, masks, gt_alphas, mem_feat)
        if n_f < 2:
            return out
        feat = out[1]
        return tuple(out) + (self.frame_diffs(feat.view(b, n_f, *feat.shape[1:]).detach()),)

    def bidirectional_fusion(self, feat, preds, diffs=None):
        """feat (b, n_f, h, w, 64) NHWC (detached); preds (b, n_f, n_i, H, W) fp32; diffs: frame_diffs(feat) when the trunk already
        computed them."""
        n_f = feat.shape[1]
        if diffs is None:
            diffs = self.frame_diffs(feat)
        forward_diffs, backward_diffs = [], []
        forward_preds, backward_preds = [preds[:, 0]], [preds[:, n_f - 1]]
        for i in range(1, n_f):
            diff = diffs[i - 1]
            forward_diffs.append(diff)
            forward_preds.append(forward_preds[-1] * (1 - diff.sigmoid()) + preds[:, i] * diff.sigmoid())
        forward_diffs = torch.stack([torch.zeros_like(forward_diffs[0])] + forward_diffs, dim=1)
        for j, i in enumerate(range(n_f - 1, 0, -1)):
            diff = diffs[n_f - 1 + j]
            backward_diffs.append(diff)
            backward_preds.append(backward_preds[-1] * (1 - diff.sigmoid()) + preds[:, i - 1] * diff.sigmoid())
        backward_preds = backward_preds[::-1]
        backward_diffs = backward_diffs[::-1]
        backward_diffs = torch.stack(backward_diffs + [torch.zeros_like(backward_diffs[-1])], dim=1)
        fuse_preds = []
        for i in range(n_f):
            if i == 0:
                fuse_preds.append(forward_preds[i])
            elif i == n_f - 1:
                fuse_preds.append(backward_preds[i])
            else:
                fuse_preds.append((forward_preds[i] + backward_preds[i]) / 2)
        return forward_diffs, backward_diffs, torch.stack(fuse_preds, dim=1)

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
        guided = gt_alphas if use_gt else x_os8
        if not self.training:
            x_os8 = torch.where(x_os8 >= 0.95, torch.ones_like(x_os8), x_os8)
            guided = x_os8
        n_cur = guided.shape[1]
        detail_bits = MF.unknown_bits(guided, 30, False)
        if not self.training:
            # ignore everything outside each instance's padded bounding box (:121-142), on device without host loops
            H, W = hw
            smooth = gaussian_smoothing(x_os8, 3) > 0.1                              # (N, n_i, H, W) bool
            rows = smooth.any(-1)
            cols = smooth.any(-2)
            ar_h = torch.arange(H, device=x.device)
            ar_w = torch.arange(W, device=x.device)
            big = 10 ** 6
            y_min = torch.where(rows, ar_h, big).amin(-1)
            y_max = torch.where(rows, ar_h, -big).amax(-1)
            x_min = torch.where(cols, ar_w, big).amin(-1)
            x_max = torch.where(cols, ar_w, -big).amax(-1)
            has = rows.any(-1)
            y0 = (y_min - 30).clamp(min=0)
            y1 = (y_max + 30).clamp(max=H)
            x0 = (x_min - 30).clamp(min=0)
            x1 = (x_max + 30).clamp(max=W)
            box = ((ar_h[None, None, :, None] >= y0[..., None, None]) & (ar_h[None, None, :, None] < y1[..., None, None])
                   & (ar_w[None, None, None, :] >= x0[..., None, None]) & (ar_w[None, None, None, :] < x1[..., None, None]))
            box = box | ~has[..., None, None]                                        # `continue` when the instance is empty
            x_os8 = x_os8 * box
            detail_bits = detail_bits & K.bits_pack(box.to(torch.uint8).contiguous(), mode=1)
            guided = x_os8
        x_os4, x_os1, detail_bits = self.process_os4_os1(x, b, n_f, fea1, fea2, fea3, hw, x_os8, queries, n_cur, detail_bits)
        ret = {'alpha_os1': x_os1, 'alpha_os4': x_os4, 'alpha_os8': x_os8}
        alpha_pred, weight_os4, weight_os1 = self.fuse(ret, detail_bits, widths)
        ret['refined_masks'] = alpha_pred
        ret['detail_mask'] = K.bits_unpack_u8(detail_bits, w, x_os8.shape)
        if self.use_temp:
            ret['mem_feat'] = mem_feat
        if use_gt:
            weight_os4 = K.bits_unpack_u8(MF.unknown_bits(gt_alphas, 30, self.training, andmask=detail_bits, widths=widths[2]), w, x_os8.shape)
            weight_os1 = K.bits_unpack_u8(MF.unknown_bits(gt_alphas, 15, self.training, andmask=detail_bits, widths=widths[3]), w, x_os8.shape)
        ret['weight_os4'] = weight_os4
        ret['weight_os1'] = weight_os1
        temp_alpha = alpha_pred.view(b, n_f, *alpha_pred.shape[1:])
        diff_forward, diff_backward, temp_fused_alpha = self.bidirectional_fusion(feat_os8, temp_alpha, diffs)
        if (not self.training and self.use_fusion) or self.training:
            ret['temp_alpha'] = temp_fused_alpha
            ret['diff_forward'] = diff_forward.sigmoid()
            ret['diff_backward'] = diff_backward.sigmoid()
        if self.training:
            ret['loss_max_atten'] = loss_max_atten
            ret.update(self.loss_temporal_sparsity(diff_forward, diff_backward, spar_gt))
        return ret

    def loss_temporal_sparsity(self, diff_forward, diff_backward, spar_gt):
        loss = {}
        spar_gt = spar_gt.view(diff_forward.shape[0], -1, *spar_gt.shape[1:])
        bce_f = F.binary_cross_entropy_with_logits(diff_forward[:, 1:, 0], spar_gt[:, 1:, 0], reduction='mean')
        bce_b = F.binary_cross_entropy_with_logits(diff_backward[:, :-1, 0], spar_gt[:, 1:, 0], reduction='mean')
        loss['loss_temp_bce'] = bce_f + bce_b
        ones = torch.ones_like(spar_gt[:, 1:, 0:1])
        dt_f = loss_dtSSD(diff_forward[:, 1:].sigmoid(), spar_gt[:, 1:, 0:1], ones)
        dt_b = loss_dtSSD(diff_backward[:, :-1].sigmoid(), spar_gt[:, 1:, 0:1], ones)
        loss['loss_temp_dtssd'] = dt_f + dt_b
        loss['loss_temp'] = (loss['loss_temp_bce'] + dt_f + dt_b) * 0.25
        return loss


def res_shortcut_inst_matt_spconv_temp_22(**kwargs):
    return ResShortCut_InstMattSpconv_BiTempSpar_Dec(block=BasicBlock, layers=[2, 3, 3, 2], **kwargs)
