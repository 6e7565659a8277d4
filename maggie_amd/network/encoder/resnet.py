"""ResNet-34-style shortcut encoder with mask-ID embedding -- mirrors maggie/network/encoder/resnet.py
(BasicBlock :7-39, ResNet_D :42-153, ResShortCut_D :155-200, ResMaskEmbedShortCut_D :202-229, factories :239-274).
Same module tree / state_dict keys; activations are NHWC and every conv+BN+ReLU runs on the HIP kernels."""
import torch
import torch.nn as nn

from ... import functional as MF
from ..module import SpectralNorm, conv1x1, conv3x3, ConvWeight, Marker


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = SpectralNorm(conv3x3(inplanes, planes, stride))
        self.bn1 = norm_layer(planes)
        self.conv2 = SpectralNorm(conv3x3(planes, planes))
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride
        self.link_out = False        # set by ResNet_D: this block's output feeds the next block's conv1 and nothing else (functional.BnLink)

    def forward(self, x):
        dt = x.dtype
        # the skip branch takes x back FROM the first conv (`carry`): in backward the skip gradient is added inside that conv's
        # data-gradient kernel instead of by a separate add over the feature map
        out, x = MF.conv_bn_act(x, self.conv1.krsc(dt, x.shape[-1]), self.bn1, MF.ACT_RELU, 3, 3, self.stride, 1, 1, carry=True, link_out=True, lazy_out=True)
        identity = x
        if self.downsample is not None:
            d = self.downsample
            if isinstance(d[0], Marker):                                     # AvgPool2d(2, stride) -> SN 1x1 -> BN
                identity = MF.avg_pool2x2(x) if self.stride == 2 else x
                identity = MF.conv_bn_act(identity, d[1].krsc(dt, identity.shape[-1]), d[2], MF.ACT_NONE, 1, 1, 1, 0, 1)
            else:
                identity = MF.conv_bn_act(x, d[0].krsc(dt, x.shape[-1]), d[1], MF.ACT_NONE, 1, 1, self.stride, 0, 1)
        return MF.conv_bn_act(out, self.conv2.krsc(dt, out.shape[-1]), self.bn2, MF.ACT_RELU, 3, 3, 1, 1, 1, res=identity, link_out=self.link_out)


class ResNet_D(nn.Module):
    def __init__(self, block, layers, norm_layer=None, late_downsample=False, is_additional_branch=False, mask_channel=0, **kwargs):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self._norm_layer = norm_layer
        self.inplanes = 64
        self.late_downsample = late_downsample
        self.midplanes = 64 if late_downsample else 32
        self.start_stride = [1, 2, 1, 2] if late_downsample else [2, 1, 2, 1]
        self.conv1 = SpectralNorm(ConvWeight(3 + mask_channel, 32, 3, self.start_stride[0], 1))
        self.conv2 = SpectralNorm(ConvWeight(32, self.midplanes, 3, self.start_stride[1], 1))
        self.conv3 = SpectralNorm(ConvWeight(self.midplanes, self.inplanes, 3, self.start_stride[2], 1))
        self.bn1 = norm_layer(32)
        self.bn2 = norm_layer(self.midplanes)
        self.bn3 = norm_layer(self.inplanes)
        self.layer1 = self._make_layer(block, 64, layers[0], stride=self.start_stride[3])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        if not is_additional_branch:
            self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
            self.layer_bottleneck = self._make_layer(block, 512, layers[3], stride=2)
        self.out_channels = {'os1': 32, 'os2': 32, 'os4': 64, 'os8': 128, 'os16': 256, 'os32': 512}
        for m in self.modules():
            if isinstance(m, ConvWeight) and hasattr(m, 'weight_bar'):
                nn.init.xavier_uniform_(m.weight_bar)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        for m in self.modules():
            if isinstance(m, BasicBlock):
                nn.init.constant_(m.bn2.weight, 0)
        self.conv1.module.weight_bar.data[:, 3:, :, :] = 0

    def _make_layer(self, block, planes, blocks, stride=1):
        if blocks == 0:
            return nn.Sequential(nn.Identity())
        norm_layer = self._norm_layer
        downsample = None
        if stride != 1:
            downsample = nn.Sequential(Marker('AvgPool2d(2,%d)' % stride), SpectralNorm(conv1x1(self.inplanes, planes * block.expansion)),
                                       norm_layer(planes * block.expansion))
        elif self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(SpectralNorm(conv1x1(self.inplanes, planes * block.expansion, stride)),
                                       norm_layer(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, norm_layer=norm_layer))
        for blk in layers[:-1]:
            blk.link_out = True          # the last block's output also feeds a shortcut branch / the ASPP: several consumers
        return nn.Sequential(*layers)


class ResShortCut_D(ResNet_D):
    def __init__(self, block, layers, num_mask=1, norm_layer=None, late_downsample=False, **kwargs):
        super().__init__(block, layers, norm_layer, late_downsample=late_downsample, mask_channel=num_mask)
        first_inplane = 3 + num_mask
        self.shortcut_inplane = [first_inplane, self.midplanes, 64, 128, 256]
        self.shortcut_plane = [32, self.midplanes, 64, 128, 256]
        self.shortcut = nn.ModuleList()
        for stage, inplane in enumerate(self.shortcut_inplane):
            self.shortcut.append(self._make_shortcut(inplane, self.shortcut_plane[stage]))

    def _make_shortcut(self, inplane, planes):
        return nn.Sequential(
            SpectralNorm(ConvWeight(inplane, planes, 3, 1, 1)), Marker('ReLU'), self._norm_layer(planes),
            SpectralNorm(ConvWeight(planes, planes, 3, 1, 1)), Marker('ReLU'), self._norm_layer(planes))

    @staticmethod
    def _run_shortcut(seq, x, carry=True):
        """-> (shortcut feature, x handed back). The tapped activation has a second consumer (the next stage of the backbone): it takes x back FROM
        the branch's first conv (`carry`), whose data-gradient kernel then adds the backbone's gradient in its epilogue instead of autograd summing
        the two with a feature-map-sized add (five of them per step, 16 MB each at the fine levels)."""
        dt = x.dtype
        if not carry:                                             # (a deferred branch: the backbone went on with x itself)
            y = MF.conv_bn_act(x, seq[0].krsc(dt, x.shape[-1]), seq[2], MF.ACT_NONE, 3, 3, 1, 1, 1, relu_before_bn=True, link_out=True, lazy_out=True)
            return MF.conv_bn_act(y, seq[3].krsc(dt, y.shape[-1]), seq[5], MF.ACT_NONE, 3, 3, 1, 1, 1, relu_before_bn=True)
        y, x = MF.conv_bn_act(x, seq[0].krsc(dt, x.shape[-1]), seq[2], MF.ACT_NONE, 3, 3, 1, 1, 1, relu_before_bn=True, link_out=True, carry=True, lazy_out=True)
        return MF.conv_bn_act(y, seq[3].krsc(dt, y.shape[-1]), seq[5], MF.ACT_NONE, 3, 3, 1, 1, 1, relu_before_bn=True), x

    def forward_features(self, x):
        """x: (N, H, W, 8) NHWC (RGB + 3 embedding channels + 2 zero pad)."""
        dt = x.dtype
        # MAGGIE_SIDE_SHORTCUTS=1 (measured slower, off by default: DESIGN.md 11.11): fea1..fea3 (the 512^2 / 256^2 / 128^2 branches) are read by the
        # detail stage only -- they are handed on as DEFERRED calls; the decoder issues them on the side stream right before the instance-token chain
        # (functional.on_side_lane), forward and -- through autograd's per-node streams -- backward. fea4 / fea5 feed the decoder directly.
        defer = bool(self.__dict__.get('defer_shortcuts')) and MF.side_lane(x.device) is not None and torch.is_grad_enabled()

        def branch(seq, t):
            if not defer:
                return self._run_shortcut(seq, t)
            return MF.Deferred(lambda: self._run_shortcut(seq, t, carry=False), t), t

        fea1, x = branch(self.shortcut[0], x)
        out = MF.conv_bn_act(x, self.conv1.krsc(dt, 8), self.bn1, MF.ACT_RELU, 3, 3, self.start_stride[0], 1, 1, link_out=True, lazy_out=True)
        x1 = MF.conv_bn_act(out, self.conv2.krsc(dt), self.bn2, MF.ACT_RELU, 3, 3, self.start_stride[1], 1, 1)       # x1 also feeds shortcut[1]
        fea2, x1 = branch(self.shortcut[1], x1)
        out = MF.conv_bn_act(x1, self.conv3.krsc(dt), self.bn3, MF.ACT_RELU, 3, 3, self.start_stride[2], 1, 1, link_out=True)
        x2 = self.layer1(out)
        fea3, x2 = branch(self.shortcut[2], x2)
        x3 = self.layer2(x2)
        fea4, x3 = self._run_shortcut(self.shortcut[3], x3)
        x4 = self.layer3(x3)
        fea5, x4 = self._run_shortcut(self.shortcut[4], x4)
        out = self.layer_bottleneck(x4)
        return out, {'shortcut': (fea1, fea2, fea3, fea4, fea5), 'backbone_feat': (x2, x3, x4, out)}


class ResMaskEmbedShortCut_D(ResShortCut_D):
    def __init__(self, block, layers, num_mask=1, num_embed=1, norm_layer=None, late_downsample=False, **kwargs):
        super().__init__(block, layers, num_embed, norm_layer, late_downsample=late_downsample, **kwargs)
        self.num_embed = num_embed
        if self.num_embed > 0:
            self.mask_embed_layer = nn.Embedding(num_mask + 1, num_embed)

    def forward(self, image, masks, **kwargs):
        """image (N,3,H,W) fp32 NCHW; masks (N,num_mask,Hm,Wm) fp32 at any integer down-scale of (H,W)
        (the nearest up-scaling of arch/maggie.py:176-178 is folded into the packing kernel)."""
        x = MF.mask_embed(image, masks, self.mask_embed_layer.weight, MF.compute_dtype())
        out, mid = self.forward_features(x)
        mid['image'] = image
        return out, mid


def res_shortcut_embed_29(**kwargs):
    return ResMaskEmbedShortCut_D(BasicBlock, [3, 4, 4, 2], **kwargs)


def res_shortcut_29(**kwargs):
    raise NotImplementedError('only res_shortcut_embed_29 (configs/maggie_{image,video}.yaml) is built')
