"""ConvGRU -- mirrors maggie/network/module/conv_gru.py:4-69 (per-frame GRU over OS8 features, 'bi' = second pass over
the reversed clip, averaged). The two 3x3 gate convolutions run on the implicit-GEMM HIP kernel (bias in the epilogue)."""
import torch
from torch import nn

from ... import functional as MF
from .base import ConvWeight


class ConvGRU(nn.Module):
    def __init__(self, channels, dilation=1, kernel_size=3, padding=1):
        super().__init__()
        self.channels = channels
        self.ih = nn.Sequential(ConvWeight(channels * 2, channels * 2, kernel_size, 1, padding, dilation, bias=True), nn.Sigmoid())
        self.hh = nn.Sequential(ConvWeight(channels * 2, channels, kernel_size, 1, padding, dilation, bias=True), nn.Tanh())

    def _conv(self, holder, x):
        c = holder[0]
        w = MF.weight_oihw_to_krsc(c.weight, x.dtype)
        return MF.conv2d(x, w, c.bias.float(), c.kernel_size, c.kernel_size, 1, c.padding, c.dilation)

    def forward_single_frame(self, x, h):
        """x, h: (b, H, W, C) NHWC."""
        rz = torch.sigmoid(self._conv(self.ih, torch.cat([x, h], -1)))
        r, z = rz.split(self.channels, dim=-1)
        c = torch.tanh(self._conv(self.hh, torch.cat([x, r * h], -1)))
        h = (1 - z) * h + z * c
        return h, h

    def forward_time_series(self, x, h):
        o = []
        for t in range(x.shape[1]):
            ot, h = self.forward_single_frame(x[:, t], h)
            o.append(ot)
        o = torch.stack(o, dim=1)
        return o, o

    def forward(self, x, h):
        if h is None:
            h = torch.zeros((x.size(0), x.size(-3), x.size(-2), x.size(-1)), device=x.device, dtype=x.dtype)
        if x.ndim == 5:
            return self.forward_time_series(x, h)
        return self.forward_single_frame(x, h)

    def propagate_features(self, feat, n_f, prev_h_state=None, temp_method='none'):
        """feat: (b, n_f, H, W, C) NHWC."""
        hidden_state = None
        if temp_method == 'none':
            all_x = []
            for j in range(n_f):
                o, hidden_state = self.forward(x=feat[:, j], h=None)
                all_x.append(o)
            feat = torch.stack(all_x, dim=1)
        else:
            feat_forward, hidden_state = self.forward(x=feat, h=prev_h_state)
            if temp_method == 'bi':
                feat_backward, _ = self.forward(x=torch.flip(feat[:, :-1], dims=(1,)), h=hidden_state[:, -1])
                feat_backward = torch.flip(feat_backward, dims=(1,))
                feat = torch.cat([(feat_forward[:, :-1] + feat_backward) / 2, feat_forward[:, -1:]], 1)
            else:
                feat = feat_forward
        return feat, hidden_state
