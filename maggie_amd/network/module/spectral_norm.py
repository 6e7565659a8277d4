"""SpectralNorm wrapper -- mirrors maggie/network/module/spectral_norm.py:9-80 (state: module.weight_bar /
weight_u / weight_v; ONE power iteration on EVERY forward, train and eval, written back into u and v)."""
import torch
from torch import nn
from torch.nn import Parameter

from ... import functional as MF


def l2normalize(v, eps=1e-12):
    return v / (v.norm() + eps)


class SpectralNorm(nn.Module):
    def __init__(self, module, name='weight', power_iterations=1):
        super().__init__()
        self.module = module
        self.name = name
        self.power_iterations = power_iterations
        self._prepared = None            # set by functional.spectral_norm_prepare (batched path), consumed by krsc()
        if not hasattr(module, name + '_bar'):
            w = getattr(module, name)
            height = w.data.shape[0]
            width = w.view(height, -1).data.shape[1]
            u = Parameter(l2normalize(w.data.new(height).normal_(0, 1)), requires_grad=False)
            v = Parameter(l2normalize(w.data.new(width).normal_(0, 1)), requires_grad=False)
            w_bar = Parameter(w.data)
            del module._parameters[name]
            module.register_parameter(name + '_u', u)
            module.register_parameter(name + '_v', v)
            module.register_parameter(name + '_bar', w_bar)

    def normalized_weight(self):
        """W_bar / sigma with sigma = u^T W v after one power iteration (u, v updated in place, no grad through them)."""
        m = self.module
        w, u, v = m.weight_bar, m.weight_u, m.weight_v
        height = w.shape[0]
        with torch.no_grad():
            wm = w.detach().reshape(height, -1)
            for _ in range(self.power_iterations):
                nv = l2normalize(torch.mv(wm.t(), u.data))
                nu = l2normalize(torch.mv(wm, nv))
                v.data = nv
                u.data = nu
        sigma = u.data.dot(w.reshape(height, -1).mv(v.data))
        return w / sigma

    def krsc(self, dtype, cin_pad=None):
        """Spectrally-normalised weight in the kernels' (Cout, taps, Cin_pad) layout and compute dtype (fused HIP path:
        power iteration + sigma + scaling + layout/dtype conversion; u and v are updated in place)."""
        m = self.module
        cin = m.in_channels
        pad_in = MF.pad8(cin) if cin_pad is None else cin_pad
        assert self.power_iterations == 1
        w = self._prepared
        if w is not None:
            self._prepared = None
            if w.dtype == dtype and w.shape[-1] == pad_in:
                return w
        return MF.spectral_norm_weight(m.weight_bar, m.weight_u, m.weight_v, m.transposed, dtype, pad_in)

    def forward(self, x, **kw):
        m = self.module
        w = self.krsc(x.dtype, x.shape[-1])
        return MF.conv2d(x, w, None, m.kernel_size, m.kernel_size, m.stride, m.padding, m.dilation, m.transposed, **kw)
