"""ConvGRU over the OS8 feature clip (maggie/network/module/conv_gru.py:4-69; same parameters: `ih.0`, `hh.0` conv weights).

Per frame:  [r | z] = sigmoid(W_ih * [x, h]);  c = tanh(W_hh * [x, r.h]);  h' = (1 - z).h + z.c.
The two 3x3 gate convolutions run on the implicit-GEMM HIP kernel (bias in the epilogue); everything between them is two fused
HIP kernels (functional.GruGate / GruOut: sigmoid, gating, concatenation, tanh, blend -- forward and backward).
Tensors are NHWC: x, h (b, H, W, C); clips (b, n_f, H, W, C)."""
import torch
from torch import nn

from ... import functional as MF
from .base import ConvWeight


class ConvGRU(nn.Module):
    def __init__(self, channels, dilation=1, kernel_size=3, padding=1):
        super().__init__()
        self.channels = channels
        gate = lambda cout: ConvWeight(channels * 2, cout, kernel_size, 1, padding, dilation, bias=True)      # noqa: E731
        self.ih = nn.Sequential(gate(channels * 2), nn.Sigmoid())        # activations live in the fused kernels; kept for the
        self.hh = nn.Sequential(gate(channels), nn.Tanh())               # reference's module / state_dict layout

    def _gate_conv(self, seq, inp):
        conv = seq[0]
        fan = self.__dict__.get('_bias_fans', {}).get(id(conv))          # the bias serves every cell of the clip: one alias per cell (functional.Fan)
        bias = MF.take(fan) if fan is not None else conv.bias.float()
        return MF.conv2d(inp, MF.plain_krsc(conv, inp.dtype, keep=True), bias, conv.kernel_size, conv.kernel_size, 1,
                         conv.padding, conv.dilation)                    # keep: one cell per frame and direction, same weights

    def _fan_biases(self, cells):
        """`cells` gate evaluations follow: their bias gradients meet in ONE launch each (was cells - 1 autograd adds per bias)."""
        self.__dict__['_bias_fans'] = {id(c): MF.Fan(c.bias.float(), cells) for c in (self.ih[0], self.hh[0])} if cells > 2 else {}

    def plain_convs(self):
        return [self.ih[0], self.hh[0]]

    def cell(self, x, h):
        rz = self._gate_conv(self.ih, torch.cat((x, h), -1))             # pre-activation [r | z]
        cand = self._gate_conv(self.hh, MF.GruGate.apply(rz, x, h))      # pre-activation candidate from [x | r.h]
        return MF.GruOut.apply(rz, cand, h)

    def run_frames(self, frames, h):
        """frames: sequence of (b, H, W, C) -> list of hidden states, one per frame."""
        states = []
        for x in frames:
            h = self.cell(x, h)
            states.append(h)
        return states

    def run(self, clip, h):
        """clip (b, n, H, W, C) -> all hidden states (b, n, H, W, C)."""
        return torch.stack(self.run_frames(clip.unbind(1), h), 1)

    def forward(self, x, h):
        """Single frame (b, H, W, C) -> (h', h'); clip (b, n, H, W, C) -> (states, states), like the reference."""
        if h is None:
            h = x.new_zeros(x.shape[:1] + x.shape[-3:])
        if x.dim() == 5:
            out = self.run(x, h)
            return out, out
        h = self.cell(x, h)
        return h, h

    # reference-compatible aliases
    def forward_single_frame(self, x, h):
        h = self.cell(x, h)
        return h, h

    def forward_time_series(self, x, h):
        out = self.run(x, h)
        return out, out

    def propagate_features(self, feat, n_f, prev_h_state=None, temp_method='none'):
        """feat (b, n_f, H, W, C). 'none': every frame on its own; otherwise a forward pass over the clip, and for 'bi' a second
        pass over the reversed clip (seeded with the last forward state) averaged into all frames but the last."""
        if temp_method == 'none':
            frames = [self.forward(feat[:, j], None) for j in range(n_f)]
            return torch.stack([f[0] for f in frames], 1), frames[-1][1]
        # frames leave the clip through ONE unbind (its backward is one stack; n selects cost a fill, a copy and an accumulation add each) and
        # the reversed pass runs over the frame list backwards -- no flipped copies of the clip or of its states
        frames = feat.unbind(1)
        self._fan_biases(len(frames) + (len(frames) - 1 if temp_method == 'bi' else 0))
        states = self.run_frames(frames, prev_h_state if prev_h_state is not None else feat.new_zeros(feat.shape[:1] + feat.shape[-3:]))
        fwd = torch.stack(states, 1)
        if temp_method != 'bi':
            self.__dict__['_bias_fans'] = {}
            return fwd, fwd
        rev = self.run_frames(frames[-2::-1], states[-1])         # rev[k] belongs to frame n - 2 - k
        self.__dict__['_bias_fans'] = {}
        # every frame but the last is the mean of the two passes; the last one pairs with itself: (f + f) * 0.5 == f bit for bit
        return (fwd + torch.stack(rev[::-1] + [states[-1]], 1)) * 0.5, fwd
