"""ORACLE-ONLY stand-in for the third-party `spconv.pytorch` API surface the reference calls
(/root/reference/maggie/network/decoder/resnet_inst_matt_spconv.py:61-130,168,184,214,225,248,265).

spconv (CUDA library, un-pinned in the reference's requirements.txt:4) is not installed here. This is a
*dense-masked* PyTorch-CPU restatement of its published semantics, written independently of the
gather-table restatement in oracle/refmodel.py so that the two can be cross-checked:

  SubMConv2d            dense conv of the zero-filled map, read back at the input's active sites
                        (output sites == input sites; `padding` is ignored, the kernel is centre-aligned)
  SparseConv2d(k,s,p)   output site active iff any active input lies in its window; features = dense conv
  SparseInverseConv2d   reuses the (in -> out) pairs stored under `indice_key` by the SparseConv2d with the
                        SAME kernel offset (no flip): == conv_transpose2d of the zero-filled coarse map,
                        read back at the fine active sites
  weights               (Cout, kh, kw, Cin)  ("KRSC", spconv >= 2.2)

It is used only inside this container by tests/golden/make_golden.py to run the reference's own glue code
end to end. It never ships in the product and is never measured. Parity vs the real spconv: UNPINNED.
"""
import math

import torch
from torch import nn
import torch.nn.functional as F


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, indice_dict=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = list(spatial_shape)
        self.batch_size = batch_size
        self.indice_dict = {} if indice_dict is None else indice_dict

    def replace_feature(self, feature):
        return SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.indice_dict)

    def dense(self):
        H, W = self.spatial_shape
        out = self.features.new_zeros((self.batch_size, H, W, self.features.shape[1]))
        idx = self.indices.long()
        out = out.index_put((idx[:, 0], idx[:, 1], idx[:, 2]), self.features)
        return out.permute(0, 3, 1, 2).contiguous()

    def active_map(self):
        H, W = self.spatial_shape
        m = torch.zeros((self.batch_size, H, W), dtype=torch.bool)
        idx = self.indices.long()
        m[idx[:, 0], idx[:, 1], idx[:, 2]] = True
        return m


class _SparseConvBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, kernel_size, kernel_size, in_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter('bias', None)

    def _oihw(self):
        return self.weight.permute(0, 3, 1, 2)

    @staticmethod
    def _gather(dense, indices):
        idx = indices.long()
        return dense.permute(0, 2, 3, 1)[idx[:, 0], idx[:, 1], idx[:, 2]]


class SubMConv2d(_SparseConvBase):
    def forward(self, x):
        y = F.conv2d(x.dense(), self._oihw(), self.bias, 1, self.kernel_size // 2)
        return x.replace_feature(self._gather(y, x.indices))


class SparseConv2d(_SparseConvBase):
    def forward(self, x):
        k, s, p = self.kernel_size, self.stride, self.padding
        y = F.conv2d(x.dense(), self._oihw(), self.bias, s, p)
        act = F.max_pool2d(x.active_map().float()[:, None], k, s, p)[:, 0] > 0
        out_idx = torch.nonzero(act).int()                          # row-major sorted (batch, y, x)
        out = SparseConvTensor(self._gather(y, out_idx), out_idx, list(y.shape[-2:]), x.batch_size, x.indice_dict)
        if self.indice_key is not None:
            out.indice_dict[self.indice_key] = dict(in_indices=x.indices, in_shape=list(x.spatial_shape),
                                                    kernel=k, stride=s, padding=p)
        return out


class SparseInverseConv2d(_SparseConvBase):
    def forward(self, x):
        rec = x.indice_dict[self.indice_key]
        k, s, p = rec['kernel'], rec['stride'], rec['padding']
        Hf, Wf = rec['in_shape']
        Hc, Wc = x.spatial_shape
        oph = Hf - ((Hc - 1) * s - 2 * p + k)
        opw = Wf - ((Wc - 1) * s - 2 * p + k)
        w = self.weight.permute(3, 0, 1, 2)                          # (Cin, Cout, kh, kw), same offsets, no flip
        y = F.conv_transpose2d(x.dense(), w, self.bias, s, p, (oph, opw))
        out = SparseConvTensor(self._gather(y, rec['in_indices']), rec['in_indices'], [Hf, Wf], x.batch_size,
                               x.indice_dict)
        return out


class SparseSequential(nn.Sequential):
    def forward(self, x):
        for m in self:
            if isinstance(m, _SparseConvBase):
                x = m(x)
            else:
                if x.features.shape[0] > 0:
                    x = x.replace_feature(m(x.features))
        return x
