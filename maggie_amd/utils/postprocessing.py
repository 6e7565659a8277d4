"""Inference post-path on the device -- mirrors maggie/utils/postprocessing.py:36-64 `reverse_transform_tensor` (same signature) and
fuses the alpha snapping the reference's eval loops apply afterwards on the host (maggie/engine/test.py:139-142,229-231).

`transform_info` is the list the reference's dataloader attaches to a batch: dicts with name 'resize' (`ori_size`) / 'padding'
(`pad_size`), applied in that order on the way in and undone in reverse here. One HIP kernel (mg_postprocess_alpha) per resize."""
import torch

from .. import hip
from ..hip import c_int


def _scalar(v):
    return int(v.item()) if torch.is_tensor(v) else int(v)


def _run(x, crop_h, crop_w, out_h, out_w, snap):
    P, Hin, Win = x.shape
    out = torch.empty((P, out_h, out_w), dtype=torch.float32, device=x.device)
    hip.call('mg_postprocess_alpha', hip.ptr(x), c_int(P), c_int(Hin), c_int(Win), c_int(crop_h), c_int(crop_w), c_int(out_h), c_int(out_w),
             c_int(int(snap)), hip.ptr(out), hip.stream())
    return out


def reverse_transform_tensor(img, transform_info, snap=False):
    """img: (bs, ..., h, w) device tensor -> (bs, ..., ori_h, ori_w) fp32. snap=True additionally sets alpha <= 1/255 to 0 and
    >= 254/255 to 1 (what the reference does in numpy after the host copy)."""
    hip.need_cuda(img)
    shape = list(img.shape)
    x = img.reshape(-1, shape[-2], shape[-1]).float().contiguous()
    crop_h, crop_w = x.shape[-2:]
    ops = []
    for tr in transform_info[::-1]:
        name = tr['name'][0] if isinstance(tr['name'], list) else tr['name']
        if name == 'padding':
            ops.append(('pad',) + tuple(_scalar(v) for v in tr['pad_size']))
        elif name == 'resize':
            ops.append(('resize',) + tuple(_scalar(v) for v in tr['ori_size']))
    snapped = False
    for i, (name, a, b) in enumerate(ops):
        if name == 'pad':
            crop_h, crop_w = crop_h - a, crop_w - b
        else:
            last = snap and i == len(ops) - 1                     # the snapping rides on the last resize: no extra pass
            x = _run(x, crop_h, crop_w, a, b, last)
            crop_h, crop_w = a, b
            snapped = snapped or last
    if (snap and not snapped) or (crop_h, crop_w) != tuple(x.shape[-2:]):
        x = _run(x, crop_h, crop_w, crop_h, crop_w, snap and not snapped)      # pending crop and/or snapping (exact copy: scale 1)
    shape[-2:] = crop_h, crop_w
    return x.reshape(shape)
