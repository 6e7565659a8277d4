"""Input side of the hot path on the device -- the tensor work the reference's DataLoader does on the CPU after
decoding an item, fed from uint8 buffers instead:

  * `ToTensor` + `Normalize` (maggie/dataloader/transforms.py:720-778): frames (T, H, W, 3) uint8 -> (T, 3, H, W) fp32,
    `/ 255`, `(x - mean) / std`; alphas below 5 are zeroed (`alphas[alphas < 5] = 0`, :744);
  * item assembly of `HIMDataset.__getitem__` (maggie/dataloader/him.py:157-173): `alpha / 255`, `mask / 255`, scatter of the
    real instances into `max_inst` slots (`chosen_ids`), nearest downscale of the masks to (H // 8, W // 8).

The caller keeps the host logic (file decoding, augmentation, which instance goes to which slot); what reaches the GPU is uint8 --
4x fewer PCIe bytes than the fp32 `image/alpha/mask` tensors of the reference, and no fp32 `fg`/`bg` tensors at all (the model
never reads them). Two HIP kernels (mg_preprocess_image, mg_preprocess_planes), bit-exact against the reference's arithmetic."""
import numpy as np
import torch

from .. import hip
from ..hip import c_int, c_long

IMAGENET_MEAN = (0.485, 0.456, 0.406)        # maggie/dataloader/him.py: T.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
IMAGENET_STD = (0.229, 0.224, 0.225)


def _u8(x, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if x.dtype != torch.uint8:
        raise TypeError('expected uint8 pixels, got %s' % x.dtype)
    return x.to(device, non_blocking=True).contiguous()


def normalize_frames(frames_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD, device=None):
    """(..., H, W, 3) uint8 -> (..., 3, H, W) fp32 normalised (ToTensor + Normalize.norm)."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    x = _u8(frames_u8, device)
    hip.need_cuda(x)
    *lead, H, W, C = x.shape
    if C != 3:
        raise ValueError('frames must be (..., H, W, 3)')
    n = int(np.prod(lead)) if lead else 1
    out = torch.empty(tuple(lead) + (3, H, W), dtype=torch.float32, device=device)
    m = (hip.ctypes.c_float * 3)(*mean)
    s = (hip.ctypes.c_float * 3)(*std)
    for f0 in range(0, n, 65535):
        f1 = min(n, f0 + 65535)
        hip.call('mg_preprocess_image', hip.ctypes.c_void_p(x.data_ptr() + f0 * H * W * 3), hip.ctypes.c_void_p(out.data_ptr() + f0 * H * W * 12),
                 m, s, c_long(f1 - f0), c_long(H * W), hip.stream())
    return out


def scale_planes(planes_u8, n_slots=None, slot_ids=None, out_size=None, thresh=0, device=None):
    """(F, n_i, H, W) uint8 -> (F, n_slots, Ho, Wo) fp32 = v / 255 (0 below `thresh`), plane j of every frame written to slot
    slot_ids[j] (default: identity, n_slots = n_i), other slots zero; (Ho, Wo) != (H, W): F.interpolate(mode='nearest')."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    x = _u8(planes_u8, device)
    hip.need_cuda(x)
    F_, n_i, H, W = x.shape
    n_slots = n_i if n_slots is None else int(n_slots)
    Ho, Wo = (H, W) if out_size is None else (int(out_size[0]), int(out_size[1]))
    table = None
    if slot_ids is not None:
        ids = [int(i) for i in slot_ids]
        if len(ids) != n_i or len(set(ids)) != n_i or min(ids) < 0 or max(ids) >= n_slots:
            raise ValueError('slot_ids must name %d distinct slots below %d' % (n_i, n_slots))
        src = np.full((n_slots,), -1, np.int32)
        src[ids] = np.arange(n_i, dtype=np.int32)
        table = torch.from_numpy(np.tile(src, F_)).to(device, non_blocking=True)
    elif n_slots != n_i:
        raise ValueError('n_slots != n_i needs slot_ids')
    out = torch.empty((F_, n_slots, Ho, Wo), dtype=torch.float32, device=device)
    per = max(1, 65535 // n_slots)
    for f0 in range(0, F_, per):
        f1 = min(F_, f0 + per)
        hip.call('mg_preprocess_planes', hip.ctypes.c_void_p(x.data_ptr() + f0 * n_i * H * W), hip.ctypes.c_void_p(out.data_ptr() + f0 * n_slots * Ho * Wo * 4),
                 hip.ptr(None if table is None else table[f0 * n_slots:]), c_int(f1 - f0), c_int(n_i), c_int(n_slots), c_int(H), c_int(W),
                 c_int(Ho), c_int(Wo), c_int(int(thresh)), hip.stream())
    return out


class DevicePreprocessor:
    """frames / alphas / masks of ONE item as uint8 -> the `image`, `alpha`, `mask` entries of the reference's item dict
    (him.py:175-181), on the device. `slot_ids`: the `chosen_ids` of him.py:161 (training pads the instances to `max_inst` slots);
    None keeps the instances where they are (evaluation)."""

    def __init__(self, max_inst=10, downscale_mask=True, mean=IMAGENET_MEAN, std=IMAGENET_STD, device=None):
        self.max_inst, self.downscale_mask, self.mean, self.std, self.device = max_inst, downscale_mask, mean, std, device

    def __call__(self, frames_u8, alphas_u8=None, masks_u8=None, slot_ids=None):
        out = {'image': normalize_frames(frames_u8, self.mean, self.std, self.device)}
        T, _, H, W = out['image'].shape
        n_slots = self.max_inst if slot_ids is not None else None
        if alphas_u8 is not None:
            a = alphas_u8.reshape(T, -1, H, W)
            out['alpha'] = scale_planes(a, n_slots, slot_ids, None, 5, self.device)                      # transforms.py:744
        if masks_u8 is not None:
            m = masks_u8.reshape(T, -1, H, W)
            size = (H // 8, W // 8) if self.downscale_mask else None                                   # him.py:172-173
            out['mask'] = scale_planes(m, n_slots, slot_ids, size, 0, self.device)
        return out
