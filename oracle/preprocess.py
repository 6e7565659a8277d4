"""TEST INFRASTRUCTURE ONLY (imported by tests/ and tests/golden/make_golden.py, never by the product).

CPU restatement of the input-side tensor work of the reference's DataLoader:
  normalize_frames : ToTensor + Normalize.norm      maggie/dataloader/transforms.py:720-778
  scale_planes     : alphas[alphas < 5] = 0 (:744); alpha/255, mask/255, slot scatter, nearest mask downscale
                     maggie/dataloader/him.py:157-173
PINNED by tests/golden/preprocess_pinned.npz: outputs of the reference's own ToTensor/Normalize classes and of the him.py
statements run verbatim on seeded uint8 inputs (tests/golden/make_golden.py: preprocess_fixture)."""
import numpy as np


def normalize_frames(frames_u8, mean, std):
    x = frames_u8.astype(np.float32)
    x = np.moveaxis(x, -1, -3)                                    # (..., H, W, 3) -> (..., 3, H, W)
    x = x / np.float32(255.0)
    m = np.asarray(mean, np.float32).reshape(3, 1, 1)
    s = np.asarray(std, np.float32).reshape(3, 1, 1)
    return ((x - m) / s).astype(np.float32)


def scale_planes(planes_u8, n_slots=None, slot_ids=None, out_size=None, thresh=0):
    F_, n_i, H, W = planes_u8.shape
    n_slots = n_i if n_slots is None else n_slots
    p = planes_u8.copy()
    p[p < thresh] = 0
    v = p.astype(np.float32) / np.float32(255.0)
    if out_size is not None and tuple(out_size) != (H, W):
        Ho, Wo = out_size
        sy = np.minimum(np.floor(np.arange(Ho, dtype=np.float32) * (np.float32(H) / np.float32(Ho))).astype(np.int64), H - 1)
        sx = np.minimum(np.floor(np.arange(Wo, dtype=np.float32) * (np.float32(W) / np.float32(Wo))).astype(np.int64), W - 1)
        v = v[:, :, sy][:, :, :, sx]
    out = np.zeros((F_, n_slots) + v.shape[2:], np.float32)
    ids = list(range(n_i)) if slot_ids is None else list(slot_ids)
    out[:, ids] = v
    return out
