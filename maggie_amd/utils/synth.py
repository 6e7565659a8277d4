"""Portable deterministic generators for weights and synthetic batches.

No checkpoint or dataset is reachable offline, so parity tests and bench.py use
  * weights filled *by parameter name* from ``np.random.RandomState`` -- regenerable bit-identically on the
    CPU oracle side and on the GPU product side, independent of construction order or torch RNG;
  * synthetic batches following the reference's batch contract (`/root/reference/maggie/dataloader/him.py:157-191`):
    image N(0,1) (already "normalised"), per-instance filled ellipses as masks (rasterised at 1/8 resolution,
    `him.py:175-176`), soft-edged ellipses as alphas, transition = dilated (0 < alpha < 1).
"""
import zlib

import numpy as np
import torch


def _rs(name, seed):
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def fill_state_dict_(sd, seed=1234):
    """In-place deterministic fill of a reference-layout state_dict (values depend only on key name, shape, seed)."""
    with torch.no_grad():
        for name in sorted(sd.keys()):
            t = sd[name]
            rs = _rs(name, seed)
            shape = tuple(t.shape)
            leaf = name.rsplit('.', 1)[-1]
            if leaf == 'num_batches_tracked':
                t.zero_()
                continue
            if leaf == 'running_mean':
                v = rs.normal(0.0, 0.1, shape)
            elif leaf == 'running_var':
                v = rs.uniform(0.5, 1.5, shape)
            elif leaf in ('weight_u', 'weight_v'):
                v = rs.normal(0.0, 1.0, shape)
                v = v / (np.linalg.norm(v) + 1e-12)
            elif t.dim() <= 1:
                if leaf == 'weight':            # BN / LayerNorm gains
                    v = rs.uniform(0.5, 1.5, shape)
                else:                           # biases
                    v = rs.normal(0.0, 0.05, shape)
            elif 'embed' in name or 'query_feat' in name:
                v = rs.normal(0.0, 0.5, shape)
            else:
                fan_in = int(np.prod(shape[1:]))
                v = rs.normal(0.0, 1.0, shape) * np.sqrt(2.0 / max(fan_in, 1))
            t.copy_(torch.from_numpy(np.asarray(v, np.float32)).reshape(shape).to(t.dtype))
    return sd


def _ellipse(h, w, cy, cx, ry, rx):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    return ((yy + 0.5 - cy) / ry) ** 2 + ((xx + 0.5 - cx) / rx) ** 2      # < 1 inside


def synthetic_batch(b, n_f, n_inst, h, w, seed=1234, train=True, edge=8.0, mask_scale=8, it=10000, max_inst=None,
                    motion=0.01):
    """Returns a dict of CPU float32 tensors following the reference batch contract.

    image (b,n_f,3,h,w); mask (b,n_f,n_i,h/8,w/8) {0,1}; train adds alpha, transition (b,n_f,n_i,h,w) and iter.
    `max_inst`: if given, instance slots are zero-padded to that many channels (dataset behaviour, him.py:159-174,
    with the real instances in the first slots so that no host RNG is consumed by prepare_input).
    """
    rs = np.random.RandomState(seed)
    n_slots = n_inst if max_inst is None else max_inst
    image = rs.normal(0.0, 1.0, (b, n_f, 3, h, w)).astype(np.float32)
    hm, wm = h // mask_scale, w // mask_scale
    mask = np.zeros((b, n_f, n_slots, hm, wm), np.float32)
    alpha = np.zeros((b, n_f, n_slots, h, w), np.float32)
    trans = np.zeros((b, n_f, n_slots, h, w), np.float32)
    for bi in range(b):
        for ii in range(n_inst):
            cy, cx = rs.uniform(0.3, 0.7, 2)
            ry, rx = rs.uniform(0.15, 0.3, 2)
            for fi in range(n_f):
                dy, dx = motion * fi * h, motion * fi * w
                q = _ellipse(h, w, cy * h + dy, cx * w + dx, ry * h, rx * w)
                # signed distance-like soft edge of ~`edge` pixels
                r = np.sqrt(q)
                d = (1.0 - r) * min(ry * h, rx * w)
                a = np.clip(0.5 + d / max(edge, 1e-3), 0.0, 1.0).astype(np.float32)
                alpha[bi, fi, ii] = a
                qm = _ellipse(hm, wm, (cy * h + dy) / mask_scale, (cx * w + dx) / mask_scale,
                              ry * h / mask_scale, rx * w / mask_scale)
                mask[bi, fi, ii] = (qm < 1.0).astype(np.float32)
                t = ((a > 0) & (a < 1)).astype(np.float32)
                # dilate the transition band by ~10 px with a box (cheap, deterministic)
                k = 10
                cs = np.cumsum(np.pad(t, ((k, k), (k, k))), 0)
                cs = np.cumsum(cs, 1)
                cs = np.pad(cs, ((1, 0), (1, 0)))
                win = cs[2 * k + 1:, 2 * k + 1:] - cs[:-2 * k - 1, 2 * k + 1:] - cs[2 * k + 1:, :-2 * k - 1] + cs[:-2 * k - 1, :-2 * k - 1]
                trans[bi, fi, ii] = (win > 0).astype(np.float32)
    batch = {'image': torch.from_numpy(image), 'mask': torch.from_numpy(mask)}
    if train:
        batch['alpha'] = torch.from_numpy(alpha)
        batch['transition'] = torch.from_numpy(trans)
        batch['iter'] = it
    return batch
