"""Shared helpers for the parity tests (CPU oracle side)."""
import copy
import os
import random

import numpy as np
import torch

from maggie_amd.utils import synth, config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
WSEED, DSEED, RSEED = 7, 3, 11


def seed_all(s):
    np.random.seed(s)
    random.seed(s)
    torch.manual_seed(s)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def unpack_bits(arr, shape):
    n = int(np.prod(shape))
    return np.unpackbits(arr)[:n].reshape(shape)


_SD_CACHE = {}


def reference_layout_state_dict(kind='image', seed=WSEED, requires_grad=False):
    """A reference-layout state_dict (same keys/shapes as the reference checkpoints), deterministically filled.
    Keys and shapes come from the product's own modules (maggie_amd.network), which mirror the reference."""
    from maggie_amd.network import build_model
    key = (kind, seed)
    if key not in _SD_CACHE:
        cfg = config.model_config(kind)
        model, _ = build_model(cfg)
        sd = model.state_dict()
        synth.fill_state_dict_(sd, seed)
        _SD_CACHE[key] = {k: v.clone() for k, v in sd.items()}
    sd = {k: v.clone() for k, v in _SD_CACHE[key].items()}
    if requires_grad:
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(('_u', '_v', 'running_mean', 'running_var')):
                v.requires_grad_(True)
    return sd


def model_cfg(kind):
    return copy.deepcopy(config.MODEL_IMAGE if kind == 'image' else config.MODEL_VIDEO)
