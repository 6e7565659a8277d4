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


PREPROCESS_CASES = {'image_train': (1, 3, 64, 96, 10, 5), 'video_eval': (3, 2, 40, 56, None, 6), 'odd_size': (1, 1, 36, 52, 10, 7)}


def preprocess_inputs(key):
    """The seeded uint8 inputs tests/golden/make_golden.py:preprocess_fixture fed to the reference (same draws, same order)."""
    import numpy as np
    n_f, n_i, H, W, max_inst, seed = PREPROCESS_CASES[key]
    rs = np.random.RandomState(seed)
    frames = rs.randint(0, 256, size=(n_f, H, W, 3)).astype(np.uint8)
    alphas = rs.randint(0, 256, size=(n_f * n_i, H, W)).astype(np.uint8)
    alphas[rs.rand(*alphas.shape) < 0.3] = rs.randint(0, 8)
    masks = (rs.rand(n_f * n_i, H, W) < 0.5).astype(np.uint8) * 255
    return frames, alphas.reshape(n_f, n_i, H, W), masks.reshape(n_f, n_i, H, W), max_inst


METRIC_CASES = {'image': ((2, 3, 48, 40), 31, True), 'no_trimap': ((1, 2, 33, 57), 32, False), 'clip': ((4, 2, 32, 32), 33, True)}


def metric_inputs(key):
    """The seeded planes tests/golden/make_golden.py:metric_fixture fed to the reference's metric classes."""
    import numpy as np
    shape, seed, with_tri = METRIC_CASES[key]
    rs = np.random.RandomState(seed)
    pred = rs.rand(*shape).astype(np.float32)
    gt = np.clip(pred + rs.normal(0, 0.1, size=shape), 0, 1).astype(np.float32)
    tri = rs.randint(0, 3, size=shape).astype(np.float32) if with_tri else None
    return pred, gt, tri


# ---- observed parity distances -----------------------------------------------------------------------------------------------------------
# The step is bit-reproducible (MAGGIE_DETERMINISTIC, default): the HIP-vs-oracle distance of a fixture is ONE number, not a spread. Every parity
# test records what it measured; the bars in the tests are set at <= 2x these (VERDICT round 4, next #4a). The file is evidence, not an input:
# gpurun_out/parity_observed.json on the GPU box, copied to profiles/ per round.
_OBSERVED_PATH = os.path.join(os.path.dirname(GOLDEN), os.pardir, 'gpurun_out', 'parity_observed.json')


def record(test, **values):
    import json
    path = os.path.normpath(_OBSERVED_PATH)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = {}
        if os.path.isfile(path):
            with open(path) as f:
                data = json.load(f)
        data.setdefault(test, {}).update({k: (float(v) if not isinstance(v, (str, list, dict)) else v) for k, v in values.items()})
        with open(path, 'w') as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass
