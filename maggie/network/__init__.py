"""`maggie.network` -- the model registry the reference harness imports (engine/train.py:14,150; engine/test.py; demo/):
`build_model(cfg.model) -> (model, is_from_hf)` and the architecture classes, served by the MI355X-native implementation."""
from maggie_amd.network import build_model      # noqa: F401
from .arch import MaGGIe, MaGGIe_Temp           # noqa: F401

__all__ = ['build_model', 'MaGGIe', 'MaGGIe_Temp']
