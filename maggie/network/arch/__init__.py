"""`maggie.network.arch` -- `MaGGIe` / `MaGGIe_Temp` as demo/maggie_predictor.py:9,16,21 imports them (HIP-backed classes)."""
from maggie_amd.network.arch import MaGGIe, MaGGIe_Temp      # noqa: F401

__all__ = ['MaGGIe', 'MaGGIe_Temp']
