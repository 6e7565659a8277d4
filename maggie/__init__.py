"""`maggie` import-surface overlay for the MI355X hot path.

The reference's boundary for the matting hot path is the Python import surface `maggie.network` (maggie/network/__init__.py:5-16:
`build_model`; maggie/network/arch/__init__.py: `MaGGIe`, `MaGGIe_Temp`). This package provides exactly that surface on top of
`maggie_amd` and nothing else. `extend_path` merges it with any other `maggie` package directory found LATER on sys.path, so with

    PYTHONPATH=<this repo>:<hmchuong/MaGGIe checkout>  torchrun ... tools/main.py --config configs/maggie_image.yaml

`maggie.network` resolves here (the HIP path) while `maggie.engine`, `maggie.dataloader` and `maggie.utils` keep resolving to the
reference's own, unchanged files -- no edit of the reference tree is needed."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
