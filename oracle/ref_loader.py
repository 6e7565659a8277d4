"""ORACLE-ONLY (this container only): import the reference's own hot-path modules from /root/reference
so the CPU restatement can be pinned against them and golden vectors generated.

The reference cannot be imported as a package here (missing third-party deps: yacs, cv2, kornia, fvcore,
spconv -- SURVEY.md section 8c). This loader
  (i)  registers *empty* package objects for maggie, maggie.network{,.module,.encoder,.decoder,.arch},
       maggie.utils so that the star-import `__init__`s (which drag in baselines) never run;
  (ii) installs small stand-ins for the absent third-party modules: `yacs.config.CfgNode` (attr-dict),
       `kornia.morphology.dilation` / `fvcore.nn.weight_init` (never called on the path),
       `cv2.getStructuringElement/dilate` (oracle/region.py restatement) and `spconv.pytorch`
       (oracle/standins/spconv_standin.py);
  (iii) imports only the files on the hot path, by their real module names.

Nothing from /root/reference is copied; the reference never travels to the GPU box, and nothing under
tests -m gpu / smoke() / bench.py calls this module.
"""
import importlib
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get('MAGGIE_REFERENCE', '/root/reference')


class CfgNode(dict):
    """Minimal yacs.config.CfgNode look-alike: dict with attribute access, recursive on nested dicts."""

    def __init__(self, init_dict=None, new_allowed=False, **kw):
        super().__init__()
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _install_stubs():
    from . import region
    from .standins import spconv_standin

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if 'yacs' not in sys.modules:
        mod('yacs')
        mod('yacs.config', CfgNode=CfgNode)
    if 'cv2' not in sys.modules:
        MORPH_ELLIPSE = 2

        def getStructuringElement(shape, ksize):
            assert shape == MORPH_ELLIPSE and ksize[0] == ksize[1]
            return region.ellipse_kernel(int(ksize[0]))

        def dilate(img, kernel):
            return region.dilate(np.ascontiguousarray(img), int(kernel.shape[0]))
        mod('cv2', MORPH_ELLIPSE=MORPH_ELLIPSE, getStructuringElement=getStructuringElement, dilate=dilate)
    if 'kornia' not in sys.modules:
        mod('kornia')
        mod('kornia.morphology', dilation=None)
    if 'fvcore' not in sys.modules:
        mod('fvcore')
        mod('fvcore.nn')
        mod('fvcore.nn.weight_init')
    if 'spconv' not in sys.modules:
        sp = mod('spconv')
        sys.modules['spconv.pytorch'] = spconv_standin
        sp.pytorch = spconv_standin


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def load_reference():
    """Returns a namespace with the reference classes/functions on the hot path."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError('reference not present at %s (expected: only in the build container)' % REF_ROOT)
    sys.dont_write_bytecode = True
    _install_stubs()
    if 'maggie' in sys.modules and getattr(sys.modules['maggie'], '_oracle_loaded', False):
        return sys.modules['maggie']._oracle_ns
    base = os.path.join(REF_ROOT, 'maggie')
    top = _pkg('maggie', base)
    _pkg('maggie.utils', os.path.join(base, 'utils'))
    net = _pkg('maggie.network', os.path.join(base, 'network'))
    pmod = _pkg('maggie.network.module', os.path.join(base, 'network', 'module'))
    penc = _pkg('maggie.network.encoder', os.path.join(base, 'network', 'encoder'))
    pdec = _pkg('maggie.network.decoder', os.path.join(base, 'network', 'decoder'))
    parch = _pkg('maggie.network.arch', os.path.join(base, 'network', 'arch'))
    imp = importlib.import_module

    utils = imp('maggie.utils.utils')
    sn = imp('maggie.network.module.spectral_norm')
    basem = imp('maggie.network.module.base')
    pmod.SpectralNorm, pmod.conv1x1, pmod.conv3x3 = sn.SpectralNorm, basem.conv1x1, basem.conv3x3
    pmod.ASPP = imp('maggie.network.module.aspp').ASPP
    pmod.ConvGRU = imp('maggie.network.module.conv_gru').ConvGRU
    imp('maggie.network.module.position_encoding')
    imp('maggie.network.module.mask_attention')
    pmod.InstanceMatteDecoder = imp('maggie.network.module.instance_matte_decoder').InstanceMatteDecoder
    loss = imp('maggie.network.loss')
    enc = imp('maggie.network.encoder.resnet')
    penc.res_shortcut_29, penc.res_shortcut_embed_29 = enc.res_shortcut_29, enc.res_shortcut_embed_29
    imp('maggie.network.decoder.resnet')
    dec = imp('maggie.network.decoder.resnet_inst_matt_spconv')
    dect = imp('maggie.network.decoder.resnet_inst_matt_spconv_temp')
    pdec.res_shortcut_inst_matt_spconv_22 = dec.res_shortcut_inst_matt_spconv_22
    pdec.res_shortcut_inst_matt_spconv_temp_22 = dect.res_shortcut_inst_matt_spconv_temp_22
    arch = imp('maggie.network.arch.maggie')
    archt = imp('maggie.network.arch.maggie_temp')
    parch.MaGGIe, parch.MaGGIe_Temp = arch.MaGGIe, archt.MaGGIe_Temp

    ns = types.SimpleNamespace(
        CfgNode=CfgNode, utils=utils, loss=loss, SpectralNorm=sn.SpectralNorm, ASPP=pmod.ASPP, ConvGRU=pmod.ConvGRU,
        InstanceMatteDecoder=pmod.InstanceMatteDecoder, encoder=enc, decoder=dec, decoder_temp=dect,
        MaGGIe=arch.MaGGIe, MaGGIe_Temp=archt.MaGGIe_Temp)
    top._oracle_loaded = True
    top._oracle_ns = ns
    return ns
