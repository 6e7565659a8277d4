"""Config objects for the MaGGIe hot path.

The reference reads a yacs ``CfgNode`` (`/root/reference/maggie/utils/config.py:55-85`, `model.*` keys) and
`MaGGIe.__init__` also accepts a plain ``dict`` (`/root/reference/maggie/network/arch/maggie.py:21-22`).
yacs is an optional dependency here: :class:`CfgNode` below is a minimal attribute-dict with the same access
pattern, and anything that behaves like a mapping with attribute access (a real yacs node included) is accepted.

``MODEL_IMAGE`` / ``MODEL_VIDEO`` restate the ``model:`` sections of `configs/maggie_image.yaml:30-65` and
`configs/maggie_video.yaml:34-69` (configuration data, needed because the yaml files do not travel to the GPU box).
"""
import copy


class CfgNode(dict):
    def __init__(self, init_dict=None, **kw):
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

    def clone(self):
        return CfgNode(copy.deepcopy(dict(self)))


def as_cfg(cfg):
    """dict -> CfgNode; leaves yacs nodes / CfgNode untouched."""
    if isinstance(cfg, dict) and not hasattr(cfg, 'encoder_args'):
        return CfgNode(cfg)
    if isinstance(cfg, dict) and not isinstance(cfg, CfgNode) and type(cfg) is dict:
        return CfgNode(cfg)
    return cfg


MODEL_IMAGE = {
    'arch': 'MaGGIe',
    'weights': '',
    'sync_bn': True,
    'having_unused_params': True,
    'warmup_iters': 3000,
    'encoder': 'res_shortcut_embed_29',
    'encoder_args': {'num_embed': 3, 'num_mask': 10, 'pretrained': True},
    'aspp': {'in_channels': 512, 'out_channels': 512},
    'decoder': 'res_shortcut_inst_matt_spconv_22',
    'decoder_args': {
        'atten_block': 2, 'atten_dim': 128, 'atten_head': 1, 'atten_stride': 1, 'detail_mask_dropout': 0.1,
        'final_channel': 64, 'freeze_detail_branch': False, 'head_channel': 120, 'max_inst': 10, 'use_id_pe': True,
        'warmup_detail_iter': 3000, 'warmup_mask_atten_iter': 0,
    },
    'loss_alpha_w': 1.0,
    'loss_alpha_type': 'l1',
    'loss_alpha_grad_w': 0.05,
    'loss_alpha_lap_w': 0.05,
    'loss_atten_w': 5.0,
    'loss_reweight_os8': True,
    'loss_dtSSD_w': 0.0,
}

MODEL_VIDEO = copy.deepcopy(MODEL_IMAGE)
MODEL_VIDEO.update({'arch': 'MaGGIe_Temp', 'warmup_iters': 500, 'decoder': 'res_shortcut_inst_matt_spconv_temp_22',
                    'loss_dtSSD_w': 1.0})
MODEL_VIDEO['decoder_args']['temp_method'] = 'bi_fusion'


def model_config(name='image'):
    return CfgNode(copy.deepcopy(MODEL_IMAGE if name == 'image' else MODEL_VIDEO))
