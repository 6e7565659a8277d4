"""Model registry -- mirrors maggie/network/__init__.py:5-16: build_model(cfg.model) -> (model, is_from_hf)."""
import os
import logging

from .arch import *      # noqa: F401,F403


def build_model(cfg):
    is_from_hf = False
    weights = getattr(cfg, 'weights', '') if not isinstance(cfg, dict) else cfg.get('weights', '')
    arch = getattr(cfg, 'arch', None) if not isinstance(cfg, dict) else cfg.get('arch')
    if weights != '' and not os.path.isfile(weights):
        model = eval(arch).from_pretrained(weights)
        logging.info(f"Load pretrained model {weights} from Hugging Face")
        is_from_hf = True
    else:
        model = eval(arch)(cfg)
    return model, is_from_hf
