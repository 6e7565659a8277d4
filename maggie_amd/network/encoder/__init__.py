from .resnet import res_shortcut_29, res_shortcut_embed_29
