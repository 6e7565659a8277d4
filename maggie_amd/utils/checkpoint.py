"""Checkpoint format bridge: the published MaGGIe checkpoints and the reference's own training state load into
this build's modules unchanged, and what this build saves loads back into the reference.

  read_state_dict   : `.pth` (torch.save of a state_dict, maggie/engine/train.py:324,343), `.safetensors`, or a Hugging Face snapshot
                      directory (`model.safetensors` / `pytorch_model.bin`, the layout PyTorchModelHubMixin.from_pretrained reads,
                      maggie/network/__init__.py:9)
  load_state_dict   : maggie/engine/train.py:80-96 -- copy what matches, report (missing, unexpected, mismatched) instead of raising
  save_model / save_training_state / load_resume_model : `last_model.pth` + `last_opt.pth` exactly as train.py:334-343 writes and
                      train.py:98-114 reads them

Keys and shapes are the reference's (tests/golden/state_dict_layout_*.json pins them). Two tolerated differences in a source file:
a DistributedDataParallel `module.` prefix, and sparse-conv weights stored in spconv < 2.2's (kh, kw, Cin, Cout) layout instead of
(Cout, kh, kw, Cin) -- recognised by shape and permuted on load."""
import os

import torch

MODEL_FILE, OPT_FILE = 'last_model.pth', 'last_opt.pth'


def read_state_dict(path, map_location='cpu'):
    if os.path.isdir(path):
        for name in ('model.safetensors', 'pytorch_model.bin', MODEL_FILE):
            if os.path.isfile(os.path.join(path, name)):
                return read_state_dict(os.path.join(path, name), map_location)
        raise FileNotFoundError('no model.safetensors / pytorch_model.bin / %s under %s' % (MODEL_FILE, path))
    if path.endswith('.safetensors'):
        from safetensors.torch import load_file
        return load_file(path, device=str(map_location))
    sd = torch.load(path, map_location=map_location, weights_only=True)
    if isinstance(sd, dict) and 'state_dict' in sd and all(not torch.is_tensor(v) for v in sd.values() if not isinstance(v, dict)):
        sd = sd['state_dict']
    return sd


def _sparse_weight_names(model):
    from ..network.decoder.resnet_inst_matt_spconv import SparseConvWeight
    return {name + '.weight' for name, m in model.named_modules() if isinstance(m, SparseConvWeight)}


def load_state_dict(model, state_dict):
    """-> (missing_keys, unexpected_keys, mismatch_keys), like maggie/engine/train.py:80-96."""
    current = model.state_dict()
    sparse = _sparse_weight_names(model)
    missing, unexpected, mismatch = [], [], []
    seen = set()
    with torch.no_grad():
        for name, param in state_dict.items():
            if name.startswith('module.') and name not in current:
                name = name[len('module.'):]
            seen.add(name)
            if name not in current:
                unexpected.append(name)
                continue
            dst = current[name]
            if param.shape != dst.shape:
                if name in sparse and param.dim() == 4 and tuple(param.permute(3, 0, 1, 2).shape) == tuple(dst.shape):
                    param = param.permute(3, 0, 1, 2)              # (kh, kw, Cin, Cout) -> (Cout, kh, kw, Cin)
                else:
                    mismatch.append(name)
                    continue
            dst.copy_(param)
    missing = [name for name in current if name not in seen]
    return missing, unexpected, mismatch


def load_pretrained(model, path, strict=True):
    missing, unexpected, mismatch = load_state_dict(model, read_state_dict(path))
    if strict and (missing or unexpected or mismatch):
        raise RuntimeError('checkpoint %s does not match the model: missing %s, unexpected %s, shape mismatch %s'
                           % (path, missing[:5], unexpected[:5], mismatch[:5]))
    return missing, unexpected, mismatch


def save_model(model, path):
    """state_dict with the reference's keys: `.safetensors` (what the hub mixin publishes) or torch.save (`.pth`)."""
    sd = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    if path.endswith('.safetensors'):
        from safetensors.torch import save_file
        save_file(sd, path)
    else:
        torch.save(sd, path)


def save_training_state(output_dir, model, optimizer, lr_scheduler, iter, best_score):
    """train.py:334-343."""
    os.makedirs(output_dir, exist_ok=True)
    torch.save({'optimizer': optimizer.state_dict(), 'lr_scheduler': lr_scheduler.state_dict(), 'iter': iter, 'best_score': best_score},
               os.path.join(output_dir, OPT_FILE))
    save_model(model, os.path.join(output_dir, MODEL_FILE))


def load_resume_model(model, optimizer, lr_scheduler, resume_path, device):
    """train.py:98-114 -> (iter, best_score)."""
    model_path, opt_path = os.path.join(resume_path, MODEL_FILE), os.path.join(resume_path, OPT_FILE)
    if not os.path.exists(model_path) or not os.path.exists(opt_path):
        raise ValueError("Cannot resume model from {}".format(resume_path))
    load_pretrained(model, model_path, strict=True)
    opt = torch.load(opt_path, map_location=device, weights_only=False)
    optimizer.load_state_dict(opt['optimizer'])
    lr_scheduler.load_state_dict(opt['lr_scheduler'])
    return opt['iter'], opt['best_score']
