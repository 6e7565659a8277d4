"""Forward slices/selects of tensors that require grad (each costs zeros + copy [+ add] in the backward): which lines?"""
import sys, os, random, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.overrides import TorchFunctionMode
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth

dev = torch.device('cuda:0')
kind = sys.argv[1] if len(sys.argv) > 1 else 'image'
model, _ = build_model(config.model_config(kind))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train(); model.hip_graphs = False
n_f = 1 if kind == 'image' else 3
batch = synth.synthetic_batch(4 if kind == 'image' else 2, n_f, 2, 512, 512, seed=1234, train=True, it=100, max_inst=10)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
agg = collections.Counter()

class Spy(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = getattr(func, '__name__', str(func))
        if name in ('__getitem__', 'select', 'narrow', 'chunk', 'split', 'unbind', 'index_select') and torch.is_tensor(args[0]) and args[0].requires_grad \
                and torch.is_grad_enabled():
            for fs in reversed(traceback.extract_stack(limit=25)):
                if 'maggie_amd' in fs.filename:
                    agg[('%s:%d %s' % (fs.filename.split('maggie_amd/')[-1], fs.lineno, fs.name), name, tuple(args[0].shape))] += 1
                    break
        return out

with Spy():
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
for k, v in agg.most_common(60):
    print(v, k)
