"""Which source lines still run ATen ops in a training step? TorchDispatchMode over one EAGER step (no hipGraphs): every aten op that is not a
view / metadata op is attributed to the innermost maggie_amd frame on the Python stack (ops run by built-in autograd nodes have none).
usage: python tools/dispatch_ops.py [top]"""
import collections
import os
import random
import sys
import traceback

REPLAY = os.environ.get('GRAPHS', '0') == '2'          # 2: the ops issued AROUND the graph replays of a steady-state step
GRAPHS = os.environ.get('GRAPHS', '0') == '1'      # 1: count the ops recorded INTO the hipGraphs (first step = warm-up + capture; only ops issued while capturing)
if not GRAPHS and not REPLAY:
    os.environ['MAGGIE_HIP_GRAPHS'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_leaves

from maggie_amd.network import build_model
from maggie_amd.optim import FlatAdamW
from maggie_amd.utils import config, synth

dev = torch.device('cuda:0')
KIND = os.environ.get('KIND', 'image')
model, _ = build_model(config.model_config(KIND))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train()
batch = synth.synthetic_batch(1 if KIND == 'video' else 4, 3 if KIND == 'video' else 1, 2, 512, 512, seed=1234, train=True, it=100, max_inst=10, edge=40.0)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
params = [p for p in model.parameters() if p.requires_grad]
opt = FlatAdamW(params, lr=1.5e-4 / 25, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=0.01)

SKIP = ('view', 'reshape', 'expand', 'permute', 'transpose', 'unsqueeze', 'squeeze', 'slice', 'select', 'detach', 'alias', 'as_strided', 't.default',
        'unbind', 'split', 'empty', 'size', 'stride', 'is_', '_unsafe_view', 'lift_fresh', 'unfold', 'narrow', 'flatten', 'record_stream', 'chunk',
        '_local_scalar_dense', 'sym_', 'resize_', 'set_', 'item')
hits = collections.Counter()
SHAPES = os.environ.get('SHAPES', '0') == '1'


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        if not any(s in name for s in SKIP):
            # operands AND results, lists flattened (cat / stack / _foreach_* take lists; zeros / full have no tensor operand at all)
            leaves = [a for a in tree_leaves((args, kwargs or {}, out)) if torch.is_tensor(a) and a.is_cuda]
            if leaves and (not GRAPHS or torch.cuda.is_current_stream_capturing()):
                where = 'autograd engine / no maggie_amd frame'
                for fs in reversed(traceback.extract_stack(limit=40)):
                    if ('maggie_amd' in fs.filename or 'dispatch_ops' in fs.filename) and fs.name != '__torch_dispatch__':
                        where = '%s:%d %s' % (fs.filename.split('repo/')[-1], fs.lineno, fs.name)
                        break
                if SHAPES and ('dispatch_ops' in where or where.startswith('autograd engine') or 'graphs.py' in where):
                    # ops run by built-in autograd nodes (gradient accumulation of a tensor with several consumers, casts of gradients):
                    # which tensors? -> shapes and dtypes of the operands
                    where += '  ' + ' '.join('%s%s' % (str(a.dtype).replace('torch.', ''), list(a.shape)) for a in leaves[:3])
                hits[(name.replace('aten.', ''), where)] += 1
        return out


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    loss['total'].backward()
    opt.step()


for _ in range(1 if GRAPHS else 4):
    step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
with Mode():
    step()
torch.cuda.synchronize()
top = int(sys.argv[1]) if len(sys.argv) > 1 else 80
print('ATen ops touching CUDA tensors in one eager step: %d' % sum(hits.values()))
for (name, where), n in hits.most_common(top):
    print('%4d  %-34s %s' % (n, name[:34], where[:200]))
