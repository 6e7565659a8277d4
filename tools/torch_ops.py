"""Which lines of maggie_amd still launch torch (ATen) kernels in a training step? One EAGER step (no hipGraphs) under torch.profiler with
stacks; every device kernel that is not one of ours is attributed to the innermost maggie_amd / bench frame of the op that launched it.
usage: python tools/torch_ops.py [top]"""
import collections
import os
import random
import sys

os.environ['MAGGIE_HIP_GRAPHS'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity

from maggie_amd.network import build_model
from maggie_amd.optim import FlatAdamW
from maggie_amd.utils import config, synth

dev = torch.device('cuda:0')
VIDEO = '--video' in sys.argv
model, _ = build_model(config.model_config('video' if VIDEO else 'image'))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train()
batch = synth.synthetic_batch(1 if VIDEO else 4, 3 if VIDEO else 1, 2, 512, 512, seed=1234, train=True, it=100, max_inst=10, edge=40.0)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
params = [p for p in model.parameters() if p.requires_grad]
opt = FlatAdamW(params, lr=1.5e-4 / 25, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=0.01)


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    loss['total'].backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

hits = collections.Counter()
time_us = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    names = [k.name for k in ev.kernels]
    if all(('at::native' not in n and 'rocclr' not in n and 'Cijk' not in n and 'multi_tensor' not in n and 'layer_norm' not in n) for n in names):
        continue
    where = 'autograd engine (no python frame)'
    node, stack = ev, list(ev.stack)
    while not stack and node.cpu_parent is not None:           # the stack is recorded on the outermost op
        node = node.cpu_parent
        stack = list(node.stack)
    for fr in stack:
        if ('maggie_amd' in fr or 'torch_ops.py' in fr) and 'graphs.py' not in fr:
            where = fr.split('repo/')[-1]
            break
    if where.startswith('autograd') and node is not ev:
        where = 'under ' + node.name[:60]
    key = '%-28s %s' % (ev.name[:28], where)
    hits[key] += len(names)
    time_us[key] += sum(k.duration for k in ev.kernels)
top = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
print('torch kernel launches in one eager step: %d, %.2f ms' % (sum(hits.values()), sum(time_us.values()) / 1e3))
for k, v in sorted(hits.items(), key=lambda kv: -time_us[kv[0]])[:top]:
    print('%4d  %8.1f us  %s' % (v, time_us[k], k))
