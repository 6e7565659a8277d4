"""GPU idle time BETWEEN the hipGraph replays of a training step (HIP events around every replay): trunk forward | detail forward (+ losses) |
detail backward | trunk backward | optimizer. A gap is time the GPU waited for the host (the one flag read of the step, graph launches, the
autograd engine between the two backward graphs). usage: python tools/graph_gaps.py"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from maggie_amd import graphs as G
from maggie_amd.network import build_model
from maggie_amd.optim import FlatAdamW
from maggie_amd.utils import config, synth

dev = torch.device('cuda:0')
model, _ = build_model(config.model_config('image'))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train()
batch = synth.synthetic_batch(4, 1, 2, 512, 512, seed=1234, train=True, it=100, max_inst=10, edge=40.0)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
params = [p for p in model.parameters() if p.requires_grad]
opt = FlatAdamW(params, lr=1.5e-4 / 25, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=0.01)
model.grad_sink = opt.grad_views
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


orig_call, orig_bwd = G.GraphedCallable.__call__, G._Replay.backward


def call(self, *a):
    mark('fwd_begin'); r = orig_call(self, *a); mark('fwd_end'); return r


def bwd(ctx, *g):
    mark('bwd_begin'); r = orig_bwd(ctx, *g); mark('bwd_end'); return r


G.GraphedCallable.__call__ = call
G._Replay.backward = staticmethod(bwd)


def step():
    mark('step_begin')
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    loss['total'].backward()
    mark('opt_begin')
    opt.step()
    mark('step_end')


for _ in range(6):
    step()
torch.cuda.synchronize()
marks.clear()
n = 20
for _ in range(n):
    step()
torch.cuda.synchronize()
names = [m[0] for m in marks]
per = len(marks) // n
assert names[:per] == names[per:2 * per], 'steady state expected'
acc = np.zeros(per - 1)
for s in range(n):
    seg = marks[s * per:(s + 1) * per]
    for i in range(per - 1):
        acc[i] += seg[i][1].elapsed_time(seg[i + 1][1])
acc /= n
tot = 0.0
for i in range(per - 1):
    kind = 'GAP ' if not (names[i].endswith('begin') and names[i + 1].endswith('end')) else 'busy'
    print('%-11s -> %-11s %7.3f ms  %s' % (names[i], names[i + 1], acc[i], kind))
    tot += acc[i]
print('sum %.3f ms per step (+ step_end -> next step_begin)' % tot)
