"""Slab bytes of the weight-gradient GEMMs of one training step: which layers write how many fp32 partial slabs (mg_conv_wgrad_workspace)?
usage: python tools/wgrad_slabs.py [top]   (eager step, default image workload)"""
import collections
import os
import random
import sys

os.environ['MAGGIE_HIP_GRAPHS'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from maggie_amd import kernels as K
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

rec = collections.OrderedDict()
orig = K.hip.call


def call(name, *a, **kw):
    if name in ('mg_conv_wgrad_park', 'mg_conv_wgrad_ws') and REC[0]:
        p = a[0]._obj
        need = int(a[2].value)
        key = (name[8:], p.mode, p.Cout, p.R * p.S, p.Cin, p.M, bool(p.m_dev))
        d = rec.setdefault(key, [0, 0])
        d[0] += 1; d[1] += need * 4
    return orig(name, *a, **kw)


REC = [False]
K.hip.call = call


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    loss['total'].backward()
    opt.step()


for _ in range(2):
    step()
REC[0] = True
step()
torch.cuda.synchronize()
tot = sum(v[1] for v in rec.values())
print('slab bytes per step: %.1f MB over %d launches' % (tot / 1e6, sum(v[0] for v in rec.values())))
print('%-12s %4s %5s %4s %5s %9s %4s %6s %10s' % ('entry', 'mode', 'Cout', 'taps', 'Cin', 'M', 'dev', 'calls', 'MB/call'))
for k, v in sorted(rec.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[1]) if len(sys.argv) > 1 else 40]:
    print('%-12s %4d %5d %4d %5d %9d %4d %6d %10.2f' % (k + (v[0], v[1] / v[0] / 1e6)))
