"""After the graphs are captured: does every trainable parameter receive a gradient and an optimizer update? (bench.py's flow)"""
import sys, os, random; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth
from maggie_amd.optim import FlatAdamW
dev = torch.device('cuda:0')
kind = 'video' if '--video' in sys.argv else 'image'
model, _ = build_model(config.model_config(kind))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd); model.to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = FlatAdamW(params, lr=6e-6, weight_decay=0.01, max_grad_norm=0.01)
batch = synth.synthetic_batch(1 if kind == 'video' else 4, 3 if kind == 'video' else 1, 2, 512, 512, seed=1234, train=True, it=10000, max_inst=10)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
for i in range(5):
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    loss['total'].backward()
    have = sum(p.grad is not None for p in params)
    before = opt.flat_p.clone()
    opt.step()
    graphs = [type(v).__name__ for v in model._trunk_graphs.values()]
    moved = sum(bool((p.detach() != before[o:o + p.numel()].view(p.shape)).any()) for p, o in zip(params, opt._offsets))
    print('step', i, 'graphs', graphs, 'params with grad %d / %d' % (have, len(params)), 'params changed by the step', moved, 'optimizer steps', sorted(set(opt._steps)))
