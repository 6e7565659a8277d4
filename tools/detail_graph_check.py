"""Same-state step run eagerly, then through the trunk + detail hipGraphs (first sight, capture, replays): losses, gradient norms, mattes."""
import sys, time, copy
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from helpers import *
from test_gpu_model import _build, _to
dev = torch.device('cuda:0')
from maggie_amd.utils import synth
kind = sys.argv[1] if len(sys.argv) > 1 else 'image'
n_f = 3 if kind == 'video' else 1
b = 1 if kind == 'video' else 2
model, _ = _build(kind, dev, True)
model.decoder.inst_spec_layer.dropout.p = 0.0
batch = _to(synth.synthetic_batch(b, n_f, 2, 128, 128, seed=DSEED, train=True, max_inst=10, it=10000), dev)
state = copy.deepcopy(model.state_dict())
res = []
for graphs in (False, True, True, True, True):
    model.load_state_dict(state)
    model.hip_graphs = graphs
    model.zero_grad(set_to_none=True)
    seed_all(11)
    out, loss = model(batch)
    loss['total'].backward()
    torch.cuda.synchronize()
    res.append((out, {k: float(v) for k, v in loss.items()}, {n: p.grad.float().norm().item() for n, p in model.named_parameters() if p.grad is not None}))
    print('graphs', graphs, 'total', res[-1][1]['total'], 'nparams with grad', len(res[-1][2]), 'detail graphs', {k[3:5]: type(v).__name__ for k, v in model.__dict__.get('_detail_graphs', {}).items()})
ref = res[0]
for i, r in enumerate(res[1:], 1):
    dl = max(abs(r[1][k] - ref[1][k]) / max(1, abs(ref[1][k])) for k in ref[1])
    rel = sorted(abs(r[2].get(k, 0.0) - ref[2][k]) / max(ref[2][k], 1e-12) for k in ref[2])
    da = float((r[0]['refined_masks'].float() - ref[0]['refined_masks'].float()).abs().max())
    print(i, 'max rel loss diff %.3g  grad-norm rel median %.3g p90 %.3g max %.3g  alpha max diff %.3g  missing grads %d' % (dl, rel[len(rel)//2], rel[int(.9*len(rel))], rel[-1], da, len(set(ref[2]) - set(r[2]))))
