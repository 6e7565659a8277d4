"""Where does autograd ADD gradients? Walks the autograd graph of one training step and lists every (node, input) that receives more than one
gradient edge -- each costs (edges - 1) feature-map-sized add launches inside the backward graphs -- with the nodes the edges come from.
usage: python tools/grad_fanin.py [image|video]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, random, torch
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth
kind = sys.argv[1] if len(sys.argv) > 1 else 'image'
dev = torch.device('cuda:0')
model, _ = build_model(config.model_config(kind))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train(); model.hip_graphs = False
batch = synth.synthetic_batch(4 if kind == 'image' else 1, 3 if kind == 'video' else 1, 2, 512, 512, seed=1234, train=True, it=100, max_inst=10)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
with torch.autocast('cuda', dtype=torch.bfloat16):
    out, loss = model(batch)
root = loss['total'].grad_fn
indeg = collections.defaultdict(list)
seen, stack = set(), [root]
while stack:
    n = stack.pop()
    if n in seen:
        continue
    seen.add(n)
    for (m, idx) in n.next_functions:
        if m is None:
            continue
        indeg[(m, idx)].append(n)
        stack.append(m)
rows = []
for (m, idx), parents in indeg.items():
    if len(parents) > 1 and 'AccumulateGrad' not in m.name():
        shape = None
        try:
            shape = m._input_metadata[idx].shape if hasattr(m, '_input_metadata') else None
        except Exception:
            pass
        rows.append((len(parents) - 1, m.name(), idx, tuple(shape) if shape is not None else None, sorted(collections.Counter(p.name() for p in parents).items())))
rows.sort(key=lambda r: (-r[0], r[1]))
print('autograd nodes:', len(seen), ' accumulation adds:', sum(r[0] for r in rows))
for r in rows:
    print(r)
acc = [(len(ps) - 1, m) for (m, idx), ps in indeg.items() if len(ps) > 1 and 'AccumulateGrad' in m.name()]
print('parameters with several gradient producers:', len(acc), 'adds:', sum(a for a, _ in acc))
for a, m in acc[:40]:
    v = m.variable
    print('  ', a + 1, tuple(v.shape), v.dtype)
