"""GPU time of every HIP C-ABI call that still runs EAGERLY (outside the trunk graphs), per entry point, per step."""
import sys, os, random, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth
from maggie_amd import hip
it = int(sys.argv[sys.argv.index('--iter') + 1]) if '--iter' in sys.argv else 10000
dev = torch.device('cuda:0')
model, _ = build_model(config.model_config('image'))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train()
batch = synth.synthetic_batch(4, 1, 2, 512, 512, seed=1234, train=True, it=it, max_inst=10)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1.5e-4 / 25, weight_decay=0.01)
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    loss['total'].backward()
    torch.nn.utils.clip_grad_norm_(params, 0.01); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
import re
names = sorted(set(re.findall(r'\bmg_[a-z0-9_]+', open(os.path.join(os.path.dirname(hip.__file__), '..', 'include', 'maggie_hip.h')).read())))
names = [n for n in names if hasattr(hip.lib(), n)]
hip.enable_timing(names)
n = 5
for _ in range(n): step()
torch.cuda.synchronize()
rec = hip.disable_timing()['records']
tot = 0.0
rows = []
for k, v in rec.items():
    if v:
        ms = sum(s.elapsed_time(e) for s, e, _, _ in v) / n
        rows.append((ms, len(v) / n, k)); tot += ms
for ms, c, k in sorted(rows, reverse=True):
    print('%-28s %6.1f calls/step %7.3f ms/step' % (k, c, ms))
print('total eager HIP-call GPU time %.2f ms/step (event-bracketed, includes inter-launch gaps inside a call)' % tot)
