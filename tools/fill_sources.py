"""Which Python frames are behind aten::fill_/zero_ (FillFunctor kernels) in one eager training step incl. clip + AdamW?"""
import sys, os, random, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth

dev = torch.device('cuda:0')
model, _ = build_model(config.model_config('image'))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train(); model.hip_graphs = False
batch = synth.synthetic_batch(4, 1, 2, 512, 512, seed=1234, train=True, it=100, max_inst=10)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=6e-6)
torch.autograd.set_multithreading_enabled(False)

def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    loss['total'].backward()
    torch.nn.utils.clip_grad_norm_(params, 0.01)
    opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
names = sys.argv[1:] or ['aten::fill_', 'aten::zero_']
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if ev.name in names:
        frames = [s for s in ev.stack if 'maggie_amd' in s or 'bench' in s or 'optim' in s or 'clip_grad' in s]
        par = ev.cpu_parent
        chain = []
        while par is not None and len(chain) < 3:
            chain.append(par.name); par = par.cpu_parent
        agg[(ev.name, ' < '.join(chain), (frames[0].split('maggie_amd/')[-1] if frames else '-')[:110])] += 1
for k, v in agg.most_common(40):
    print(v, k)
