"""Where does a training step synchronise the host with the GPU? (torch.cuda.set_sync_debug_mode('warn'))"""
import sys, os, random, warnings, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth
dev = torch.device('cuda:0')
model, _ = build_model(config.model_config('image'))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train()
batch = synth.synthetic_batch(4, 1, 2, 512, 512, seed=1234, train=True, it=100, max_inst=10, edge=40.0)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
params = [p for p in model.parameters() if p.requires_grad]
from maggie_amd.optim import FlatAdamW
opt = FlatAdamW(params, lr=1.5e-4 / 25, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=0.01)
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    loss['total'].backward()
    opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
hits = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    if 'synchroniz' in str(message):
        for fs in reversed(traceback.extract_stack(limit=25)):
            if fs.name == 'showwarning':
                continue
            if 'maggie_amd' in fs.filename or 'sync_points' in fs.filename:
                hits['%s:%d %s' % (fs.filename.split('repo/')[-1], fs.lineno, fs.name)] += 1
                break
warnings.showwarning = showwarning
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
step()
torch.cuda.set_sync_debug_mode('default')
for k, v in hits.most_common():
    print(v, k)
print('total host syncs per step:', sum(hits.values()))
