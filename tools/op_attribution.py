"""Which source lines launch the small torch kernels? One eager training step under torch.profiler (with_stack), aten ops that
launch device kernels grouped by the innermost maggie_amd frame. usage: python tools/op_attribution.py [op-substring ...]"""
import sys, os, random, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth

dev = torch.device('cuda:0')
model, _ = build_model(config.model_config('image'))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train()
model.hip_graphs = False
batch = synth.synthetic_batch(4, 1, 2, 512, 512, seed=1234, train=True, it=100, max_inst=10)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
params = [p for p in model.parameters() if p.requires_grad]

def step():
    model.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    loss['total'].backward()

torch.autograd.set_multithreading_enabled(False)
for _ in range(2):
    step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
want = sys.argv[1:] or ['copy_', '_to_copy', 'clone', 'fill_', 'zero', 'add', 'mul', 'contiguous', 'sum', 'cat', 'stack', 'permute']
agg = collections.Counter()

class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        if any(w in name for w in want):
            frame = 'autograd/other'
            for fs in reversed(traceback.extract_stack(limit=25)):
                if 'maggie_amd' in fs.filename and not fs.filename.endswith('hip.py'):
                    frame = '%s:%d %s' % (fs.filename.split('maggie_amd/')[-1], fs.lineno, fs.name)
                    break
            agg[(name, frame)] += 1
        return func(*args, **(kwargs or {}))

with Spy():
    step()
torch.cuda.synchronize()
tot = collections.Counter()
for (name, frame), n in agg.items():
    tot[name] += n
print('dispatches by aten op:', dict(tot.most_common(25)))
for (name, frame), n in sorted(agg.items(), key=lambda kv: -kv[1])[:90]:
    print('%5d  %-28s %s' % (n, name, frame[:120]))
