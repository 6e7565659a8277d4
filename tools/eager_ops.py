"""With the trunk replayed from hipGraphs, what does the EAGER part of a step still dispatch? (aten ops + HIP C-ABI calls)"""
import sys, os, random, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth
from maggie_amd import hip

dev = torch.device('cuda:0')
model, _ = build_model(config.model_config('image'))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train()
batch = synth.synthetic_batch(4, 1, 2, 512, 512, seed=1234, train=True, it=100, max_inst=10)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
phase = ['fwd']
agg = collections.Counter(); hipc = collections.Counter(); where = collections.Counter()
orig_call = hip.call
def spy_call(name, *a, **k):
    hipc[(phase[0], name)] += 1
    return orig_call(name, *a, **k)
hip.call = spy_call
import maggie_amd.kernels as K, maggie_amd.functional as MF
K.hip.call = spy_call

class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        agg[(phase[0], name)] += 1
        frame = 'autograd/other'
        for fs in reversed(traceback.extract_stack(limit=30)):
            if 'maggie_amd' in fs.filename and not fs.filename.endswith('hip.py') and 'graphs.py' not in fs.filename:
                frame = '%s:%d' % (fs.filename.split('maggie_amd/')[-1], fs.lineno)
                break
        where[(phase[0], frame)] += 1
        return func(*args, **(kwargs or {}))

def step(spy):
    model.zero_grad(set_to_none=True)
    phase[0] = 'fwd'
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    phase[0] = 'bwd'
    loss['total'].backward()

torch.autograd.set_multithreading_enabled(False)
for _ in range(3):
    step(False)
torch.cuda.synchronize(); agg.clear(); hipc.clear(); where.clear()
with Spy():
    step(True)
torch.cuda.synchronize()
skip = ('view', 'reshape', '_unsafe_view', 'expand', 'slice', 'select', 'detach', 'alias', 't.default', 'transpose', 'permute', 'unsqueeze', 'squeeze', 'as_strided', 'empty', 'lift_fresh', 'is_same_size', 'sym_', '_local_scalar_dense', 'unbind', 'split')
for ph in ('fwd', 'bwd'):
    ops = {k[1]: v for k, v in agg.items() if k[0] == ph and not any(k[1].startswith(s) or s in k[1] for s in skip)}
    hc = {k[1]: v for k, v in hipc.items() if k[0] == ph}
    print('== %s: %d kernel-launching aten ops, %d HIP C-ABI calls' % (ph, sum(ops.values()), sum(hc.values())))
    print('  aten:', sorted(ops.items(), key=lambda kv: -kv[1])[:40])
    print('  hip :', sorted(hc.items(), key=lambda kv: -kv[1])[:40])
    print('  lines:', sorted(((k[1], v) for k, v in where.items() if k[0] == ph), key=lambda kv: -kv[1])[:40])
