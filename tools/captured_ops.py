"""Which aten ops end up INSIDE the captured trunk graphs? (TorchDispatchMode active during the capture step; edit the op filter)"""
import sys, os, random, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth
dev = torch.device('cuda:0')
KIND = 'video' if '--video' in sys.argv else 'image'
model, _ = build_model(config.model_config(KIND))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train(); model.hip_graphs = True
batch = synth.synthetic_batch(4 if KIND == 'image' else 1, 1 if KIND == 'image' else 3, 2, 512, 512, seed=1234, train=True, it=100, max_inst=10)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
agg = collections.Counter()
ALL = '--all' in sys.argv
BIG = '--big' in sys.argv          # every op that touches >= 1M elements, with shapes
big = []
VIEWS = ('view', 'reshape', 'expand', 'permute', 'transpose', 'slice', 'select', 'unsqueeze', 'squeeze', 't.default', 'detach', 'alias', 'as_strided', 'unbind', 'split', 'empty', 'size', 'stride', 'is_', 'sym_', '_unsafe', 'lift', 'unfold', 'narrow', 'chunk', 'flatten')
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        if torch.cuda.is_current_stream_capturing() and (ALL or BIG or any(w in name for w in ('clone', 'copy_', 'contiguous', 'fill_', 'zeros', 'zero_', 'add.Tensor', 'add_.Tensor', 'sum'))) and not any(v in name for v in VIEWS):
            frame = 'autograd/other'
            for fs in reversed(traceback.extract_stack(limit=30)):
                if 'maggie_amd' in fs.filename and not fs.filename.endswith('hip.py'):
                    frame = '%s:%d' % (fs.filename.split('maggie_amd/')[-1], fs.lineno); break
            dts = tuple(str(a.dtype).replace('torch.', '') for a in args if torch.is_tensor(a))
            contig = tuple(bool(a.is_contiguous()) for a in args if torch.is_tensor(a))[:2]
            agg[(name, dts, frame, contig)] += 1
            if BIG and any(torch.is_tensor(a) and a.numel() >= (1 << 20) for a in args):
                big.append((name, [(tuple(a.shape), str(a.dtype).replace('torch.', ''), a.is_contiguous()) for a in args if torch.is_tensor(a)], frame))
        return func(*args, **(kwargs or {}))
def step():
    model.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    loss['total'].backward()
torch.autograd.set_multithreading_enabled(False)
step()
with Spy():
    step()
if BIG:
    for b_ in big:
        print(b_)
if ALL:
    by = collections.Counter()
    for (name, dts, frame, contig), v in agg.items():
        by[(name, frame.split(':')[0])] += v
    print('kernel-launching aten ops inside the captured graphs:', sum(by.values()))
    for k, v in by.most_common(45):
        print(v, k)
else:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:60]:
        print(v, k)
