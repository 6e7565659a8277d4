"""GPU-timeline duration of the phases of a training step (HIP events): forward graph, eager detail phase, backward graph, optimizer."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maggie_amd.network import build_model
from maggie_amd import parallel
from maggie_amd.utils import config, synth
from maggie_amd import graphs as G
it = int(sys.argv[sys.argv.index('--iter') + 1]) if '--iter' in sys.argv else 100
dev = torch.device('cuda:0')
VIDEO = '--video' in sys.argv
model, _ = build_model(config.model_config('video' if VIDEO else 'image'))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train()
batch = synth.synthetic_batch(1 if VIDEO else 4, 3 if VIDEO else 1, 2, 512, 512, seed=1234, train=True, it=it, max_inst=10)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1.5e-4 / 25, weight_decay=0.01, fused=True)       # as bench.py
ev = {}
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.setdefault(name, []).append(e)
orig_call = G.GraphedCallable.__call__
def call(self, *a):
    mark('fwd_graph_begin'); r = orig_call(self, *a); mark('fwd_graph_end'); return r
G.GraphedCallable.__call__ = call
orig_bwd = G._Replay.backward
def bwd(ctx, *g):
    mark('bwd_graph_begin'); r = orig_bwd(ctx, *g); mark('bwd_graph_end'); return r
G._Replay.backward = staticmethod(bwd)
def step():
    mark('step_begin')
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    loss['total'].backward()
    mark('backward_end')
    parallel.clip_grad_norm_(params, 0.01)
    opt.step()
    mark('step_end')
for _ in range(5): step()
torch.cuda.synchronize(); ev.clear()
n = 10
t = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print('wall ms/step %.2f' % (1e3 * (time.perf_counter() - t) / n))
def span(a, b):
    return np.mean([x.elapsed_time(y) for x, y in zip(ev[a], ev[b])])
print('step_begin -> fwd graph begin  %.2f ms' % span('step_begin', 'fwd_graph_begin'))
print('forward graph                  %.2f ms' % span('fwd_graph_begin', 'fwd_graph_end'))
print('eager detail phase (fwd+loss+bwd) %.2f ms' % span('fwd_graph_end', 'bwd_graph_begin'))
print('backward graph (+grad export)  %.2f ms' % span('bwd_graph_begin', 'bwd_graph_end'))
print('clip + AdamW                   %.2f ms' % span('backward_end', 'step_end'))
