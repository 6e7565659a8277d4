"""Host-side timeline of one training step (where does Python block / spend its time?). Run on the GPU box:
python tools/host_timeline.py [--iter 100]"""
import sys, os, time, random, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth
import maggie_amd.network.module.instance_matte_decoder as IMD
from maggie_amd import graphs as G

it = int(sys.argv[sys.argv.index('--iter') + 1]) if '--iter' in sys.argv else 100
dev = torch.device('cuda:0')
model, _ = build_model(config.model_config('image'))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train()
batch = synth.synthetic_batch(4, 1, 2, 512, 512, seed=1234, train=True, it=it, max_inst=10)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1.5e-4 / 25, weight_decay=0.01)
T = collections.defaultdict(float)

def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[label] += time.perf_counter() - t; return r
    setattr(obj, name, g)

wrap(model, 'forward_inputs', 'fwd.inputs')
wrap(model, '_run_trunk', 'fwd.trunk(total)')
wrap(IMD, 'check_tokens', 'fwd.trunk.check_tokens(sync)')
wrap(model.decoder, 'detail_stage', 'fwd.detail_stage')
wrap(model.decoder, 'predict_details', 'fwd.detail.predict_details')
wrap(model.decoder, 'fuse', 'fwd.detail.fuse')
wrap(model, 'compute_loss', 'fwd.loss')
wrap(model, '_begin_step', 'fwd.begin_step')
wrap(G.GraphedCallable, '__call__', 'fwd.trunk.graph_call')
wrap(G.GraphedCallable, 'export_param_grads', 'bwd.export_grads')
import maggie_amd.functional as MF
wrap(MF, 'unknown_bits', 'fwd.unknown_bits(all)')

def step():
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, loss = model(batch)
    t1 = time.perf_counter()
    loss['total'].backward()
    t2 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(params, 0.01)
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    T['FORWARD'] += t1 - t0; T['BACKWARD'] += t2 - t1; T['clip'] += t3 - t2; T['adamw'] += t4 - t3

for _ in range(5):
    step()
torch.cuda.synchronize(); T.clear()
n = 10
t = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print('wall ms/step %.2f' % (1e3 * (time.perf_counter() - t) / n))
for k, v in sorted(T.items()):
    print('  %-34s %7.2f ms' % (k, 1e3 * v / n))
