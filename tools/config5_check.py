"""BASELINE configs[4] geometry on one GPU: maggie_video.yaml, T=5, 768x768, 3 instances, one clip per GPU, bf16 train step."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth
dev = torch.device('cuda:0')
KIND = os.environ.get('KIND', 'video')
model, _ = build_model(config.model_config(KIND))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train()
model.hip_graphs = '--eager' not in sys.argv
T = int(sys.argv[sys.argv.index('--T') + 1]) if '--T' in sys.argv else 5
S = int(sys.argv[sys.argv.index('--size') + 1]) if '--size' in sys.argv else 768
batch = synth.synthetic_batch(1, T, int(os.environ.get('NI', 3)), S, S, seed=1234, train=True, it=10000, max_inst=10)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=5e-5 / 25, weight_decay=0.01)
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=os.environ.get('DT', 'bf16') == 'bf16'):
        out, loss = model(batch)
    loss['total'].backward()
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    gn = torch.nn.utils.clip_grad_norm_(params, 0.01)
    if os.environ.get('NOOPT') != '1':
        opt.step()
    print('  loss %.4f grad norm %.3g nonfinite grads %d %s' % (float(loss['total']), float(gn), len(bad), bad[:3]), flush=True)
    return out, loss
for _ in range(4): out, loss = step()
torch.cuda.synchronize(); t = time.perf_counter(); n = int(os.environ.get('NSTEPS', 8))
for _ in range(n): out, loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print('video T=%d %dx%d 3 inst:' % (T, S, S), '%.1f ms/step, %.1f instance-frames/s, loss %.4f, active ratio %.3f, peak mem %.1f GB' % (
    1e3 * dt, 3 * T / dt, float(loss['total']), float(out['detail_mask'].float().mean()) * 10 / 3, torch.cuda.max_memory_allocated() / 2**30))
print({k: tuple(v.shape) for k, v in out.items()})
