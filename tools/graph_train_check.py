import sys, torch, numpy as np, random
sys.path.insert(0, 'tests')
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth
import maggie_amd.network.module.instance_matte_decoder as IMD
IMD.check_tokens = lambda t: None
dev = torch.device('cuda:0')
size, bf16, b = int(sys.argv[1]), sys.argv[2] == 'bf16', int(sys.argv[3])
variant = sys.argv[4]
model, _ = build_model(config.model_config('image'))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).train()
batch = synth.synthetic_batch(b, 1, 2, size, size, seed=1234, train=True, it=100, max_inst=10)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
np.random.seed(1); random.seed(1); torch.manual_seed(1)
orig = model._run_trunk
def spy(*a, **k):
    out = orig(*a, **k)
    print('  trunk outputs nan:', [bool(torch.isnan(t.float()).any()) if t is not None else None for t in out]); sys.stdout.flush()
    return out
model._run_trunk = spy
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1.5e-4 / 25, betas=(0.9, 0.999), weight_decay=0.01)
for i in range(6):
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
        out, loss = model(batch)
    badf = [n for n, p in model.state_dict().items() if not bool(torch.isfinite(p.float()).all())]
    print('  after forward: nonfinite state', badf[:5], len(badf))
    loss['total'].backward()
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    gn = torch.nn.utils.clip_grad_norm_(params, 0.01) if 'clip' in variant else 0.0
    if 'opt' in variant:
        opt.step()
    if 'perturb' in variant:
        with torch.no_grad():
            torch._foreach_mul_(params, 1.0001)
    badp = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
    print('  grad norm', float(gn), 'nonfinite grads', bad[:6], len(bad), 'nonfinite params', badp[:4], len(badp))
    print(i, float(loss['total'].detach()), 'graphs', [type(v).__name__ for v in model._trunk_graphs.values()]); sys.stdout.flush()
