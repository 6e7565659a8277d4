"""Where do two runs of the same step first differ? Forward hooks hash every module output (exact bit hash), parameter gradients are hashed after
backward; run A and run B (eager / eager, or eager / graph replay for the final outputs and gradients) are compared in call order.
usage: python tools/det_probe.py [image|video] [size] [batch] [--graphs]"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from helpers import seed_all, DSEED, reference_layout_state_dict      # noqa: E402


def bit_hash(t):
    t = t.detach().contiguous()
    if t.numel() == 0:
        return 0
    v = t.reshape(-1).view(torch.uint8).to(torch.int64)
    w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 65521
    return int((v.flatten() * w).sum().item())


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'image'
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    b = int(sys.argv[3]) if len(sys.argv) > 3 else (4 if kind == 'image' else 1)
    graphs = '--graphs' in sys.argv
    bf16 = '--bf16' in sys.argv
    from maggie_amd.network import build_model
    from maggie_amd.utils import config, synth
    dev = torch.device('cuda:0')
    model, _ = build_model(config.model_config(kind))
    model.load_state_dict(reference_layout_state_dict(kind))
    model.to(dev).train(True)
    n_f = 3 if kind == 'video' else 1
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in
             synth.synthetic_batch(b, n_f, 2, size, size, seed=DSEED, train=True, max_inst=10, it=10000).items()}
    state = copy.deepcopy(model.state_dict())
    log = []

    def hook(name):
        def fn(mod, inp, out):
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for i, o in enumerate(outs):
                if torch.is_tensor(o):
                    log.append(('%s[%d]' % (name, i), bit_hash(o)))
        return fn
    handles = [m.register_forward_hook(hook(n)) for n, m in model.named_modules() if n]

    def step(use_graphs):
        model.load_state_dict(state)
        rng = model.decoder.__dict__.get('_head_rng')
        if rng is not None:
            rng.state.copy_(torch.tensor([5, 0], dtype=torch.int64))
        model.hip_graphs = use_graphs
        model.zero_grad(set_to_none=True)
        seed_all(5)
        del log[:]
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
            out, loss = model(batch)
        loss['total'].backward()
        fwd = list(log)
        res = [('out/' + k, bit_hash(v)) for k, v in out.items() if torch.is_tensor(v)]
        res += [('loss/' + k, bit_hash(v)) for k, v in loss.items()]
        res += [('grad/' + n, bit_hash(p.grad)) for n, p in model.named_parameters() if p.grad is not None]
        res += [('state/' + n, bit_hash(v)) for n, v in model.state_dict().items() if v.is_floating_point() and not n.endswith('weight_bar')]
        return fwd, res

    runs = [step(False), step(False)]
    for h in handles:                                    # the hooks read back hashes (host syncs): not inside a capture
        h.remove()
    if graphs:
        for _ in range(3):
            runs.append(step(True))
    ref_f, ref_r = runs[0]
    for i, (f, r) in enumerate(runs[1:], 1):
        tag = 'eager' if i == 1 else 'graphs#%d' % (i - 1)
        if i == 1 or not graphs:
            nd = [(a[0], a[1] != b_[1]) for a, b_ in zip(ref_f, f)]
            first = next((n for n, d in nd if d), None)
            print('[%s vs eager0] forward module outputs: %d of %d differ; first: %s' % (tag, sum(d for _, d in nd), len(nd), first))
            if first:
                print('   differing (first 25):', [n for n, d in nd if d][:25])
        dr = [a[0] for a, b_ in zip(ref_r, r) if a[1] != b_[1]]
        print('[%s vs eager0] results: %d of %d differ' % (tag, len(dr), len(ref_r)))
        print('   out/loss:', [n for n in dr if n.startswith(('out/', 'loss/'))])
        print('   grads (first 20):', [n for n in dr if n.startswith('grad/')][:20])
        print('   state (first 20):', [n for n in dr if n.startswith('state/')][:20])


if __name__ == '__main__':
    main()
