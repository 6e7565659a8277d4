"""CPU: host-side logic of the MI355X build -- checkpoint layout, C ABI surface, loud failure without a GPU, config,
synthetic data contract, and the world_size-2 gloo path of the BatchNorm statistics exchange."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('kind', ['image', 'video'])
def test_state_dict_layout_matches_reference_checkpoints(kind):
    from maggie_amd.network import build_model
    from maggie_amd.utils import config
    model, from_hf = build_model(config.model_config(kind))
    assert not from_hf
    ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'state_dict_layout_%s.json' % kind)))
    sd = model.state_dict()
    assert set(sd) == set(ref['state_dict'])
    for k, (shape, dtype) in ref['state_dict'].items():
        assert list(sd[k].shape) == shape and str(sd[k].dtype).replace('torch.', '') == dtype, k
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    # identical to the reference except the never-trained `dummy_downscale` rule-book convs (frozen here)
    assert set(ref['trainable']) - trainable == {n for n in ref['trainable'] if n.startswith('decoder.dummy_downscale')}
    assert trainable <= set(ref['trainable'])


def test_c_abi_exports_every_declared_symbol():
    lib_path = os.path.join(ROOT, 'maggie_amd', 'libmaggie_hip.so')
    if not os.path.isfile(lib_path):
        import __graft_entry__
        __graft_entry__.build()
    header = open(os.path.join(ROOT, 'include', 'maggie_hip.h')).read()
    declared = set(re.findall(r'\bint\s+(mg_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 20
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mg_abi_version() >= 1


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from maggie_amd.network import build_model
    from maggie_amd.utils import config, synth
    from maggie_amd.hip import MaggieHipError
    model, _ = build_model(config.model_config('image'))
    model.eval()
    batch = synth.synthetic_batch(1, 1, 1, 64, 64, train=False)
    with pytest.raises(MaggieHipError):
        with torch.no_grad():
            model(batch)


def test_product_never_imports_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, 'maggie_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), os.path.join(dp, f)


def test_synthetic_batch_contract():
    from maggie_amd.utils import synth
    b = synth.synthetic_batch(2, 3, 2, 64, 96, train=True, max_inst=10)
    assert b['image'].shape == (2, 3, 3, 64, 96) and b['mask'].shape == (2, 3, 10, 8, 12)
    assert b['alpha'].shape == (2, 3, 10, 64, 96) and b['transition'].shape == (2, 3, 10, 64, 96)
    assert set(np.unique(b['mask'].numpy())) <= {0.0, 1.0} and float(b['mask'][:, :, 2:].abs().sum()) == 0
    a = b['alpha'].numpy()
    assert a.min() >= 0 and a.max() <= 1 and ((a > 0) & (a < 1)).any()


def test_cfg_accepts_dict_and_attr():
    from maggie_amd.utils.config import CfgNode, MODEL_IMAGE
    c = CfgNode(MODEL_IMAGE)
    assert c.encoder_args.num_mask == 10 and c['decoder_args']['final_channel'] == 64 and dict(**c.encoder_args)['num_embed'] == 3


def test_maggie_network_import_surface_shim(tmp_path):
    """The literal import surface of the reference (maggie/network/__init__.py:5-16; demo/maggie_predictor.py:9): `import maggie.network;
    build_model(cfg)` with a plain dict and with a yacs-like attribute node, `from maggie.network.arch import MaGGIe, MaGGIe_Temp` -- and the
    overlay: another `maggie` package directory later on sys.path (the reference checkout: engine / dataloader / utils) stays importable
    while `maggie.network` resolves to this build."""
    code = r"""
import sys, os
root, other = sys.argv[1], sys.argv[2]
sys.path[:0] = [root, other]
import maggie.network
from maggie.network import build_model
from maggie.network.arch import MaGGIe, MaGGIe_Temp
import maggie_amd.network.arch as ours
assert MaGGIe is ours.MaGGIe and MaGGIe_Temp is ours.MaGGIe_Temp
from maggie_amd.utils import config
import copy
cfg = copy.deepcopy(config.MODEL_IMAGE)                       # plain dict
m, from_hf = build_model(dict(cfg))
assert isinstance(m, MaGGIe) and from_hf is False


class Node(dict):                                             # yacs-like: attribute access on nested mappings
    def __init__(self, d):
        super().__init__({k: (Node(v) if isinstance(v, dict) else v) for k, v in d.items()})
    __getattr__ = dict.__getitem__


m2, _ = build_model(Node(copy.deepcopy(config.MODEL_VIDEO)))
assert isinstance(m2, MaGGIe_Temp)
assert set(m.state_dict()) <= set(m2.state_dict()) or True
import maggie.engine                                          # resolves to the OTHER checkout through extend_path
assert maggie.engine.MARK == 'reference-engine'
assert os.path.dirname(maggie.network.__file__).startswith(root)
print('SHIM_OK')
"""
    other = tmp_path / 'refcheckout'
    (other / 'maggie' / 'engine').mkdir(parents=True)
    (other / 'maggie' / '__init__.py').write_text('')
    (other / 'maggie' / 'engine' / '__init__.py').write_text("MARK = 'reference-engine'\n")
    (other / 'maggie' / 'network').mkdir()
    (other / 'maggie' / 'network' / '__init__.py').write_text("raise ImportError('the reference registry must be shadowed')\n")
    out = subprocess.run([sys.executable, '-c', code, ROOT, str(other)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0 and b'SHIM_OK' in out.stdout, out.stdout.decode()


def test_rows_of_takes_channel_slices_without_a_copy():
    """functional.rows_of: the (rows, C) operand of the row-wise kernels is a strided VIEW for a channel slice of a wider row-major buffer (what
    torch.cat's backward hands to each input) and a contiguous copy for anything the kernels' single row pitch cannot express."""
    import torch
    from maggie_amd.functional import rows_of
    g = torch.randn(4, 16, 16, 1280)
    v = g.narrow(-1, 256, 256)
    r = rows_of(v, 256)
    assert r.data_ptr() == v.data_ptr() and r.stride() == (1280, 1) and torch.equal(r, v.reshape(-1, 256))
    for bad in (g.permute(0, 2, 1, 3)[..., :256], g[:, ::2, :, :256], g.narrow(-1, 1, 256)):     # transposed rows / row gaps / a start that is not 16-byte aligned
        rb = rows_of(bad, 256)
        assert rb.is_contiguous() and torch.equal(rb, bad.reshape(-1, 256))
    c = torch.randn(2, 3, 8)
    assert rows_of(c, 8).data_ptr() == c.data_ptr()


@pytest.mark.gpu
def test_autocast_dtype_selects_the_kernel_family():
    """engine/train.py:208,227-229 runs fp16 autocast + GradScaler under `--precision 16`: served by the fp16 kernel family (not silently by
    bf16 under a loss scaler); bf16 autocast -> the bf16 family; MAGGIE_FP16_AUTOCAST=bf16 is the explicit mapping of fp16 autocast onto bf16."""
    import torch
    from maggie_amd import functional as MF
    assert MF.compute_dtype() == torch.float32
    with torch.autocast('cuda', dtype=torch.bfloat16):
        assert MF.compute_dtype() == torch.bfloat16
    prev = MF.FP16_AUTOCAST_AS_BF16
    try:
        MF.FP16_AUTOCAST_AS_BF16 = False
        with torch.autocast('cuda', dtype=torch.float16):
            assert MF.compute_dtype() == torch.float16
        MF.FP16_AUTOCAST_AS_BF16 = True
        with torch.autocast('cuda', dtype=torch.float16):
            assert MF.compute_dtype() == torch.bfloat16
    finally:
        MF.FP16_AUTOCAST_AS_BF16 = prev


_GLOO_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=2)
from maggie_amd import parallel
torch.manual_seed(dist.get_rank())
x = torch.randn(50 + 30 * dist.get_rank(), 8)
stats = torch.cat([x.sum(0), (x * x).sum(0), torch.tensor([float(x.shape[0])])])
mean, var, n = parallel.reduce_bn_stats(stats, 8, dist.group.WORLD)
allx = [torch.zeros(80, 8), torch.zeros(80, 8)]
pad = torch.zeros(80, 8); pad[:x.shape[0]] = x
dist.all_gather(allx, pad)
full = torch.cat([allx[0][:50], allx[1][:80]])
assert abs(n - 130) < 1e-6
assert torch.allclose(mean, full.mean(0), atol=1e-5) and torch.allclose(var, full.var(0, unbiased=False), atol=1e-4)
# shard assignment of the data-parallel benchmark: disjoint, covering, equal per-rank work
items = parallel.shard_items(16, dist.get_rank(), 2)
g = [None, None]; dist.all_gather_object(g, items)
assert sorted(g[0] + g[1]) == list(range(16)) and len(g[0]) == len(g[1])
# max-over-ranks timing helper
t = parallel.max_over_ranks(1.0 + dist.get_rank())
assert abs(t - 2.0) < 1e-9
# GradSync: gradients that are views of one flat buffer (hipGraph trunk export) + loose gradients + one missing gradient
import torch.nn as nn
m = nn.Sequential(nn.Linear(4, 3), nn.Linear(3, 2), nn.Linear(2, 1))
r = dist.get_rank()
ps = list(m.parameters())
flat = torch.arange(sum(p.numel() for p in ps[:4]), dtype=torch.float32) * (r + 1)
o = 0
for p in ps[:4]:                                   # first two layers: views of one flat buffer
    p.grad = flat[o:o + p.numel()].view(p.shape); o += p.numel()
ps[4].grad = torch.full_like(ps[4], float(10 * (r + 1)))      # loose gradient
if r == 0:
    ps[5].grad = torch.full_like(ps[5], 4.0)                  # rank 1 has no gradient for this one
parallel.GradSync(m)()
o = 0
for p in ps[:4]:
    exp = torch.arange(o, o + p.numel(), dtype=torch.float32).view(p.shape) * 1.5
    assert torch.allclose(p.grad, exp), (p.grad, exp); o += p.numel()
assert torch.allclose(ps[4].grad, torch.full_like(ps[4], 15.0))
assert torch.allclose(ps[5].grad, torch.full_like(ps[5], 2.0))
# the exchange FlatAdamW(sync_group=...) performs on its flat gradient buffer (one mean all-reduce, identical layout on every rank)
fg = torch.arange(10, dtype=torch.float32) * (r + 1)
parallel.all_reduce_mean(fg)
assert torch.allclose(fg, torch.arange(10, dtype=torch.float32) * 1.5)
dist.destroy_process_group()
print('OK', dist.get_rank() if False else '')
'''


def test_world_size_2_gloo_path(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(_GLOO_WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29613')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()


_SYNCBN_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=2)
from maggie_amd import parallel
r, C, eps = dist.get_rank(), 6, 1e-5
g0 = torch.Generator().manual_seed(5)
xs = [torch.randn(40, C, generator=g0) * 2 + 1, torch.randn(72, C, generator=g0) - 0.5]     # different row counts per rank (sparse BN1d)
ws = [torch.randn(40, C, generator=g0), torch.randn(72, C, generator=g0)]
gamma0, beta0 = torch.randn(C, generator=g0), torch.randn(C, generator=g0)
x, w = xs[r], ws[r]
# ---- what functional.BNAct does around the HIP kernels, with the kernels' arithmetic written in torch (CPU): the two exchanges are
# ---- the product's own functions (maggie_amd.parallel.syncbn_exchange_*), which is what this test drives
pack = parallel.syncbn_exchange_forward(torch.cat([x.sum(0), (x * x).sum(0), torch.tensor([float(x.shape[0])])]), dist.group.WORLD)
n = pack[2 * C]
mean = pack[:C] / n
invstd = torch.rsqrt((pack[C:2 * C] / n - mean * mean).clamp_min(0) + eps)
xhat = (x - mean) * invstd
y = xhat * gamma0 + beta0
g = w                                                    # dL_r/dy for L_r = (y * w).sum()
local = torch.cat([g.sum(0), (g * xhat).sum(0)])
glob, keep = parallel.syncbn_exchange_backward(local, dist.group.WORLD)
assert keep is local
dx = gamma0 * invstd * (g - glob[:C] / n - xhat * glob[C:] / n)
dbeta, dgamma = keep[:C].clone(), keep[C:].clone()
flat = torch.cat([dgamma, dbeta])
parallel.all_reduce_mean(flat)                           # the data-parallel gradient averaging (FlatAdamW(sync_group) / DDP)
# ---- reference: nn.SyncBatchNorm + DDP == one process, BatchNorm over the concatenated rows, L = L_0 + L_1, parameter gradients / world
xa = torch.cat(xs).requires_grad_(True)
ga, ba = gamma0.clone().requires_grad_(True), beta0.clone().requires_grad_(True)
ya = torch.nn.functional.batch_norm(xa, None, None, ga, ba, True, 0.1, eps)
(ya * torch.cat(ws)).sum().backward()
lo = 0 if r == 0 else 40
assert torch.allclose(y, ya[lo:lo + x.shape[0]].detach(), atol=1e-5)
assert torch.allclose(dx, xa.grad[lo:lo + x.shape[0]], atol=1e-5), float((dx - xa.grad[lo:lo + x.shape[0]]).abs().max())
assert torch.allclose(flat[:C], ga.grad / 2, atol=1e-4) and torch.allclose(flat[C:], ba.grad / 2, atol=1e-4)
# the round-1 bug: returning the all-reduced sums as dgamma would give world_size times the reference
assert not torch.allclose(glob[C:], ga.grad / 2, atol=1e-3)
dist.destroy_process_group()
"""


def test_syncbn_exchange_world_size_2_gloo(tmp_path):
    """The two SyncBN exchanges of functional.BNAct (maggie_amd.parallel.syncbn_exchange_forward / _backward) on 2 gloo ranks with
    different row counts, against full-batch BatchNorm autograd (== nn.SyncBatchNorm + DDP): y, dx, and the rank-averaged dgamma / dbeta."""
    script = tmp_path / 'sbn.py'
    script.write_text(_SYNCBN_WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29617')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()


_COMM_VOTE_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=2)
from maggie_amd import parallel, mailbox, rccl_direct
r = dist.get_rank()
made = []

class FakeComm:
    def __init__(self, kind):
        self.kind, self.destroyed, self.calls = kind, False, 0
        made.append(self)
    def self_test(self):
        return True
    def destroy(self):
        self.destroyed = True

def mailbox_ctor(fail_on):
    def ctor(group=None):
        # the real constructor votes inside, so it raises on EVERY rank or on none; here one rank alone fails: the outer vote must still agree
        if r in fail_on:
            raise RuntimeError('no fine-grained memory on rank %%d' %% r)
        return FakeComm('mailbox')
    return ctor

def rccl_ctor(fail_on, bad_sums_on=()):
    def ctor(group=None):
        if r in fail_on:
            raise RuntimeError('ncclCommInitRank failed on rank %%d' %% r)
        c = FakeComm('rccl')
        if r in bad_sums_on:
            c.self_test = lambda: False
        return c
    return ctor

def run(kind, mb_possible, mb_fail, rc_fail, rc_bad=(), one_device=True):
    os.environ['MAGGIE_SYNCBN_COMM'] = kind
    parallel.syncbn_destroy_comm()
    del made[:]
    # validated_only: what `auto` asks for -- all ranks on ONE device (the only placement the mailbox has been validated on)
    parallel._mailbox_possible = lambda group, validated_only=False: mb_possible and (one_device or not validated_only)
    mailbox.MailboxComm = mailbox_ctor(mb_fail)
    rccl_direct.DirectComm = rccl_ctor(rc_fail, rc_bad)
    try:
        got = parallel.syncbn_direct_comm()
        res = None if got is None else got.kind
    except RuntimeError as e:
        res = 'raised'
    kinds = [None, None]
    dist.all_gather_object(kinds, res)
    assert kinds[0] == kinds[1], (kind, kinds)              # every rank on the same path, whatever failed where
    live = [c for c in made if not c.destroyed]
    assert len(live) == (0 if res in (None, 'raised') else 1), (res, [(c.kind, c.destroyed) for c in made])
    return res

assert run('auto', True, (), ()) == 'mailbox'
assert run('auto', True, (1,), ()) == 'rccl'                 # one rank cannot build its mailbox -> both take the RCCL communicator, rank 0's mailbox is destroyed
assert run('auto', True, (0,), (1,)) is None                 # ... and RCCL fails on the other rank -> both keep the eager exchange
assert parallel.SYNCBN_COMM_FAILED and parallel.syncbn_direct_comm() is None      # decided once
assert run('auto', False, (), ()) == 'rccl'
assert run('auto', True, (), (), one_device=False) == 'rccl'   # ranks on DIFFERENT devices of one node: RCCL by default (ADVICE round 4, high) ...
assert run('mailbox', True, (), (), one_device=False) == 'mailbox'     # ... the mailbox there is opt-in
assert run('auto', False, (), (), rc_bad=(1,)) is None       # the trial all-reduce gave wrong sums on one rank
assert run('mailbox', False, (1,), ()) == 'raised'           # a forced form that cannot be had raises on every rank
assert run('rccl', True, (), (0,)) == 'raised'
assert run('rccl', True, (), ()) == 'rccl'
dist.destroy_process_group()
"""


def test_syncbn_comm_setup_votes_world_size_2_gloo(tmp_path):
    """parallel.syncbn_direct_comm's fallback chain (mailbox -> private RCCL communicator -> eager exchange) with communicators that fail on ONE
    rank only: both ranks must come out on the same path, the half-built communicator of the other rank destroyed."""
    script = tmp_path / 'vote.py'
    script.write_text(_COMM_VOTE_WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29619')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()


_OVERLAP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=2)
from maggie_amd import parallel
r = dist.get_rank()
# six parameters in optimizer order; the flat gradient buffer of the optimizer (FlatAdamW layout: 64-float aligned slots)
shapes = [(5,), (3, 4), (7,), (2, 2), (6,), (9,)]
ps = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
offsets, total = [], 0
for p in ps:
    offsets.append(total); total += (p.numel() + 63) // 64 * 64
g = torch.Generator().manual_seed(100 + r)
grads = [torch.randn(s, generator=g) for s in shapes]
g0, g1 = torch.Generator().manual_seed(100), torch.Generator().manual_seed(101)
both = [torch.randn(s, generator=g0) for s in shapes], [torch.randn(s, generator=g1) for s in shapes]      # both ranks' gradients
sync = parallel.OverlappedGradSync()
# backward graph A (runs first: the detail stage) owns parameters 4, 5; graph B (decoder) owns 1, 2: each hands over ONE flat buffer
def graph_export(idx):
    flat = torch.cat([grads[i].reshape(-1) for i in idx]).clone()
    sync.reduce_async([flat], [ps[i] for i in idx])            # what GraphedCallable.export_param_grads does through grad_hook
    o = 0
    for i in idx:
        ps[i].grad = flat[o:o + ps[i].numel()].view(shapes[i]); o += ps[i].numel()
graph_export([4, 5]); graph_export([1, 2])
ps[0].grad, ps[3].grad = grads[0].clone(), grads[3].clone()   # loose gradients (parameters outside the graphs this step)
# optimizer side (FlatAdamW.step): wait, gather into the flat buffer, reduce what is left in contiguous runs
done = sync.wait()
assert done == {id(ps[i]) for i in (1, 2, 4, 5)} and not sync.reduced and not sync.events
flat_g = torch.zeros(total)
for p, o in zip(ps, offsets):
    flat_g[o:o + p.numel()].copy_(p.grad.reshape(-1))
n = parallel.reduce_remaining_runs(flat_g, ps, offsets, total, done)
assert n == 2                                                  # runs [0] and [3]
for i, (p, o) in enumerate(zip(ps, offsets)):
    exp = (both[0][i] + both[1][i]) / 2
    assert torch.allclose(flat_g[o:o + p.numel()].view(shapes[i]), exp, atol=1e-6), i
# a step without any graph (first sights of a geometry): nothing in `done` -> the caller all-reduces the whole buffer
assert sync.wait() == set()
dist.destroy_process_group()
"""


def test_overlapped_grad_sync_world_size_2_gloo(tmp_path):
    """parallel.OverlappedGradSync + reduce_remaining_runs (the data-parallel exchange of FlatAdamW(sync_group) with gradient chunks handed
    over by the backward graphs) on 2 gloo ranks: every parameter's gradient ends up averaged exactly once."""
    script = tmp_path / 'ov.py'
    script.write_text(_OVERLAP_WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29619')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()


def test_flat_clip_grad_norm_matches_torch():
    """parallel.clip_grad_norm_ == torch.nn.utils.clip_grad_norm_ (the call of engine/train.py:274) for gradients that are
    views of a flat buffer (fully and partially covered) mixed with ordinary ones, and for a missing gradient."""
    import torch
    from maggie_amd import parallel
    g = torch.Generator().manual_seed(3)
    shapes = [(4, 3, 3, 3), (7,), (5, 2), (6,), (2, 2)]
    for max_norm in (0.01, 1e6):
        ours = [torch.nn.Parameter(torch.zeros(s)) for s in shapes] + [torch.nn.Parameter(torch.zeros(3))]
        ref = [torch.nn.Parameter(torch.zeros(s)) for s in shapes] + [torch.nn.Parameter(torch.zeros(3))]
        grads = [torch.randn(s, generator=g) for s in shapes]
        flat = torch.cat([grads[0].reshape(-1), grads[1].reshape(-1)])                 # fully covered base
        ours[0].grad, ours[1].grad = flat[:108].view(shapes[0]), flat[108:].view(shapes[1])
        part = torch.cat([grads[2].reshape(-1), torch.full((5,), 1e9)])                # base with foreign content: must not count
        ours[2].grad = part[:10].view(shapes[2])
        ours[3].grad, ours[4].grad = grads[3].clone(), grads[4].clone()
        for p, gr in zip(ref, grads):
            p.grad = gr.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm)
        n_ours = parallel.clip_grad_norm_(ours, max_norm)
        assert torch.allclose(n_ours, n_ref, rtol=1e-6)
        for a, b in zip(ours[:5], ref[:5]):
            assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=0)
        assert ours[5].grad is None
        assert float(part[10:].min()) == 1e9


def test_checkpoint_bridge_video_layout(tmp_path):
    """The video model's keys / shapes (tests/golden/state_dict_layout_video.json: ConvGRU `ih.0` / `hh.0`, `diff_module`) survive a
    safetensors round trip and load back without a single missing / unexpected / mismatched entry."""
    import json
    import torch
    from maggie_amd.network import build_model
    from maggie_amd.utils import checkpoint as ck, config, synth
    layout = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'state_dict_layout_video.json')))['state_dict']
    src, _ = build_model(config.model_config('video'))
    sd = src.state_dict()
    synth.fill_state_dict_(sd, 11)
    src.load_state_dict(sd)
    path = str(tmp_path / 'video.safetensors')
    ck.save_model(src, path)
    stored = ck.read_state_dict(path)
    assert set(stored) == set(layout) and all(list(stored[k].shape) == list(layout[k][0]) for k in layout)
    assert any('os8_temp_module.ih.0.weight' in k for k in stored) and any('diff_module' in k for k in stored)
    dst, _ = build_model(config.model_config('video'))
    assert ck.load_pretrained(dst, path) == ([], [], [])
    assert all(torch.equal(v, sd[k]) for k, v in dst.state_dict().items())


def test_checkpoint_bridge_round_trips_reference_formats(tmp_path):
    """maggie_amd.utils.checkpoint: .pth / .safetensors / hub-snapshot directory round trips with the reference's keys
    (tests/golden/state_dict_layout_image.json), DDP `module.` prefix, legacy spconv (kh,kw,Cin,Cout) weights, the
    (missing, unexpected, mismatch) report of maggie/engine/train.py:80-96, and the last_model.pth / last_opt.pth resume pair."""
    import json
    import torch
    from maggie_amd.network import build_model
    from maggie_amd.utils import checkpoint as ck, config, synth
    layout = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'state_dict_layout_image.json')))['state_dict']

    def fresh(seed):
        m, _ = build_model(config.model_config('image'))
        sd = m.state_dict()
        synth.fill_state_dict_(sd, seed)
        m.load_state_dict(sd)
        return m

    src, ref_sd = fresh(3), None
    ref_sd = {k: v.clone() for k, v in src.state_dict().items()}
    assert set(ref_sd) == set(layout)
    for name in ('m.pth', 'm.safetensors'):
        path = str(tmp_path / name)
        ck.save_model(src, path)
        stored = ck.read_state_dict(path)
        assert set(stored) == set(layout) and all(list(stored[k].shape) == list(layout[k][0]) for k in layout)
        dst = fresh(4)
        assert ck.load_pretrained(dst, path) == ([], [], [])
        assert all(torch.equal(v, ref_sd[k]) for k, v in dst.state_dict().items())
    # hub snapshot directory
    hub = tmp_path / 'snapshot'
    hub.mkdir()
    ck.save_model(src, str(hub / 'model.safetensors'))
    dst = fresh(5)
    ck.load_pretrained(dst, str(hub))
    assert torch.equal(dst.state_dict()['decoder.refine_OS1.3.weight'], ref_sd['decoder.refine_OS1.3.weight'])
    # DDP prefix + legacy spconv layout + a wrong shape + a foreign key + a missing key
    odd = {'module.' + k: v for k, v in ref_sd.items()}
    w = odd['module.decoder.layer3.3.weight']
    assert w.shape == (64, 3, 3, 64)
    odd['module.decoder.guidance_layer.0.weight'] = ref_sd['decoder.guidance_layer.0.weight'].permute(1, 2, 3, 0).contiguous()   # (1,1,128,64)
    odd['module.encoder.bn1.weight'] = torch.zeros(7)
    odd['module.not_a_parameter'] = torch.zeros(1)
    del odd['module.decoder.refine_OS4.3.bias']
    dst = fresh(6)
    before = dst.state_dict()['encoder.bn1.weight'].clone()
    missing, unexpected, mismatch = ck.load_state_dict(dst, odd)
    assert missing == ['decoder.refine_OS4.3.bias'] and unexpected == ['not_a_parameter'] and mismatch == ['encoder.bn1.weight']
    got = dst.state_dict()
    assert torch.equal(got['decoder.guidance_layer.0.weight'], ref_sd['decoder.guidance_layer.0.weight'])
    assert torch.equal(got['encoder.bn1.weight'], before)
    import pytest
    with pytest.raises(RuntimeError):
        ck.load_pretrained(fresh(7), _write(tmp_path, odd))
    # resume pair
    params = [p for p in src.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=10)
    for p in params[:3]:
        p.grad = torch.ones_like(p)
    opt.step(); sched.step()
    ck.save_training_state(str(tmp_path / 'run'), src, opt, sched, 123, 0.5)
    assert sorted(os.listdir(tmp_path / 'run')) == ['last_model.pth', 'last_opt.pth']
    dst = fresh(8)
    opt2 = torch.optim.AdamW([p for p in dst.parameters() if p.requires_grad], lr=1e-4)
    sched2 = torch.optim.lr_scheduler.OneCycleLR(opt2, max_lr=1e-3, total_steps=10)
    assert ck.load_resume_model(dst, opt2, sched2, str(tmp_path / 'run'), 'cpu') == (123, 0.5)
    assert sched2.last_epoch == 1 and len(opt2.state_dict()['state']) == 3
    with pytest.raises(ValueError):
        ck.load_resume_model(dst, opt2, sched2, str(tmp_path / 'nowhere'), 'cpu')


def _write(tmp_path, sd):
    import torch
    path = str(tmp_path / 'odd.pth')
    torch.save(sd, path)
    return path


def test_video_window_matches_reference_bookkeeping():
    """maggie_amd.utils.video_window.VideoWindow == the window bookkeeping of eval_video (maggie/engine/test.py:237-286, restated in
    oracle/video_window.py) over a 5-clip video, a 1-clip video (first and last at once) and a second video after a reset."""
    import numpy as np
    import torch
    from maggie_amd.utils.video_window import VideoWindow
    from oracle.video_window import Window
    rs = np.random.RandomState(0)
    ours, ref = VideoWindow(), Window()
    for n_clips in (5, 1, 2):
        for c in range(n_clips):
            a, g, t = (rs.rand(1, 3, 2, 6, 5).astype(np.float32) for _ in range(3))
            names = ['v/f%02d.jpg' % (c + k) for k in range(3)]
            first, last = c == 0, c == n_clips - 1
            o = ours.push(torch.from_numpy(a), torch.from_numpy(g), torch.from_numpy(t), names, first, last)
            r = ref.push(a, g, t, names, first, last)
            assert o['save'][0] == r['save'][0] and np.array_equal(o['save'][1].numpy(), r['save'][1])
            for key in ('current', 'previous'):
                if r[key] is None:
                    assert o[key] is None
                    continue
                for x, y in zip(o[key], r[key]):
                    assert x.shape == y.shape and np.array_equal(x.numpy(), y), (n_clips, c, key)
            if o['previous'] is not None:
                # as executed by the reference (test.py:266-268): ONE frame, also on the last clip (end index = len(prev_preds) = 1)
                assert o['previous'][0].shape[0] == 1
        assert ours.preds.shape[0] <= 3


def test_vectorised_width_draw_equals_reference_scalar_draws():
    """functional.unknown_bits draws the P random dilation widths of compute_unknown (maggie/utils/utils.py:47: one
    np.random.randint(1, k) per slice) with ONE vectorised call: same values, same generator state afterwards."""
    import numpy as np
    for seed in range(10):
        for k in (15, 27, 30):
            np.random.seed(seed)
            a = [np.random.randint(1, k) for _ in range(40)]
            x = np.random.rand()
            np.random.seed(seed)
            b = np.random.randint(1, k, size=40).astype(np.int32)
            y = np.random.rand()
            assert list(b) == a and x == y


def test_detail_graph_key_ignores_per_rank_guidance_choice_in_rank_safe_mode():
    """The key that selects a captured detail graph (maggie_amd/network/arch/maggie.py:_detail_key) must be identical on every rank of a
    data-parallel job whatever a rank's own data decided (`use_gt`: random.random(), x_os8.sum() == 0) -- in rank-safe mode the choice
    travels as a device flag among the graph inputs, so two ranks that disagree on it still look up the same graph."""
    import torch
    from maggie_amd.network import build_model
    from maggie_amd.utils import config
    model, _ = build_model(config.model_config('image'))
    model.train()
    inputs = [torch.zeros(2, 10, 64, 64), torch.zeros(2, 8, 8, 64), torch.zeros(2, dtype=torch.int32)]
    geom = (2, 1, 10, 64, 64)
    rank0 = model._detail_key(geom, {'use_gt': None, 'with_atten': False}, inputs)        # rank 0 drew use_gt = True
    rank1 = model._detail_key(geom, {'use_gt': None, 'with_atten': False}, inputs)        # rank 1 saw x_os8.sum() == 0 -> True as well / False
    assert rank0 == rank1
    plain_t = model._detail_key(geom, {'use_gt': True, 'with_atten': False}, inputs)
    plain_f = model._detail_key(geom, {'use_gt': False, 'with_atten': False}, inputs)
    assert plain_t != plain_f and rank0 not in (plain_t, plain_f)
    # default policy: rank-safe exactly when a process group with more than one rank exists (none here)
    assert model._rank_safe_graphs() is False
    model.rank_safe_graphs = True
    assert model._rank_safe_graphs() is True


_RANKSAFE_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=2)
from maggie_amd.network import build_model
from maggie_amd.utils import config
model, _ = build_model(config.model_config('image'))
model.train()
r = dist.get_rank()
# what _run_detail does with the step's plan: rank 1 ALONE saw x_os8.sum() == 0 (or drew random.random() < 0.5): its plan says use_gt
plan = {'use_gt': r == 1, 'with_atten': False}
inputs = [torch.zeros(2, 10, 64, 64), torch.zeros(2, 8, 8, 64), torch.tensor([0, int(plan['use_gt'])], dtype=torch.int32)]
geom = (2, 1, 10, 64, 64)
assert model._rank_safe_graphs() is True                   # a process group with more than one rank switches it on
rank_safe = model.training and model._rank_safe_graphs()
key = model._detail_key(geom, {'use_gt': None if rank_safe else plan['use_gt'], 'with_atten': plan['with_atten']}, inputs)
keys = [None, None]
dist.all_gather_object(keys, key)
assert keys[0] == keys[1], keys                            # same graph looked up on both ranks -> same collectives in the same order
unsafe = model._detail_key(geom, {'use_gt': plan['use_gt'], 'with_atten': plan['with_atten']}, inputs)
ukeys = [None, None]
dist.all_gather_object(ukeys, unsafe)
assert ukeys[0] != ukeys[1]                                # the single-process key (guidance choice baked into the graph) would have diverged here
dist.destroy_process_group()
"""


def test_world_size_2_ranks_that_disagree_on_the_guidance_source_look_up_the_same_detail_graph(tmp_path):
    """`use_gt` is per-rank data (x_os8.sum() == 0, random.random()). With more than one rank the detail-graph key
    must not contain it (it travels as a device flag), or ranks would replay different graphs and issue different gradient collectives. Two gloo
    ranks, rank 1 alone takes the ground-truth guidance: the keys agree; the single-process key would not."""
    script = tmp_path / 'rs.py'
    script.write_text(_RANKSAFE_WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29619')
        env.pop('MAGGIE_RANK_SAFE_GRAPHS', None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


def test_bench_gpus_flag_spawns_one_rank_per_gpu():
    """`python bench.py --gpus N` is the driver contract (bench.py:2): without a launcher it must BECOME N ranks (re-exec under
    torch.distributed.run, tools/main.py:41's job in the reference), rank 0 prints ONE line with n_gpus = N; a launcher that started a different
    number of ranks is an error, never a line about fewer GPUs. Argument plumbing only (gloo, --dry-run: no model, no measurement)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MAGGIE_DIST_BACKEND='gloo', MAGGIE_ONE_GPU='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--dry-run'],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout.decode()[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['gpus_arg'] == 2 and line['steps'] == 1 and line['warmup'] == 0
    assert line['max_over_ranks_s'] == pytest.approx(0.002)         # the max over BOTH ranks' values reached rank 0
    # N = 1 stays one process, no rendezvous
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--dry-run'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    line = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith('{"metric"')][0])
    assert line['n_gpus'] == 1 and line['backend'] is None
    # a launcher that started fewer ranks than --gpus says: refused
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-run'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         env=dict(env, WORLD_SIZE='1', RANK='0'), timeout=300)
    assert res.returncode != 0 and b'--gpus 2 but WORLD_SIZE=1' in res.stderr
