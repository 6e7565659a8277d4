"""Child process of tests/test_gpu_graphs.py::test_syncbn_single_rank_rccl_*: one training step sequence of the image model with its BatchNorm
layers converted to nn.SyncBatchNorm inside a 1-rank RCCL process group (the all-reduce of a 1-rank group is the identity, so the result must
equal local BatchNorm up to the different statistics kernels). Runs in its own process: it owns a process group, and capturing ProcessGroupNCCL
collectives into hipGraphs used to abort with hipErrorCapturedEvent once in ~10 runs (the in-graph exchange now goes through
maggie_amd/rccl_direct.py instead) -- an abort must not take the test session down.
usage: python tests/syncbn_worker.py {local|sync_eager|sync_graphs} <port>   -> one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
mode, port = sys.argv[1], sys.argv[2]
os.environ['MAGGIE_SYNCBN_GRAPHS'] = '1' if mode == 'sync_graphs' else '0'
os.environ['MAGGIE_SYNCBN_COMM'] = 'rccl'          # this worker holds the RCCL form (a 1-rank group would also qualify for the mailbox)
os.environ['MAGGIE_SYNCBN_WORLD1'] = '1'

import numpy as np          # noqa: E402
import torch                # noqa: E402
import torch.distributed as dist   # noqa: E402

from helpers import seed_all, reference_layout_state_dict, DSEED    # noqa: E402
from maggie_amd.network import build_model                          # noqa: E402
from maggie_amd.utils import config, synth                          # noqa: E402

dev = torch.device('cuda:0')
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK='0', WORLD_SIZE='1')
dist.init_process_group('nccl', rank=0, world_size=1)
model, _ = build_model(config.model_config('image'))
model.load_state_dict(reference_layout_state_dict('image'))
model.to(dev).train()
model.decoder.inst_spec_layer.dropout.p = 0.0
if mode != 'local':
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
model.hip_graphs = mode != 'sync_eager'
batch = synth.synthetic_batch(2, 1, 2, 64, 64, seed=DSEED, train=True, max_inst=10, it=100)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
res = {'mode': mode, 'steps': []}
for i in range(4):
    seed_all(100 + i)
    model.zero_grad(set_to_none=True)
    out, loss = model(batch)
    loss['total'].backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.float() for n, p in model.named_parameters() if p.grad is not None}
    res['steps'].append({
        'loss': float(loss['total']), 'alpha_mean': float(out['refined_masks'].float().mean()), 'alpha_abs': float(out['refined_masks'].float().abs().sum()),
        'active': int(out['detail_mask'].sum()), 'n_grads': len(grads),
        'grad_norms': {n: float(g.norm()) for n, g in list(grads.items())[::7]},
        'bn_mean': float(model.encoder.bn1.running_mean.float().sum()), 'bn_var': float(model.encoder.bn1.running_var.float().sum()),
        'alpha_os8_sample': out['alpha_os8'].float().flatten()[::997][:64].cpu().tolist()})
res['graphs'] = sum(1 for st in ('_trunk_graphs', '_detail_graphs') for v in model.__dict__.get(st, {}).values() if not isinstance(v, (int, str)))
res['sync_layers'] = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in model.modules())
torch.cuda.synchronize()                                      # never destroy a graph the device may still be executing
for store in ('_trunk_graphs', '_trunk_enc_graphs', '_detail_graphs'):
    model.__dict__.get(store, {}).clear()
from maggie_amd import parallel                                     # noqa: E402
res['direct_comm_calls'] = 0 if parallel.SYNCBN_COMM is None else parallel.SYNCBN_COMM.calls
parallel.syncbn_destroy_comm()
torch.cuda.synchronize()
dist.destroy_process_group()
print('RESULT ' + json.dumps(res))
