#!/usr/bin/env python
"""bench.py -- MaGGIe hot-path benchmark on MI355X (driver contract: python bench.py --gpus N --steps K --warmup W).

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): maggie_image.yaml, 512x512, 2 real
instances (10 slots), batch 4 PER GPU, bf16 autocast, one full training step = forward + losses + backward +
gradient-norm clip + AdamW update (+ DDP gradient all-reduce over RCCL when N > 1; weak scaling: per-GPU work fixed).
Synthetic data / deterministic random-init weights (maggie_amd.utils.synth): no dataset or checkpoint is reachable offline.

Prints ONE JSON line on rank 0. `value` = whole-job instance-frames/s with the batch already resident in HBM.
`roofline` = the dominant kernel family (implicit-GEMM MFMA conv kernels), measured live with HIP events on extra
instrumented steps of the same workload; `cpu_baseline` = the CPU oracle (oracle/refmodel.py, a port, never the product)
timed on this host's cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak of MI355X (/opt/skills/guides/MI355X_MICROARCH.md)
DENSE_GFLOP_PER_FRAME_FWD = 69.2
SPARSE_KFLOP_PER_ACTIVE_PX_FWD = 64.5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=4, help='frames per GPU')
    ap.add_argument('--instances', type=int, default=2)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--iter', type=int, default=10000, help="batch['iter'] (past every warm-up of the reference)")
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--sync-bn', action='store_true', help='nn.SyncBatchNorm like configs/maggie_image.yaml:33 (N > 1)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-threads', type=int, default=0)
    ap.add_argument('--layers', action='store_true', help='print the per-shape conv kernel table to stderr')
    ap.add_argument('--ddp', action='store_true', help='all-reduce gradients with torch DistributedDataParallel (like the reference) instead of '
                                                      'maggie_amd.parallel.GradSync (flat-buffer RCCL all-reduce, the default for N > 1)')
    ap.add_argument('--optimizer', default='flat', choices=['flat', 'fused', 'foreach'],
                    help='flat: maggie_amd.optim.FlatAdamW (one flat buffer, clip folded into the update; default); fused / foreach: '
                         'torch.optim.AdamW implementations after maggie_amd.parallel.clip_grad_norm_')
    ap.add_argument('--video', action='store_true', help='maggie_video.yaml, T=3 (BASELINE configs[3]); not the headline line')
    return ap.parse_args()


def main():
    args = parse()
    if args.cpu_baseline_worker:
        print('CPU_BASELINE ' + json.dumps(run_cpu_baseline('video' if args.video else 'image', args)))
        return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    force_ddp = os.environ.get('MAGGIE_FORCE_DDP') == '1'          # exercise the RCCL/DDP path on a single GPU (smoke test)
    if world > 1 or force_ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from maggie_amd.network import build_model
    from maggie_amd.utils import config, synth
    from maggie_amd import hip, parallel

    kind = 'video' if args.video else 'image'
    n_f = 3 if args.video else 1
    b = 1 if args.video else args.batch
    cfg = config.model_config(kind)
    model, _ = build_model(cfg)
    sd = model.state_dict()
    synth.fill_state_dict_(sd, 1234)
    model.load_state_dict(sd)
    model.to(dev).train()
    if (world > 1 or force_ddp) and args.sync_bn:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    net = model
    grad_sync = None
    if world > 1 or force_ddp:
        if args.ddp:
            from torch.nn.parallel import DistributedDataParallel as DDP
            net = DDP(model, device_ids=[local_rank], find_unused_parameters=True, gradient_as_bucket_view=True)
        else:
            # same start on every rank (DDP broadcasts rank 0's state at construction), then flat-buffer gradient all-reduce:
            # inside FlatAdamW.step() (one collective over its flat gradient buffer) or, with a torch optimizer, parallel.GradSync
            for t_ in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t_.data, 0)
            if args.optimizer != 'flat':
                grad_sync = parallel.GradSync(model)
    params = [p for p in model.parameters() if p.requires_grad]
    # AdamW as maggie_image.yaml:90-98; lr = max_lr / 25 = the first value of the reference's OneCycleLR schedule
    # (engine/optim.py:117-118, default div_factor) -- a full max_lr step on random-init weights makes the detail region
    # (and therefore the sparse workload) drift wildly between the few timed steps.
    if args.optimizer == 'flat':                              # same update rule, one HBM pass (tests: test_flat_adamw_matches_torch_adamw)
        from maggie_amd.optim import FlatAdamW
        opt = FlatAdamW(params, lr=1.5e-4 / 25, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=0.01,           # clip: engine/train.py:274
                        sync_group=True if ((world > 1 or force_ddp) and not args.ddp) else None)
    else:
        opt = torch.optim.AdamW(params, lr=1.5e-4 / 25, betas=(0.9, 0.999), weight_decay=0.01, fused=args.optimizer == 'fused')

    batch = synth.synthetic_batch(b, n_f, args.instances, args.size, args.size, seed=1234 + rank, train=True, it=args.iter, max_inst=10)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    np.random.seed(1234 + rank)
    import random
    random.seed(1234 + rank)
    torch.manual_seed(1234 + rank)
    use_bf16 = args.dtype == 'bf16'
    stats = {}

    host_t = [] if os.environ.get('MAGGIE_HOST_TIMES') == '1' else None     # host-side (launch) time per phase, no device syncs

    def step():
        t = [time.perf_counter()]
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=use_bf16):
            out, loss = net(batch)
        t.append(time.perf_counter())
        loss['total'].backward()
        if grad_sync is not None:
            grad_sync()
        t.append(time.perf_counter())
        if args.optimizer != 'flat':
            parallel.clip_grad_norm_(params, 0.01)                                      # engine/train.py:274, over the flat grad buffers
        opt.step()
        t.append(time.perf_counter())
        if host_t is not None:
            host_t.append([1e3 * (t[i + 1] - t[i]) for i in range(3)])
        stats['active_px'] = out['detail_mask']
        stats['loss'] = loss['total']
        stats.setdefault('active_hist', []).append(out['detail_mask'].sum())           # one reduction launch; normalised when reported

    def sync():
        torch.cuda.synchronize()
        if world > 1 or force_ddp:
            dist.barrier()
            torch.cuda.synchronize()

    # One-time setup, like building the extension: the trunk's hipGraphs are captured the second time a batch geometry is seen
    # (eager step, capture step, first replay). Done before the W warm-up steps so that a small --warmup never puts the capture
    # inside the timed region.
    if model._graph_policy():
        for _ in range(3):
            step()
    for _ in range(args.warmup):
        step()
    sync()
    if os.environ.get('MAGGIE_CPROFILE'):                     # host-side profile of the launch path (backward on this thread)
        import cProfile, pstats
        torch.autograd.set_multithreading_enabled(False)
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(args.steps):
            step()
        pr.disable()
        sync()
        pstats.Stats(pr, stream=sys.stderr).sort_stats(os.environ['MAGGIE_CPROFILE']).print_stats(70)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0)

    if host_t and rank == 0:
        sys.stderr.write('host ms/step (forward+loss, backward, clip+AdamW): %s\n' % np.round(np.mean(host_t[-args.steps:], 0), 2).tolist())
    inst_frames_per_step = b * n_f * args.instances * world
    value = inst_frames_per_step * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    active_px = float(stats['active_px'].float().sum().item())
    active_ratio = active_px / (b * n_f * args.instances * args.size * args.size)
    loss_val = float(stats['loss'].item())

    roofline = None
    if not args.no_roofline:
        # Per-launch HIP-event timing needs every conv launched individually: the instrumented extra steps run the trunk
        # eagerly (the timed steps above replay it from hipGraphs, where the same kernels run back to back).
        names = ['mg_conv_fprop', 'mg_conv_fprop_ws', 'mg_conv_wgrad_ws']        # _ws: the split-K form of the same fprop family
        graphs_flag = model.__dict__.get('hip_graphs')
        model.hip_graphs = False
        hip.enable_timing(names)
        n_prof = 2
        for _ in range(n_prof):
            step()
        torch.cuda.synchronize()
        rec = hip.disable_timing()['records']
        model.__dict__['hip_graphs'] = graphs_flag
        fam = {}
        for n in names:
            for s, e, work, tag in rec[n]:
                key = '%s/%s' % ('mg_conv_fprop' if n == 'mg_conv_fprop_ws' else n, tag[0])
                d = fam.setdefault(key, [0.0, 0.0, 0])
                d[0] += s.elapsed_time(e) * 1e-3
                d[1] += work
                d[2] += 1
        if args.layers and rank == 0:
            per = {}
            for n in names:
                for s_, e_, work, tag in rec[n]:
                    k_ = (n,) + tuple(tag)
                    d_ = per.setdefault(k_, [0.0, 0.0, 0])
                    d_[0] += s_.elapsed_time(e_) * 1e-3; d_[1] += work; d_[2] += 1
            sys.stderr.write('%-14s %-5s %4s %5s %6s %8s %6s %9s %8s\n' % ('entry', 'dtype', 'mode', 'Cout', 'K', 'M', 'calls', 'us/call', 'TFLOP/s'))
            for k_, d_ in sorted(per.items(), key=lambda kv: -kv[1][0]):
                sys.stderr.write('%-14s %-5s %4d %5d %6d %8d %6d %9.1f %8.1f\n' % (k_[0], k_[1], k_[2], k_[3], k_[4], k_[5], d_[2] // n_prof, 1e6 * d_[0] / d_[2], d_[1] / d_[0] / 1e12))
        dom = max(fam.items(), key=lambda kv: kv[1][0])
        tot_t = sum(v[0] for v in fam.values())
        tot_w = sum(v[1] for v in fam.values())
        ach = dom[1][1] / dom[1][0] / 1e12
        frames_s = b * n_f * args.steps / elapsed
        scale = (args.size / 512.0) ** 2
        step_tflops = (frames_s * DENSE_GFLOP_PER_FRAME_FWD * 3 * scale * 1e9 + (active_px / ms_per_step * 1e3) * SPARSE_KFLOP_PER_ACTIVE_PX_FWD * 3e3) / 1e12
        # HBM-side traffic of the same kernel family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs of this command, corrected per MI355X_MICROARCH.md; tools/pmc_traffic.py) -- only for the config they
        # were collected on
        traffic = None
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_traffic_final_eager.json')
        if os.path.isfile(pmc_path) and not args.video and args.size == 512 and args.batch == 4 and args.iter == 10000 and use_bf16:
            fam_key = 'igemm_fprop' if 'fprop' in dom[0] else 'igemm_wgrad'
            traffic = json.load(open(pmc_path)).get(fam_key, {}).get('hbm_bytes_per_launch')
        roofline = {
            'bound': 'mfma', 'kernel': 'igemm ' + dom[0], 'achieved': round(ach, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(ach / PEAK_BF16_TFLOPS, 5), 'traffic': traffic, 'traffic_unit': 'HBM-side bytes per launch (PMC, profiles/r01_pmc_traffic_final_eager.json)',
            'launches_per_step': dom[1][2] // n_prof, 'avg_launch_us': round(1e6 * dom[1][0] / dom[1][2], 2),
            'alg_gflop_per_launch': round(dom[1][1] / dom[1][2] / 1e9, 4),
            'conv_family_ms_per_step': round(1e3 * tot_t / n_prof, 3), 'conv_family_tflops': round(tot_w / tot_t / 1e12, 2),
            'families': {k: {'ms_per_step': round(1e3 * v[0] / n_prof, 3), 'tflops': round(v[1] / v[0] / 1e12, 2), 'launches': v[2] // n_prof}
                         for k, v in fam.items()},
            'step_algorithmic_tflops_per_gpu': round(step_tflops, 2), 'step_frac_of_mfma_peak': round(step_tflops / PEAK_BF16_TFLOPS, 5),
            'hip_graphs': bool(getattr(model, '_trunk_graphs', None)) and any(not isinstance(v, (int, str)) for v in model._trunk_graphs.values()),
        }

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = cpu_baseline_subprocess(args)

    if rank == 0:
        line = {
            'metric': 'instance-frames/sec (fwd+bwd+optimizer step, %dx%d, %s)' % (args.size, args.size, args.dtype),
            'value': round(value, 3), 'unit': 'instance-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'per_gpu': round(value / world, 3),
            'config': {'workload': 'maggie_%s.yaml train step: %dx%d, %d instances (10 slots), batch %d frames/GPU x %d frame(s), iter=%d, '
                                   'fwd+loss+bwd+clip+AdamW' % (kind, args.size, args.size, args.instances, b, n_f, args.iter),
                       'global_batch': b * world, 'parallelism': 'dp%d' % world, 'sync_bn': bool(args.sync_bn and world > 1), 'optimizer': 'FlatAdamW (clip 0.01 folded in)' if args.optimizer == 'flat' else 'torch AdamW(%s) + flat-buffer grad-norm clip' % args.optimizer,
                       'grad_allreduce': None if world == 1 and not force_ddp else ('torch DDP' if args.ddp else ('one RCCL all-reduce of the flat gradient buffer inside FlatAdamW.step' if args.optimizer == 'flat' else 'GradSync (flat-buffer RCCL all-reduce)')),
                       'active_ratio': round(active_ratio, 4), 'active_ratio_per_timed_step': [round(float(v) / (b * n_f * args.instances * args.size * args.size), 3) for v in stats['active_hist'][-args.steps - 2:-2]] if not args.no_roofline else None, 'active_pixels_per_step_per_gpu': int(active_px), 'loss_total': round(loss_val, 4)},
            'roofline': roofline, 'cpu_baseline': cpu_baseline,
        }
    if world > 1 or force_ddp:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes out last: anything native libraries (RCCL's version banner) left in the C stdio buffer first
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.write(json.dumps(line) + '\n')
        sys.stdout.flush()


def cpu_baseline_subprocess(args, limit_s=150):
    """Run the CPU oracle leg in a child process with a hard time limit so the default bench always finishes in minutes."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--size', str(args.size), '--instances', str(args.instances),
           '--iter', str(args.iter), '--cpu-threads', str(args.cpu_threads)] + (['--video'] if args.video else [])
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=limit_s, env=env).stdout.decode()
        for line in out.splitlines():
            if line.startswith('CPU_BASELINE '):
                return json.loads(line[len('CPU_BASELINE '):])
        return {'value': None, 'unit': 'instance-frames/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': 'failed: ' + out[-300:]}
    except subprocess.TimeoutExpired:
        return {'value': None, 'unit': 'instance-frames/s', 'cores': os.cpu_count(), 'kind': 'port',
                'sample': 'oracle train step at %dx%d did not finish within %d s' % (args.size, args.size, limit_s)}


def run_cpu_baseline(kind, args):
    """The CPU oracle (a port of the reference path, test infrastructure) timed on this host: one bounded train step."""
    import copy
    from maggie_amd.network import build_model
    from maggie_amd.utils import config, synth
    from oracle import refmodel
    cores = os.cpu_count() or 1
    threads = args.cpu_threads if args.cpu_threads > 0 else min(cores, 32)      # beyond ~32 threads the many small ops regress
    torch.set_num_threads(threads)
    model, _ = build_model(config.model_config(kind))
    sd = model.state_dict()
    synth.fill_state_dict_(sd, 1234)
    sd = {k: v.clone() for k, v in sd.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(('_u', '_v', 'running_mean', 'running_var')):
            v.requires_grad_(True)
    n_f = 3 if kind == 'video' else 1
    b = 1 if kind == 'video' else 2
    size = min(args.size, 512)
    batch = synth.synthetic_batch(b, n_f, args.instances, size, size, seed=1234, train=True, it=args.iter, max_inst=10)
    mcfg = copy.deepcopy(config.MODEL_VIDEO if kind == 'video' else config.MODEL_IMAGE)
    np.random.seed(0)
    t0 = time.perf_counter()
    out, loss = refmodel.maggie_forward(sd, mcfg, batch, True)
    loss['total'].backward()
    dt = time.perf_counter() - t0
    ratio = float(out['detail_mask'].float().mean()) * 10.0 / args.instances
    return {'value': round(b * n_f * args.instances / dt, 4), 'unit': 'instance-frames/s', 'cores': threads, 'kind': 'port',
            'sample': 'oracle/refmodel.py fp32 train step (fwd+loss+bwd, no optimizer), %dx%d, batch %d x %d frame(s), %d instances, '
                      'active ratio %.2f, one timed step (no warm-up) = %.1f s on %d of %d host cores' % (size, size, b, n_f, args.instances,
                                                                                                       ratio, dt, threads, cores)}


if __name__ == '__main__':
    main()
