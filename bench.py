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
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=4, help='frames per GPU')
    ap.add_argument('--instances', type=int, default=2)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--workload', default='gt', choices=['gt', 'pred'],
                    help="gt (default): the detail region is guided by the ground-truth alphas (the reference's first warmup_detail_iter = 3000 "
                         "iterations, resnet_inst_matt_spconv.py:312-316) with a soft edge sized for the realistic active ratio of 0.15 (SURVEY 8d) -- "
                         "constant over steps and rounds; pred: iter = 10000, region from the model's own coarse alpha (random-init weights: 3-5x "
                         "the realistic ratio, drifting as AdamW trains)")
    ap.add_argument('--iter', type=int, default=None, help="batch['iter'] (default: 100 for --workload gt, 10000 for pred)")
    ap.add_argument('--edge', type=float, default=40.0, help='soft-edge width (px at 512) of the synthetic alphas: 40 -> active ratio ~0.15 with --workload gt')
    ap.add_argument('--no-trace', action='store_true', help='skip the rocprofv3 kernel trace of the graph-replayed steps (roofline.graph_replay)')
    ap.add_argument('--cpu-baseline-full', action='store_true', help='cpu_baseline with 2 warm-ups + 5 timed steps per leg (several minutes)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32', 'fp16'],
                    help="fp16 = the reference's --precision 16 recipe: fp16 autocast + GradScaler (engine/train.py:208,227-229,265-281)")
    ap.add_argument('--sync-bn', action='store_true', help='nn.SyncBatchNorm like configs/maggie_image.yaml:33 (N > 1)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-threads', type=int, default=0, help='cpu_baseline: torch threads (0 = try 32 and all host cores, report the faster)')
    ap.add_argument('--acc-file', default='', help=argparse.SUPPRESS)
    ap.add_argument('--dry-run', action='store_true', help=argparse.SUPPRESS)     # rank plumbing only (tests): rendezvous, count the ranks, print, no model
    ap.add_argument('--layers', action='store_true', help='print the per-shape conv kernel table to stderr')
    ap.add_argument('--ddp', action='store_true', help='all-reduce gradients with torch DistributedDataParallel (like the reference) instead of '
                                                      'maggie_amd.parallel.GradSync (flat-buffer RCCL all-reduce, the default for N > 1)')
    ap.add_argument('--optimizer', default='flat', choices=['flat', 'fused', 'foreach'],
                    help='flat: maggie_amd.optim.FlatAdamW (one flat buffer, clip folded into the update; default); fused / foreach: '
                         'torch.optim.AdamW implementations after maggie_amd.parallel.clip_grad_norm_')
    ap.add_argument('--video', action='store_true', help='maggie_video.yaml, T=3 (BASELINE configs[3]); not the headline line')
    ap.add_argument('--frames', type=int, default=3, help='--video: frames per clip (maggie_video.yaml:32 trains with clip_length 8)')
    ap.add_argument('--clips', type=int, default=1, help='--video: clips per GPU (maggie_video.yaml:89 trains with batch_size 4)')
    return ap.parse_args()


def main():
    args = parse()
    if args.iter is None:
        args.iter = 100 if args.workload == 'gt' else 10000
    if args.cpu_baseline_worker:
        run_cpu_baseline('video' if args.video else 'image', args)
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        spawn_ranks(args.gpus)                                 # does not return: this process becomes the launcher of N ranks
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and os.environ.get('MAGGIE_FORCE_DDP') != '1':
        # the driver contract: --gpus N IS the number of ranks. A launcher that started a different number (or none) must not produce a line
        # that says n_gpus = something else than what was asked for
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d: launch with `python bench.py --gpus N` (spawns the ranks itself) or '
                         '`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`' % (args.gpus, world))
    import torch.distributed as dist
    if args.dry_run:
        dry_run(args, world, rank)
        return
    force_ddp = os.environ.get('MAGGIE_FORCE_DDP') == '1'          # exercise the RCCL/DDP path on a single GPU (smoke test)
    if world > 1 or force_ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        # MAGGIE_DIST_BACKEND=gloo + MAGGIE_ONE_GPU=1: dry run of the multi-rank command line with every rank on cuda:0 (RCCL refuses two ranks on
        # one device; gloo moves CUDA tensors) -- the only way to execute the N > 1 path on a one-GPU box. Never a measurement.
        dist.init_process_group(os.environ.get('MAGGIE_DIST_BACKEND', 'nccl'))
    if os.environ.get('MAGGIE_ONE_GPU') == '1':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if os.environ.get('MAGGIE_MEM_FRACTION'):                 # a bound on the caching allocator: an over-sized configuration raises instead of taking the box down
        torch.cuda.set_per_process_memory_fraction(float(os.environ['MAGGIE_MEM_FRACTION']), dev)

    from maggie_amd.network import build_model
    from maggie_amd.utils import config, synth
    from maggie_amd import hip, parallel

    ranks_seen = 1
    if dist.is_initialized():
        # how many ranks the collective backend really connects (an all-reduce of ones ON the device): goes into the line next to n_gpus
        one = torch.ones(1, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        assert ranks_seen == world, 'the %s group connects %d ranks, WORLD_SIZE says %d' % (dist.get_backend(), ranks_seen, world)
    devices = sorted(set(gather_device_ids(local_rank, world)))          # collective: every rank calls it

    kind = 'video' if args.video else 'image'
    n_f = args.frames if args.video else 1
    b = args.clips if args.video else args.batch
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
        if (world > 1 or force_ddp) and not args.ddp and os.environ.get('MAGGIE_GRAD_OVERLAP', '1') != '0':
            # gradient exchange overlapped with backward on a side stream: the trunk runs as two graph pairs (encoder | decoder) and each
            # captured backward graph hands its flat gradient chunk to RCCL as soon as it has run (parallel.OverlappedGradSync)
            model.split_trunk = True
            opt.overlap = parallel.OverlappedGradSync().attach(model)
        if os.environ.get('MAGGIE_GRAD_SINK', '1') != '0' and not args.ddp:
            # the backward graphs write the gradients straight into the optimizer's flat buffer (no export / gather copies); with the overlapped
            # exchange the all-reduce then runs in place on that buffer's stretches
            model.grad_sink = opt.grad_views
    else:
        opt = torch.optim.AdamW(params, lr=1.5e-4 / 25, betas=(0.9, 0.999), weight_decay=0.01, fused=args.optimizer == 'fused')

    batch = synth.synthetic_batch(b, n_f, args.instances, args.size, args.size, seed=1234 + rank, train=True, it=args.iter, max_inst=10,
                                  edge=args.edge * args.size / 512.0)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    np.random.seed(1234 + rank)
    import random
    random.seed(1234 + rank)
    torch.manual_seed(1234 + rank)
    use_bf16 = args.dtype in ('bf16', 'fp16')                 # a 16-bit kernel family (same MFMA rate)
    amp_dtype = torch.float16 if args.dtype == 'fp16' else torch.bfloat16
    scaler = torch.amp.GradScaler('cuda') if args.dtype == 'fp16' else None
    stats = {}

    host_t = [] if os.environ.get('MAGGIE_HOST_TIMES') == '1' else None     # host-side (launch) time per phase, no device syncs

    def step():
        t = [time.perf_counter()]
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=amp_dtype, enabled=use_bf16):
            out, loss = net(batch)
        t.append(time.perf_counter())
        if scaler is not None:
            scaler.scale(loss['total']).backward()                                      # engine/train.py:265-266
        else:
            loss['total'].backward()
        if grad_sync is not None:
            grad_sync()
        t.append(time.perf_counter())
        if scaler is not None:
            scaler.unscale_(opt)                                                        # :271-272 (before the 0.01 clip)
        if args.optimizer != 'flat':
            parallel.clip_grad_norm_(params, 0.01)                                      # engine/train.py:274, over the flat grad buffers
        if scaler is not None:
            scaler.step(opt)                                                            # :277-279 (skips the update on inf / nan gradients)
            scaler.update()
        else:
            opt.step()
        t.append(time.perf_counter())
        if host_t is not None:
            host_t.append([1e3 * (t[i + 1] - t[i]) for i in range(3)])
        stats['active_px'] = out['detail_mask']
        stats['loss'] = loss['total']
        # per-step active ratio: only a reference to the step's own mask is kept here (outputs of a replayed step are private copies);
        # the reductions run after the timed region
        hist = stats.setdefault('active_masks', [])
        hist.append(out['detail_mask'])
        del hist[:-min(args.steps + 2, 66)]                       # (a mask is 10 MB: long runs keep the last 64 timed steps' only)

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
    # per-step device times of the timed region: one event per step boundary on the launch stream (no host sync inside the region)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i_ in range(args.steps):
        step()
        marks[i_ + 1].record()
    sync()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0)
    step_ms = [marks[i_].elapsed_time(marks[i_ + 1]) for i_ in range(args.steps)]

    if host_t and rank == 0:
        sys.stderr.write('host ms/step (forward+loss, backward, clip+AdamW): %s\n' % np.round(np.mean(host_t[-args.steps:], 0), 2).tolist())
    inst_frames_per_step = b * n_f * args.instances * world
    value = inst_frames_per_step * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    active_px = float(stats['active_px'].float().sum().item())
    stats['active_hist'] = [m.sum() for m in stats.pop('active_masks')[-min(args.steps, 64):]]      # the timed steps (the instrumented ones come later)
    active_ratio = active_px / (b * n_f * args.instances * args.size * args.size)
    loss_val = float(stats['loss'].item())

    roofline = None
    if not args.no_roofline:
        # (1) live HIP events: per-launch timing needs every conv launched individually, so two extra instrumented steps run eagerly (the
        #     timed steps above replay the same kernels, same launch configurations, back to back from hipGraphs);
        # (2) the graph-replayed steps themselves: a rocprofv3 --kernel-trace child of this same command (fewer steps) gives the average
        #     duration of the same kernels inside the replayed graphs -- `frac` is computed from (2) when the tracer is available.
        # FLOPs are ALGORITHMIC (kernels.py): existing taps of transposed / strided data-gradient convs, unpadded channels.
        names = ['mg_conv_fprop', 'mg_conv_fprop_ws', 'mg_conv_wgrad_ws', 'mg_conv_wgrad_park']   # fprop_ws: the split-K form of the fprop family; wgrad_park: the same GEMMs with the slab reduction batched
        fam_of = {'mg_conv_fprop_ws': 'mg_conv_fprop', 'mg_conv_wgrad_park': 'mg_conv_wgrad_ws'}
        graphs_flag = model.__dict__.get('hip_graphs')
        model.hip_graphs = False
        hip.enable_timing(names)
        n_prof = 2
        for _ in range(n_prof):
            step()
        torch.cuda.synchronize()
        rec = hip.disable_timing()['records']
        model.__dict__['hip_graphs'] = graphs_flag
        fam, sparse_t = {}, {}
        for n in names:
            for s_, e_, work, tag in rec[n]:
                key = '%s/%s' % (fam_of.get(n, n), tag[0])
                dt_ = s_.elapsed_time(e_) * 1e-3
                if work is None:                                   # sparse head (device row count): time only
                    d = sparse_t.setdefault(key, [0.0, 0]); d[0] += dt_; d[1] += 1
                    continue
                d = fam.setdefault(key, [0.0, 0.0, 0, 0.0])
                d[0] += dt_; d[1] += work; d[2] += 1; d[3] += tag[5]
        if args.layers and rank == 0:
            per = {}
            for n in names:
                for s_, e_, work, tag in rec[n]:
                    k_ = (n,) + tuple(tag[:5])
                    d_ = per.setdefault(k_, [0.0, 0.0, 0, 0.0])
                    d_[0] += s_.elapsed_time(e_) * 1e-3; d_[1] += work or 0.0; d_[2] += 1; d_[3] += tag[5]
            sys.stderr.write('%-14s %-5s %4s %5s %6s %8s %6s %9s %9s %9s\n' % ('entry', 'dtype', 'mode', 'Cout', 'K', 'M', 'calls', 'us/call', 'alg TF/s', 'exec TF/s'))
            for k_, d_ in sorted(per.items(), key=lambda kv: -kv[1][0]):
                sys.stderr.write('%-14s %-5s %4d %5d %6d %8d %6d %9.1f %9.1f %9.1f\n' % (k_[0], k_[1], k_[2], k_[3], k_[4], k_[5], d_[2] // n_prof, 1e6 * d_[0] / d_[2], d_[1] / d_[0] / 1e12, d_[3] / d_[0] / 1e12))
        dom = max(fam.items(), key=lambda kv: kv[1][0])
        tot_t = sum(v[0] for v in fam.values())
        tot_w = sum(v[1] for v in fam.values())
        ach = dom[1][1] / dom[1][0] / 1e12
        frames_s = b * n_f * args.steps / elapsed
        scale = (args.size / 512.0) ** 2
        step_tflops = (frames_s * DENSE_GFLOP_PER_FRAME_FWD * 3 * scale * 1e9 + (active_px / ms_per_step * 1e3) * SPARSE_KFLOP_PER_ACTIVE_PX_FWD * 3e3) / 1e12
        replay = None
        if rank == 0 and world == 1 and not args.no_trace:
            replay = trace_graph_replay(args, {k: (v[1] / n_prof, v[2] // n_prof) for k, v in fam.items()})
        src = 'HIP events, eager instrumented steps'
        if replay and replay.get('fprop_ms_per_step'):
            # same launches, same algorithmic FLOPs per step; durations from inside the replayed graphs
            fp_w = sum(v[1] for k, v in fam.items() if 'fprop' in k) / n_prof
            ach = fp_w / (replay['fprop_ms_per_step'] * 1e-3) / 1e12
            src = 'rocprofv3 kernel trace of the graph-replayed steps (child run of this command)'
        # HBM-side traffic of the dominant family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs,
        # corrected per MI355X_MICROARCH.md; tools/pmc_traffic.py) -- only for the configuration they were collected on
        traffic, traffic_src = None, None
        for cand in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic_final_eager.json'):
            pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', cand)
            if os.path.isfile(pmc_path) and not args.video and args.size == 512 and args.batch == 4 and use_bf16:
                pmc_doc = json.load(open(pmc_path))
                traffic = pmc_doc.get('igemm_fprop', {}).get('hbm_bytes_per_launch')
                traffic_src = 'profiles/' + cand + (', collected on commit %s' % pmc_doc['commit'] if pmc_doc.get('commit') else ', a committed file: not collected by this run')
                break
        fp = {k: v for k, v in fam.items() if 'fprop' in k}
        fp_launches = sum(v[2] for v in fp.values()) // n_prof
        roofline = {
            'bound': 'mfma', 'kernel': 'igemm_fprop (dense conv fprop / dgrad family), ' + dom[0].split('/')[1], 'achieved': round(ach, 2),
            'peak': PEAK_BF16_TFLOPS if use_bf16 else 157.3, 'unit': 'TFLOP/s',
            'frac': round(ach / (PEAK_BF16_TFLOPS if use_bf16 else 157.3), 5), 'traffic': traffic,
            'traffic_unit': 'HBM-side bytes per launch (PMC, %s)' % traffic_src, 'achieved_source': src,
            'launches_per_step': fp_launches,
            'alg_gflop_per_launch': round(sum(v[1] for v in fp.values()) / max(1, sum(v[2] for v in fp.values())) / 1e9, 4),
            'alg_gflop_per_step': round(sum(v[1] for v in fp.values()) / n_prof / 1e9, 2),
            'executed_gflop_per_step': round(sum(v[3] for v in fp.values()) / n_prof / 1e9, 2),
            'eager_events': {k: {'ms_per_step': round(1e3 * v[0] / n_prof, 3), 'alg_tflops': round(v[1] / v[0] / 1e12, 2),
                                 'executed_tflops': round(v[3] / v[0] / 1e12, 2), 'launches': v[2] // n_prof,
                                 'avg_launch_us': round(1e6 * v[0] / v[2], 2)} for k, v in fam.items()},
            'sparse_head_eager_ms_per_step': {k: round(1e3 * v[0] / n_prof, 3) for k, v in sparse_t.items()},
            'graph_replay': replay,
            'conv_family_alg_tflops_eager': round(tot_w / tot_t / 1e12, 2),
            'step_algorithmic_tflops_per_gpu': round(step_tflops, 2), 'step_frac_of_mfma_peak': round(step_tflops / PEAK_BF16_TFLOPS, 5),
            'hip_graphs': bool(getattr(model, '_trunk_graphs', None)) and any(not isinstance(v, (int, str)) for v in model._trunk_graphs.values()),
        }

    cpu_baseline = None
    accuracy = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # BASELINE.json's metric also names "alpha max-abs err vs CPU": configs[0] (maggie_image.yaml, 1 instance, 1 x 256 x 256, batch 1, fp32 eval
        # forward) through the HIP path here, through the CPU oracle in the child below, compared there (the oracle is the checker, never the product)
        acc_file = hip_accuracy_outputs(dev)
        args.acc_file = acc_file or ''
        cpu_baseline = cpu_baseline_subprocess(args)
        if acc_file and os.path.isfile(acc_file):
            os.remove(acc_file)
        if isinstance(cpu_baseline, dict):
            accuracy = cpu_baseline.pop('accuracy', None)

    if rank == 0:
        line = {
            'metric': 'instance-frames/sec (fwd+bwd+optimizer step, %dx%d, %s)' % (args.size, args.size, args.dtype),
            'value': round(value, 3), 'unit': 'instance-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3),
            'ms_per_step_spread': {'min': round(min(step_ms), 3), 'median': round(float(np.median(step_ms)), 3), 'max': round(max(step_ms), 3),
                                   'slowest_steps': sorted(((round(v, 2), i) for i, v in enumerate(step_ms)), reverse=True)[:3],
                                   'note': 'device time between per-step events of the timed region (rank 0)'},
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'per_gpu': round(value / world, 3),
            'config': {'workload': 'maggie_%s.yaml train step: %dx%d, %d instances (10 slots), batch %d frames/GPU x %d frame(s), iter=%d (%s), '
                                   'fwd+loss+bwd+clip+AdamW' % (kind, args.size, args.size, args.instances, b, n_f, args.iter,
                                                                'detail region guided by the ground-truth alphas, soft edge %g px: constant active ratio' % args.edge
                                                                if args.workload == 'gt' else 'detail region from the predicted coarse alpha: drifts with the random-init weights'),
                       'global_batch': b * world, 'parallelism': 'dp%d' % world, 'ranks_seen_by_collective_backend': ranks_seen,
                       'collective_backend': (dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else '')) if dist.is_initialized() else None,
                       'devices': devices, 'sync_bn': bool(args.sync_bn and (world > 1 or (force_ddp and os.environ.get('MAGGIE_SYNCBN_WORLD1') == '1'))), 'sync_bn_path': None if not args.sync_bn else ('eager (host-launched collectives: MAGGIE_SYNCBN_GRAPHS=0, or no in-graph exchange could be set up on this group)' if parallel.SYNCBN_COMM is None else 'hipGraphs (statistics exchange recorded into the graphs through %s; default for sync_bn: true)' % ('the mailbox all-reduce kernels (one node, peer access)' if type(parallel.SYNCBN_COMM).__name__ == 'MailboxComm' else 'a private RCCL communicator')), 'optimizer': 'FlatAdamW (clip 0.01 folded in)' if args.optimizer == 'flat' else 'torch AdamW(%s) + flat-buffer grad-norm clip' % args.optimizer,
                       'grad_allreduce': None if world == 1 and not force_ddp else ('torch DDP' if args.ddp else ('RCCL all-reduce (mean) of each of the three backward graphs\' stretches of the optimizer\'s flat gradient buffer, in place on a side stream, overlapped with the rest of backward (parallel.OverlappedGradSync + FlatAdamW gradient sink)' if (args.optimizer == 'flat' and os.environ.get('MAGGIE_GRAD_OVERLAP', '1') != '0') else 'one RCCL all-reduce of the flat gradient buffer inside FlatAdamW.step' if args.optimizer == 'flat' else 'GradSync (flat-buffer RCCL all-reduce)')),
                       'peak_hbm_gb': round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1), 'active_ratio': round(active_ratio, 4), 'active_ratio_per_timed_step': [round(float(v) / (b * n_f * args.instances * args.size * args.size), 3) for v in stats['active_hist']] if not args.no_roofline else None, 'active_pixels_per_step_per_gpu': int(active_px), 'loss_total': round(loss_val, 4)},
            'roofline': roofline, 'cpu_baseline': cpu_baseline,
            'alpha_max_abs_err': None if not accuracy else accuracy.get('alpha_max_abs_err'),
            'detail_mask_bit_exact': None if not accuracy else accuracy.get('detail_mask_bit_exact'),
            'accuracy': accuracy,
        }
    if world > 1 or force_ddp:
        # graphs that captured RCCL collectives (SyncBN statistics) must go before the communicator does
        torch.cuda.synchronize()                                      # never destroy a graph the device may still be executing
        for store in ('_trunk_graphs', '_trunk_enc_graphs', '_detail_graphs'):
            model.__dict__.get(store, {}).clear()
        parallel.syncbn_destroy_comm()
        torch.cuda.synchronize()
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


def gather_device_ids(local_rank, world):
    """device index of every rank (rank 0 prints them: N ranks must sit on N different devices for a measurement)"""
    import torch.distributed as dist
    if not dist.is_initialized() or world == 1:
        return [local_rank]
    out = [None] * world
    dist.all_gather_object(out, int(torch.cuda.current_device()))
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-exec this command line under torch.distributed.run, one rank per GPU of this node
    (the reference's launcher does the same job: tools/main.py:41 under torchrun). 127.0.0.1 rendezvous on a free port."""
    import socket
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')           # dmabuf IPC: what RCCL needs on these hosts
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def dry_run(args, world, rank):
    """Rank plumbing only (tests/test_host_cpu.py): rendezvous over the configured backend, count the ranks with an all-reduce, rank 0 prints a
    line of the same shape as the real one with `dry_run: true`. Never a measurement."""
    import torch.distributed as dist
    from maggie_amd import parallel
    backend = os.environ.get('MAGGIE_DIST_BACKEND', 'nccl')
    if world > 1:
        dist.init_process_group(backend)
    seen = torch.ones(1, dtype=torch.float64)
    if world > 1:
        if backend == 'nccl':
            seen = seen.cuda(int(os.environ.get('LOCAL_RANK', '0')))
        dist.all_reduce(seen)
        dist.barrier()
    t = parallel.max_over_ranks(0.001 * (rank + 1))
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.write(json.dumps({'metric': 'dry run (rank plumbing only)', 'dry_run': True, 'value': None, 'n_gpus': world, 'gpus_arg': args.gpus,
                                     'ranks_seen': int(seen.item()), 'backend': backend if world > 1 else None, 'steps': args.steps, 'warmup': args.warmup,
                                     'max_over_ranks_s': t}) + '\n')
        sys.stdout.flush()


def trace_graph_replay(args, fam_alg):
    """rocprofv3 --kernel-trace over a short child run of THIS command: average durations of the conv kernels inside the replayed hipGraphs
    (HIP events cannot bracket a kernel inside a graph). -> dict or None when the tracer is unavailable / failed."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.isfile('/opt/rocm/bin/rocprofv3') else None)
    if prof is None:
        return None
    n_steps = 6
    out_dir = tempfile.mkdtemp(prefix='mg_trace_', dir='/tmp')
    cmd = [prof, '--kernel-trace', '--output-format', 'csv', '-d', out_dir, '--', sys.executable, os.path.abspath(__file__), '--steps', str(n_steps),
           '--warmup', '2', '--no-roofline', '--no-cpu-baseline', '--batch', str(args.batch), '--instances', str(args.instances), '--size', str(args.size),
           '--workload', args.workload, '--iter', str(args.iter), '--edge', str(args.edge), '--dtype', args.dtype] + (['--video', '--frames', str(args.frames), '--clips', str(args.clips)] if args.video else [])
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'))
        line = [l for l in res.stdout.decode(errors='replace').splitlines() if l.startswith('{"metric"')]
        files = glob.glob(os.path.join(out_dir, '**', '*kernel_trace.csv'), recursive=True)
        if not line or not files:
            return None
        ms = json.loads(line[-1])['ms_per_step']
        rows = []
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
        rows.sort()
        # the last (n_steps - 1) steps of the timed region: whole steps, all replayed from graphs
        k = n_steps - 1
        t_end = rows[-1][1]
        # Whole steps by a marker, not by the clock: the optimizer kernel runs exactly once per step, so the launches between the ends of marker
        # (k + 1)-from-last and the last marker are k steps whatever the host did meanwhile. (The window used to be "the last k x ms_per_step of the trace":
        # one host stall inside it -- the traced child of the round's final run: 15.6 ms per step, 63 % busy -- and it held 4.7 steps' launches counted as 5:
        # frac 0.124 instead of 0.112.) Workloads without an optimizer step (--workload pred) keep the clock window.
        marks = [e_ for _, e_, nm in rows if 'adamw_kernel' in nm]
        window_by = 'clock'
        if len(marks) >= k + 1:
            win = [r for r in rows if r[0] >= marks[-(k + 1)] and r[1] <= marks[-1]]
            window_by = 'optimizer kernel'
        else:
            win = [r for r in rows if r[0] >= t_end - k * ms * 1e6]
        acc = {}
        for s_, e_, nm in win:
            # dense fprop family = every igemm_fprop_* kernel that is not the persistent (sparse-head) form: the register-staged im2col loop,
            # the direct-to-LDS im2col ring, the spatial halo-tile kernel, split-K + its finishing kernel
            # (conv_halo3_kernel: the round-6 producer / consumer halo-tile form of the same family, csrc/conv_halo3.hip)
            fam = 'sparse' if ('igemm_fprop' in nm and 'persistent' in nm) else 'fprop' if ('igemm_fprop' in nm or 'splitk_finish' in nm or 'conv_halo3' in nm) else \
                'wgrad' if ('igemm_wgrad' in nm or 'wgrad_reduce' in nm) else None
            if fam:
                d = acc.setdefault(fam, [0, 0, 0])
                d[0] += e_ - s_
                d[1] += 1
                d[2] += int('igemm_fprop' in nm or 'igemm_wgrad' in nm or 'conv_halo3' in nm)
        busy = sum(e_ - s_ for s_, e_, _ in win)
        span = win[-1][1] - win[0][0]
        out = {'steps_in_window': k, 'window_by': window_by, 'traced_ms_per_step': ms, 'launches_per_step': round(len(win) / k, 1), 'gpu_busy_frac': round(busy / span, 4)}
        for fam, (t, c, cm) in acc.items():
            out[fam + '_ms_per_step'] = round(t / 1e6 / k, 4)
            out[fam + '_launches_per_step'] = round(c / k, 1)
            out[fam + '_avg_kernel_us'] = round(t / 1e3 / max(cm, 1), 2)          # per igemm launch (split-K finish / reduce time included)
        return out
    except Exception as e:                                          # the trace is evidence, never a reason to fail the bench
        return {'error': '%s: %s' % (type(e).__name__, e)}
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


ACC_CFG = dict(b=1, n_f=1, n_inst=1, size=256, seed=1234)         # BASELINE.json configs[0]


def hip_accuracy_outputs(dev):
    """fp32 eval forward of BASELINE configs[0] through the HIP path (fresh model, the bench's deterministic synthetic weights) -> path of an .npz
    with its outputs; the cpu_baseline child runs the CPU oracle on the same inputs / weights and compares."""
    import tempfile
    from maggie_amd.network import build_model
    from maggie_amd.utils import config, synth
    try:
        model, _ = build_model(config.model_config('image'))
        sd = model.state_dict()
        synth.fill_state_dict_(sd, ACC_CFG['seed'])
        model.load_state_dict(sd)
        model.to(dev).eval()
        c = ACC_CFG
        batch = synth.synthetic_batch(c['b'], c['n_f'], c['n_inst'], c['size'], c['size'], seed=c['seed'], train=False)
        with torch.no_grad():
            out = model({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()})
        fd, path = tempfile.mkstemp(prefix='mg_acc_', suffix='.npz', dir='/tmp')
        os.close(fd)
        np.savez(path, **{k: out[k].float().cpu().numpy() for k in ('alpha_os8', 'alpha_os4', 'alpha_os1', 'refined_masks', 'detail_mask')})
        return path
    except Exception as e:                                          # evidence, never a reason to fail the bench
        sys.stderr.write('accuracy leg (HIP side) failed: %s: %s\n' % (type(e).__name__, e))
        return None


def cpu_accuracy(acc_file):
    """CPU oracle on BASELINE configs[0] against the HIP outputs in `acc_file` -> dict(alpha_max_abs_err, detail_mask_bit_exact, ...)."""
    import copy
    from maggie_amd.network import build_model
    from maggie_amd.utils import config, synth
    from oracle import refmodel
    c = ACC_CFG
    hipo = np.load(acc_file)
    model, _ = build_model(config.model_config('image'))
    sd = model.state_dict()
    synth.fill_state_dict_(sd, c['seed'])
    batch = synth.synthetic_batch(c['b'], c['n_f'], c['n_inst'], c['size'], c['size'], seed=c['seed'], train=False)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = refmodel.maggie_forward({k: v.clone() for k, v in sd.items()}, copy.deepcopy(config.MODEL_IMAGE), batch, False)
    dt = time.perf_counter() - t0
    errs = {k: float(np.abs(hipo[k] - ref[k].float().numpy()).max()) for k in ('alpha_os8', 'alpha_os4', 'alpha_os1', 'refined_masks')}
    dm_ref = ref['detail_mask'].numpy()
    return {'config': 'BASELINE configs[0]: maggie_image.yaml, 1 instance, 1x256x256, batch 1, fp32 eval forward; HIP path vs CPU oracle (oracle/refmodel.py), same synthetic weights and inputs',
            'alpha_max_abs_err': errs['refined_masks'], 'max_abs_err_per_output': {k: float('%.3g' % v) for k, v in errs.items()},
            'detail_mask_bit_exact': bool(np.array_equal(hipo['detail_mask'] != 0, dm_ref != 0)),
            'detail_pixels': int((dm_ref != 0).sum()), 'tolerance': 1e-3, 'cpu_forward_s': round(dt, 3)}


def cpu_baseline_subprocess(args):
    """Run the CPU oracle legs in a child process with a hard time limit so the default bench always finishes in minutes."""
    import subprocess
    limit_s = 900 if args.cpu_baseline_full else 480
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--size', str(args.size), '--instances', str(args.instances),
           '--batch', str(args.batch), '--iter', str(args.iter), '--edge', str(args.edge), '--workload', args.workload,
           '--cpu-threads', str(args.cpu_threads)] + (['--video', '--frames', str(args.frames), '--clips', str(args.clips)] if args.video else []) + (['--cpu-baseline-full'] if args.cpu_baseline_full else []) + \
          (['--acc-file', args.acc_file] if getattr(args, 'acc_file', '') else [])
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    part = None
    timed_out = False
    try:
        pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
        out, _ = pr.communicate(timeout=limit_s)
        out = out.decode(errors='replace')
    except subprocess.TimeoutExpired:
        pr.kill()
        out = pr.communicate()[0].decode(errors='replace')
        timed_out = True
    for line in out.splitlines():                                   # the worker prints a (growing) result line after every timed step
        if line.startswith('CPU_BASELINE '):
            part = json.loads(line[len('CPU_BASELINE '):])
    if part is not None:
        if timed_out and 'per_thread_setting' in part and args.cpu_threads <= 0:
            # a thread setting the worker had started but not finished a single step of when the budget ran out (the all-cores setting on a
            # 256-core host: the model's many small ops take minutes per step there) is reported as such, not dropped
            cores = os.cpu_count() or 1
            for t in [min(cores, 32)] + ([cores] if cores > 32 else []):
                part['per_thread_setting'].setdefault(str(t), {'value': None, 'note': 'not one step finished within the %d s budget of the CPU legs' % limit_s})
        return part
    return {'value': None, 'unit': 'instance-frames/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': 'failed: ' + out[-300:]}


def run_cpu_baseline(kind, args):
    """The CPU oracle (oracle/refmodel.py: a port of the reference path -- test infrastructure, never the product; the reference's own
    Python cannot run on this box) timed on the host cores on the SAME workload as the GPU line: same batch, instances, size, soft edge and
    detail-region guidance. Two legs: train forward + loss + backward (= `value`), and eval forward. Each leg: warm-up steps, then
    timed steps, median reported; a result line is printed after every timed step so that the parent always has the latest median."""
    import copy
    from maggie_amd.network import build_model
    from maggie_amd.utils import config, synth
    from oracle import refmodel
    cores = os.cpu_count() or 1
    # SURVEY 8(d) asks for os.cpu_count() threads; the many small ops of this model regress beyond ~32 torch threads on a 256-core host: both
    # settings are timed (32 first), the faster one is `value` / `cores`, the other is reported next to it
    settings = [args.cpu_threads] if args.cpu_threads > 0 else ([min(cores, 32)] + ([cores] if cores > 32 else []))
    model, _ = build_model(config.model_config(kind))
    sd0 = model.state_dict()
    synth.fill_state_dict_(sd0, 1234)
    n_f = args.frames if kind == 'video' else 1
    b = args.clips if kind == 'video' else args.batch
    size = args.size
    mcfg = copy.deepcopy(config.MODEL_VIDEO if kind == 'video' else config.MODEL_IMAGE)
    n_warm, n_timed = (2, 9) if args.cpu_baseline_full else (2, 5)          # 2 warm-ups + >= 5 timed steps, medians
    res = {'unit': 'instance-frames/s', 'cores': settings[0], 'host_cores': cores, 'kind': 'port', 'per_thread_setting': {}}
    inst = b * n_f * args.instances
    if args.acc_file:
        try:
            torch.set_num_threads(min(cores, 32))
            res['accuracy'] = cpu_accuracy(args.acc_file)
        except Exception as e:
            res['accuracy'] = {'error': '%s: %s' % (type(e).__name__, e)}
    best = {}

    def emit(threads, tr, ev, note=None):
        cur = {'value': round(inst / float(np.median(tr)), 4) if tr else None, 'eval_forward_value': round(inst / float(np.median(ev)), 4) if ev else None,
               'train_median_s': round(float(np.median(tr)), 3) if tr else None, 'eval_median_s': round(float(np.median(ev)), 3) if ev else None,
               'timed_steps': [len(tr), len(ev)]}
        if note:
            cur['note'] = note
        res['per_thread_setting'][str(threads)] = cur
        done = {k: v for k, v in res['per_thread_setting'].items() if v.get('value') is not None}
        if done:                                                  # `value` / `cores`: the faster setting, by its latest median
            k_best = max(done, key=lambda k: done[k]['value'])
            res['value'], res['eval_forward_value'], res['cores'] = done[k_best]['value'], done[k_best].get('eval_forward_value'), int(k_best)
        res['sample'] = ('oracle/refmodel.py fp32 (torch threads: %s of %d host cores tried, `value` / `cores` = the faster setting), same workload as the GPU '
                         'line: %dx%d, batch %d x %d frame(s), %d instances, %s-guided detail region, active ratio %.3f; train leg = forward + losses + '
                         'backward (no optimizer): %d warm-up + %d timed steps per setting, medians; eval-forward leg likewise' % (
                             '/'.join(str(t) for t in settings), cores, size, size, b, n_f, args.instances,
                             'ground-truth' if args.workload == 'gt' else 'prediction', res.get('active_ratio', float('nan')), n_warm, n_timed))
        print('CPU_BASELINE ' + json.dumps(res), flush=True)

    batch = synth.synthetic_batch(b, n_f, args.instances, size, size, seed=1234, train=True, it=args.iter, max_inst=10, edge=args.edge * size / 512.0)
    ev_batch = {k: (v[:, :, :args.instances] if k == 'mask' else v) for k, v in batch.items() if k in ('image', 'mask')}
    if res.get('accuracy') is not None:
        print('CPU_BASELINE ' + json.dumps(dict(res, value=None, sample='accuracy leg only so far')), flush=True)
    for si, threads in enumerate(settings):
      torch.set_num_threads(threads)
      tr_s, ev_s = [], []
      slow = False
      watchdog = None
      if si > 0 and not args.cpu_baseline_full:
          n_warm, n_timed = 1, 1                                   # the second setting only has to show which one is faster (one probe step)
          # ... and must not hold the bench line back: on a 256-core host one step of this model's many small ops under 256 torch threads takes
          # minutes. If its warm-up step is not done after `cap` seconds the setting is reported as unfinished and the worker ends here.
          import threading
          cap = max(60.0, 4.0 * inst / res['value']) if res.get('value') else 120.0

          def bail(threads=threads, cap=cap):
              res['per_thread_setting'][str(threads)] = {'value': None, 'note': 'warm-up step not finished after %.0f s (the other setting: %.1f s per step): abandoned' % (
                  cap, inst / res['value'] if res.get('value') else float('nan'))}
              print('CPU_BASELINE ' + json.dumps(res), flush=True)
              os._exit(0)
          watchdog = threading.Timer(cap, bail)
          watchdog.daemon = True
          watchdog.start()
      for i in range(n_warm + n_timed):
        sd = {k: v.clone() for k, v in sd0.items()}
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(('_u', '_v', 'running_mean', 'running_var')):
                v.requires_grad_(True)
        np.random.seed(i)
        t0 = time.perf_counter()
        out, loss = refmodel.maggie_forward(sd, mcfg, batch, True)
        loss['total'].backward()
        dt = time.perf_counter() - t0
        res['active_ratio'] = round(float(out['detail_mask'].float().mean()) * 10.0 / args.instances, 4)
        if watchdog is not None:
            watchdog.cancel()
            watchdog = None
        if i >= n_warm:
            tr_s.append(dt)
            emit(threads, tr_s, ev_s)
        elif si > 0 and res.get('value') and dt > 2.5 * inst / res['value']:
            # this setting's warm-up step alone is far slower than the first setting's median: record that and do not spend minutes on it
            emit(threads, [dt], [], note='warm-up step only (%.1f s): more than 2.5x slower than the other setting, not timed further' % dt)
            slow = True
            break
      if slow:
          continue
      for i in range(n_warm + n_timed):
        sd = {k: v.clone() for k, v in sd0.items()}
        t0 = time.perf_counter()
        with torch.no_grad():
            refmodel.maggie_forward(sd, mcfg, ev_batch, False)
        dt = time.perf_counter() - t0
        if i >= n_warm:
            ev_s.append(dt)
            emit(threads, tr_s, ev_s)
    return res


if __name__ == '__main__':
    main()
