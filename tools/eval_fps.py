"""Inference throughput (eval forward, no grad) of the image and video models. python tools/eval_fps.py"""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth
dev = torch.device('cuda:0')
for kind, b, n_f in (('image', 1, 1), ('image', 4, 1), ('video', 1, 3)):
    for graphs in (False, True):
        for dt in ('fp32', 'bf16'):
            model, _ = build_model(config.model_config(kind))
            sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
            model.to(dev).eval(); model.hip_graphs = graphs
            batch = synth.synthetic_batch(b, n_f, 2, 512, 512, seed=7, train=False)
            batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
            def run():
                with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=dt == 'bf16'):
                    return model(batch)
            for _ in range(4):
                out = run()
            torch.cuda.synchronize(); t = time.perf_counter(); n = 20
            for _ in range(n):
                out = run()
            torch.cuda.synchronize(); dtm = (time.perf_counter() - t) / n
            print('%s b=%d T=%d %s graphs=%d: %.2f ms/forward, %.1f frames/s, %.1f instance-frames/s, active %.3f' % (
                kind, b, n_f, dt, graphs, 1e3 * dtm, b * n_f / dtm, 2 * b * n_f / dtm, float(out['detail_mask'].float().mean())))
