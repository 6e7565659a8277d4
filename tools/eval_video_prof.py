"""Video eval forward under a kernel trace: python tools/eval_video_prof.py  (run under rocprofv3 --kernel-trace --stats)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maggie_amd.network import build_model
from maggie_amd.utils import config, synth
dev = torch.device('cuda:0')
model, _ = build_model(config.model_config('video'))
sd = model.state_dict(); synth.fill_state_dict_(sd, 1234); model.load_state_dict(sd)
model.to(dev).eval()
batch = synth.synthetic_batch(1, 3, 2, 512, 512, seed=7, train=False)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
for _ in range(8):
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        out = model(batch)
torch.cuda.synchronize()
