"""Throughput of the SURVEY 8f kernels (input pre-processor, post-path, validation metrics) at the bench geometry against the HBM
roofline, with the CPU restatement (oracle/, numpy) timed beside each. usage: python tools/aux_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maggie_amd.utils import preprocess as P, metric as M, postprocessing as PP
from oracle import preprocess as OP, metric as OM, postprocess as OPP

dev = torch.device('cuda:0')
PEAK = 8000.0                                                     # GB/s, MI355X HBM3E


def gpu_us(fn, it=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def cpu_ms(fn):
    t = time.perf_counter()
    fn()
    return (time.perf_counter() - t) * 1e3


rs = np.random.RandomState(0)
b, n_i, H, W = 4, 2, 512, 512
frames = rs.randint(0, 256, size=(b, H, W, 3)).astype(np.uint8)
planes = rs.randint(0, 256, size=(b, n_i, H, W)).astype(np.uint8)
f_d, p_d = torch.from_numpy(frames).to(dev), torch.from_numpy(planes).to(dev)
rows = []
us = gpu_us(lambda: P.normalize_frames(f_d, device=dev))
rows.append(('mg_preprocess_image', b * H * W * (3 + 12), us, cpu_ms(lambda: OP.normalize_frames(frames, P.IMAGENET_MEAN, P.IMAGENET_STD))))
us = gpu_us(lambda: P.scale_planes(p_d, 10, [3, 7], None, 5, device=dev))
rows.append(('mg_preprocess_planes (alpha -> 10 slots)', b * H * W * (n_i + 40), us, cpu_ms(lambda: OP.scale_planes(planes, 10, [3, 7], None, 5))))
x = torch.rand(b * n_i, H, W, device=dev)
info = [{'name': 'resize', 'ori_size': (720, 1280)}, {'name': 'padding', 'pad_size': (32, 0)}]
us = gpu_us(lambda: PP.reverse_transform_tensor(x, info, snap=True))
xc = x.cpu()
rows.append(('mg_postprocess_alpha (-> 720x1280)', b * n_i * (H * W + 720 * 1280) * 4, us, cpu_ms(lambda: OPP.reverse_transform_tensor(xc, info))))
pred = torch.rand(3, n_i, H, W, device=dev)
gt = (pred + 0.05 * torch.randn_like(pred)).clamp(0, 1)
tri = torch.randint(0, 3, pred.shape, device=dev).float()
pn, gn, tn = pred.cpu().numpy(), gt.cpu().numpy(), tri.cpu().numpy()
nb = pred.numel() * 4 * 3
for name, fn in (('SAD', OM.sad), ('Grad', OM.grad), ('dtSSD', OM.dtssd)):
    m = M.build_metric([name])[name]
    us = gpu_us(lambda: m.update(pred, gt, tri), it=10)
    rows.append(('metric %s (incl. the fp64 read-back)' % name, nb if name != 'Grad' else nb + pred.numel() * 8, us, cpu_ms(lambda: fn(pn, gn, tn))))
print('%-46s %10s %10s %8s %12s' % ('kernel', 'alg MB', 'GPU us', 'GB/s', 'CPU numpy ms'))
for name, nbytes, us, cms in rows:
    print('%-46s %10.2f %10.1f %8.0f %12.1f   (%.1f%% of %.0f GB/s)' % (name, nbytes / 1e6, us, nbytes / us / 1e3, cms, 100 * nbytes / us / 1e3 / PEAK, PEAK))
