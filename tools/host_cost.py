"""Host-side cost (us per call, GPU work is tiny and asynchronous) of the detail-stage building blocks: where do the ~12 us per launch go?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maggie_amd import kernels as K, functional as MF, hip
dev = torch.device('cuda:0')
M, C = 2048, 64
x = torch.randn(M, C, device=dev).bfloat16().requires_grad_(True)
w = (torch.randn(C, 9, C, device=dev) * 0.05).bfloat16().requires_grad_(True)
w1 = (torch.randn(C, 1, C, device=dev) * 0.05).bfloat16().requires_grad_(True)
nbr = torch.randint(-1, M, (M, 9), device=dev, dtype=torch.int32)
bn = torch.nn.BatchNorm1d(C).to(dev).train()
bias = torch.zeros(C, device=dev)

def t(name, fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print('%-58s %7.1f us' % (name, dt))

p = hip.ptr(x)
t('hip.stream()', lambda: hip.stream())
t('hip.ptr(x)', lambda: hip.ptr(x))
t('torch.empty((M, C))', lambda: torch.empty((M, C), dtype=torch.bfloat16, device=dev))
t('x.view(-1, C)', lambda: x.view(-1, C))
t('hip.call mg_bias_act_bwd (9 args, raw)', lambda: hip.call('mg_bias_act_bwd', p, None, None, K.c_int(1), K.c_int(0), K.c_int(C), None, hip.stream()))
with torch.no_grad():
    t('K.conv_fprop gather 3x3 (no autograd)', lambda: K.conv_fprop(x, w, mode=K.MODE_GATHER, nbr=nbr, R=3, S=3))
    t('K.conv_wgrad gather 3x3', lambda: K.conv_wgrad(x, x, cout=C, mode=K.MODE_GATHER, nbr=nbr, R=3, S=3, out_dtype=torch.bfloat16))
    t('MF.gather_conv under no_grad', lambda: MF.gather_conv(x, w, nbr, nbr, True, 3))
    t('MF.batch_norm_act under no_grad (train stats)', lambda: (MF.ARENA.reset(dev), MF.batch_norm_act(x, bn, MF.ACT_LRELU)))
t('MF.gather_conv fwd with autograd', lambda: MF.gather_conv(x, w, nbr, nbr, True, 3))
t('MF.linear_rows fwd with autograd (+bias, relu)', lambda: MF.linear_rows(x, w1, bias, pre_relu=True))
t('MF.batch_norm_act fwd with autograd', lambda: (MF.ARENA.reset(dev), MF.batch_norm_act(x, bn, MF.ACT_LRELU)))
def fb():
    y = MF.gather_conv(x, w, nbr, nbr, True, 3)
    y.backward(x.detach())
t('gather_conv fwd+bwd (dgrad + wgrad)', fb, 200)
def fb2():
    MF.ARENA.reset(dev)
    y = MF.batch_norm_act(x, bn, MF.ACT_LRELU)
    y.backward(x.detach())
t('batch_norm_act fwd+bwd (+arena reset)', fb2, 200)
def fb3():
    y = torch.sigmoid(x.float()).to(x.dtype) * x
    y.backward(x.detach())
t('torch sigmoid(x.float()).to()*x fwd+bwd', fb3, 200)
