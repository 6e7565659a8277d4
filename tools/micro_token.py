"""Micro-benchmark of the token-side fused layer (mg_token_linear_*) against torch's F.linear + LayerNorm at the decoder's shape (40 x 128 x 128)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from maggie_amd import functional as MF
dev = torch.device('cuda:0')
R, K, N = int(os.environ.get('R', 40)), 128, 128
x = torch.randn(R, K, device=dev, requires_grad=True); W = torch.randn(N, K, device=dev, requires_grad=True); b = torch.randn(N, device=dev, requires_grad=True)
res = torch.randn(R, N, device=dev, requires_grad=True); ln = torch.nn.LayerNorm(N).to(dev); wgt = torch.randn(R, N, device=dev)


def run(fn, it=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(it): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / it * 1e3


def ours():
    y = MF.token_linear(x, W, b, res=res, ln=ln)
    torch.autograd.grad((y * wgt).sum(), [x, W, b, res, ln.weight, ln.bias])


def ours_fwd():
    with torch.no_grad():
        MF.token_linear(x, W, b, res=res, ln=ln)


def ref():
    y = ln(res + F.linear(x, W, b))
    torch.autograd.grad((y * wgt).sum(), [x, W, b, res, ln.weight, ln.bias])


def ref_fwd():
    with torch.no_grad():
        ln(res + F.linear(x, W, b))


print('R=%d: fused fwd %.1f us, fwd+bwd %.1f us | torch fwd %.1f us, fwd+bwd %.1f us (inside a replayed graph, per layer)' % (R, run(ours_fwd), run(ours), run(ref_fwd), run(ref)))
