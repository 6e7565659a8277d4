"""Would a layer's weight-gradient GEMM and the NEXT layer's BatchNorm-backward reduce overlap if they ran side by side? (they are independent: same stream today,
back to back). Serial vs two graph branches, replayed from a hipGraph; upper bound of what a horizontally fused launch could give."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maggie_amd import kernels as K
dev = torch.device('cuda:0')


def run(n, hw, c):
    M = n * hw * hw
    x = torch.randn(M, c, device=dev).bfloat16()
    dy = torch.randn(M, c, device=dev).bfloat16()
    dz = torch.randn(M, c, device=dev).bfloat16()
    xb = torch.randn(M, c, device=dev).bfloat16()
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    sc, sh, mean, invstd = K.bn_finalize(K.colstats(xb), M, gamma, beta, None, None, 0.1, 1e-5)
    z = K.affine_act(xb, sc, sh, act=1)
    kw = dict(mode=K.MODE_CONV, N=n, Hin=hw, Win=hw, Hout=hw, Wout=hw, R=3, S=3, stride=1, pad=1, dil=1)
    s2 = torch.cuda.Stream()
    wg = lambda: K.conv_wgrad(x, dy, cout=c, out_dtype=torch.bfloat16, **kw)
    red = lambda: K.bn_backward(dz, z, xb, sc, mean, invstd, M, act=1, reduce_only=True)

    def serial():
        wg(); red()

    def conc():
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            wg()
        red()
        torch.cuda.current_stream().wait_stream(s2)

    def t(fn, it=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / it * 1e3

    res = []
    for fn in (wg, red, serial, conc):
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            fn()
            with torch.cuda.graph(g, stream=st):
                for _ in range(10): fn()
        torch.cuda.current_stream().wait_stream(st)
        res.append(t(g.replay) / 10)
    print('N=%d %dx%d C=%d (graph replay, us per item): wgrad %.1f | bn reduce %.1f | serial pair %.1f | two branches %.1f' % ((n, hw, hw, c) + tuple(res)))


from maggie_amd import hip
hip.set_deterministic(False) if '--nondet' in sys.argv else None
run(4, 64, 128)
run(4, 32, 256)
run(4, 128, 64)
