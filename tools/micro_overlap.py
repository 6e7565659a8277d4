"""Do a deep-layer dgrad-shaped GEMM and a wgrad GEMM overlap when issued on two HIP streams? (serial vs concurrent, eager and graphed)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maggie_amd import kernels as K
dev = torch.device('cuda:0')
def run(n, hw, cin, cout):
    x = torch.randn(n, hw, hw, cin, device=dev).bfloat16()
    w = (torch.randn(cout, 9, cin, device=dev) * 0.05).bfloat16()
    kw = dict(mode=K.MODE_CONV, N=n, Hin=hw, Win=hw, Hout=hw, Wout=hw, R=3, S=3, stride=1, pad=1, dil=1)
    y = K.conv_fprop(x.view(-1, cin), w, **kw)
    s2 = torch.cuda.Stream()
    def serial():
        K.conv_fprop(x.view(-1, cin), w, **kw)
        K.conv_wgrad(x.view(-1, cin), y, cout=cout, out_dtype=torch.bfloat16, **kw)
    def conc():
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            K.conv_wgrad(x.view(-1, cin), y, cout=cout, out_dtype=torch.bfloat16, **kw)
        K.conv_fprop(x.view(-1, cin), w, **kw)
        torch.cuda.current_stream().wait_stream(s2)
    def t(fn, it=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / it * 1e3
    res = [t(serial), t(conc)]
    for fn in (serial, conc):                                     # the same, replayed from a hipGraph (10 pairs per graph)
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            fn()
            with torch.cuda.graph(g, stream=st):
                for _ in range(10): fn()
        torch.cuda.current_stream().wait_stream(st)
        res.append(t(g.replay, 20) / 10)
    print('N=%d hw=%d cin=%d cout=%d: eager serial %.1f us, eager 2-stream %.1f us | graph serial %.1f us, graph 2-branch %.1f us' % ((n, hw, cin, cout) + tuple(res)))
run(4, 32, 512, 256)
run(4, 16, 512, 512)
run(4, 64, 128, 128)
run(4, 128, 64, 64)
