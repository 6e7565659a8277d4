"""Phase timeline of the halo fprop kernel from s_memtime stamps (debug build: hipcc -DMG_HALO_TIMING -> tools/_dbg/libmaggie_dbg.so).
usage: python tools/halo_timeline.py N Cin Cout HW"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maggie_amd import hip
hip.LIB_PATH = os.environ.get('MAGGIE_LIB_PATH') or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'maggie_amd', '_variants', 'lib_dbg.so')
from maggie_amd import kernels as K
N, Cin, Cout, HW = map(int, sys.argv[1:5])
wgrad = len(sys.argv) > 5 and sys.argv[5] == 'wgrad' 
dev = torch.device('cuda:0')
x = torch.randn(N * HW * HW, Cin, device=dev).bfloat16()
w = torch.randn(Cout, 9, Cin, device=dev).bfloat16()
gy = torch.randn(N * HW * HW, Cout, device=dev).bfloat16()
for _ in range(5):
    if wgrad:
        K.conv_wgrad(x, gy, cout=Cout, mode=K.MODE_CONV, N=N, Hin=HW, Win=HW, Hout=HW, Wout=HW, R=3, S=3, stride=1, pad=1, dil=1)
    else:
        K.conv_fprop(x, w, mode=K.MODE_CONV, N=N, Hin=HW, Win=HW, R=3, S=3, stride=1, pad=1, dil=1)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 768)()
assert hip.lib().mg_debug_read(buf) == 0
a = np.array(buf[:], dtype=np.int64).reshape(32, 24)
names = ['start', 'issued'] + sum([['s%d top' % s, 's%d ready' % s] for s in range(6)], []) + ['loop end', 'end', 'ep setup', 'sC written', 'rows stored', '', '', '', '', '']
t0 = a[:, 0].min()
for b in range(min(8, 32)):
    if a[b, 0] == 0: continue
    print('block %4d: ' % (b * 64) + '  '.join('%s %d' % (nm, a[b, i] - a[b, 0]) for i, nm in enumerate(names) if a[b, i]) + '   (start +%d)' % (a[b, 0] - t0))
