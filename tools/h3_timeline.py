"""Phase timeline of the halo3 kernel from s_memtime stamps (variant build: tools/build_variant.sh h3dbg conv_halo3.hip "-DMG_H3_TIMING").
usage: MAGGIE_LIB_PATH=maggie_amd/_variants/lib_h3dbg.so python tools/h3_timeline.py N Cin Cout HW TH,BN,NS [mode]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maggie_amd import hip
from maggie_amd import kernels as K
N, Cin, Cout, HW = map(int, sys.argv[1:5])
th, bn, ns = map(int, sys.argv[5].split(','))
mode = int(sys.argv[6]) if len(sys.argv) > 6 else 0
lib = hip.lib()
lib.mg_set_halo3(ctypes.c_int(1)); lib.mg_set_halo3_cfg(ctypes.c_int(th), ctypes.c_int(bn), ctypes.c_int(ns))
dev = torch.device('cuda:0')
x = torch.randn(N * HW * HW, Cin, device=dev).bfloat16()
w = torch.randn(Cout, 9, Cin, device=dev).bfloat16()
cold = os.environ.get('H3_COLD') == '1'
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev) if cold else None
for _ in range(5):
    if cold:
        # the in-step situation: x was written by the PREVIOUS kernel (write-through to the memory side: L2 is not coherent across XCDs) and
        # 0.5 GB of other traffic went by since the weights were produced -- nothing of this launch's operands is in any L2
        x = (x.float() * 1.0).bfloat16()
        w = (w.float() * 1.0).bfloat16()
        big.fill_(1)
    K.conv_fprop(x, w, mode=mode, N=N, Hin=HW, Win=HW, R=3, S=3, stride=1, pad=1, dil=1)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 1024)()
assert lib.mg_h3_debug_read(buf) == 0
a = np.array(buf[:], dtype=np.int64).reshape(32, 32)
names = ['start', 'issued'] + sum([['s%d top' % s, 's%d go' % s] for s in range(6)], []) + ['loop end', 'stored', 'rows mapped', 'px0', 'px1', 'px2', 'px3']
rt0 = a[a[:, 30] > 0, 30].min()
for b in range(32):
    if a[b, 0] == 0: continue
    print('work %4d: ' % (b * 32) + '  '.join('%s %d' % (nm, a[b, i] - a[b, 0]) for i, nm in enumerate(names) if a[b, i]) +
          '   | wall: start +%.2f us, end +%.2f us' % ((a[b, 30] - rt0) / 100.0, (a[b, 31] - rt0) / 100.0))
