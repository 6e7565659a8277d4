"""Per-step kernel-family summary of the LAST `steps` steps of a rocprofv3 --kernel-trace CSV (the timed hipGraph-replay steps of bench.py).
usage: python tools/trace_summary.py <kernel_trace.csv> <steps> <ms_per_step> [top]"""
import csv, sys, collections

path, steps, ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 45
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
t_end = rows[-1][1]
win = [r for r in rows if r[0] >= t_end - steps * ms * 1e6]


def family(nm):
    if 'conv_halo3' in nm: return 'maggie: conv_halo3 (3x3 s1 halo tiles, producer / consumer waves)'
    if 'igemm_fprop' in nm and 'persistent' in nm: return 'maggie: igemm_fprop persistent (sparse head)'
    if 'igemm_fprop_halo' in nm: return 'maggie: igemm_fprop_halo (3x3 s1 halo tiles)'
    if 'igemm_fprop_async' in nm: return 'maggie: igemm_fprop_async (direct-to-LDS im2col)'
    if 'igemm_fprop' in nm: return 'maggie: igemm_fprop (register-staged im2col)'
    if 'igemm_wgrad_halo' in nm: return 'maggie: igemm_wgrad_halo (3x3 s1, all taps per tile)'
    if 'igemm_wgrad_gather9' in nm: return 'maggie: igemm_wgrad_gather9 (sparse head, all taps per tile)'
    if 'wgrad_reduce' in nm: return 'maggie: wgrad_reduce (split slabs -> dW)'
    if 'igemm_wgrad' in nm: return 'maggie: igemm_wgrad (per-tap tiles)'
    if 'anonymous namespace' in nm and 'at::native' not in nm and 'ck::' not in nm and 'Cat' not in nm and 'multi_tensor' not in nm \
            and 'layer_norm' not in nm and 'GammaBeta' not in nm and 'cuCompute' not in nm and 'reflection' not in nm:
        return 'maggie: ' + nm.split('::')[1].split('<')[0].split('(')[0]
    if nm.startswith('Cijk'): return 'hipBLASLt gemm'
    if 'at::native' in nm or 'multi_tensor' in nm or 'layer_norm' in nm: return 'torch elementwise/reduce'
    if 'rocclr' in nm: return 'rocclr copy/fill'
    if 'mg_zero_words' in nm: return 'maggie: mg_zero_words'
    return 'other: ' + nm[:50]


busy = sum(e - s for s, e, _ in win)
span = win[-1][1] - win[0][0]
print('window: %d launches, span %.2f ms = %.2f ms/step; busy %.2f ms/step (%.1f%%), idle %.2f ms/step; launches/step %.0f' % (
    len(win), span / 1e6, span / 1e6 / steps, busy / 1e6 / steps, 100.0 * busy / span, (span - busy) / 1e6 / steps, len(win) / steps))
g = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    d = g[family(n)]
    d[0] += 1; d[1] += e - s
for k, (c, t) in sorted(g.items(), key=lambda kv: -kv[1][1])[:top]:
    print('  %-58s %7.1f launches/step %7.3f ms/step  %6.1f us avg' % (k, c / steps, t / 1e6 / steps, t / 1e3 / c))
if '--torch' in sys.argv:
    tg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in win:
        if family(n) in ('torch elementwise/reduce', 'rocclr copy/fill'):
            d = tg[n[:150]]
            d[0] += 1; d[1] += e - s
    print('torch / copy kernels:')
    for k, (c, t) in sorted(tg.items(), key=lambda kv: -kv[1][1])[:30]:
        print('  %7.1f /step %7.3f ms/step  %s' % (c / steps, t / 1e6 / steps, k))
