"""Group a rocprofv3 *_kernel_stats.csv by kernel family (ms per step).  usage: summarize.py stats.csv n_steps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot_calls = sum(int(r['Calls']) for r in rows); tot_ns = sum(int(r['TotalDurationNs']) for r in rows)
print('kernels %d  launches/step %.0f  GPU ms/step %.2f' % (len(rows), tot_calls / n, tot_ns / n / 1e6))
groups = {}
for r in rows:
    nm = r['Name']
    if 'igemm_fprop' in nm: g = 'maggie: igemm_fprop (conv fprop/dgrad/gather)'
    elif 'igemm_wgrad' in nm or 'wgrad_reduce' in nm: g = 'maggie: igemm_wgrad'
    elif 'anonymous namespace' in nm and 'at::native' not in nm and 'ck::' not in nm and 'Cat' not in nm and 'multi_tensor' not in nm and 'layer_norm' not in nm and 'GammaBeta' not in nm and 'cuCompute' not in nm and 'reflection' not in nm:
        g = 'maggie: ' + nm.split('::')[1].split('<')[0].split('(')[0]
    elif nm.startswith('Cijk'): g = 'hipBLASLt gemm (attention/FFN matmuls)'
    elif 'at::native' in nm or 'multi_tensor' in nm or 'layer_norm' in nm: g = 'torch elementwise/reduce/optimizer'
    elif 'rocclr' in nm: g = 'rocclr copy/fill'
    else: g = 'other: ' + nm[:50]
    d = groups.setdefault(g, [0, 0]); d[0] += int(r['Calls']); d[1] += int(r['TotalDurationNs'])
for g, (c, t) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:40]:
    print('  %-56s %7.1f launches/step %7.2f ms/step' % (g, c / n, t / n / 1e6))
