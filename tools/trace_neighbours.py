"""Where in the step do the torch / rocclr launches sit? For every non-library kernel of ONE replayed step of a rocprofv3 --kernel-trace CSV:
its duration, grid size and the library kernels launched right before and after it (which name the place in the step).
usage: python tools/trace_neighbours.py <kernel_trace.csv> <ms_per_step>"""
import csv, sys, collections
path, ms = sys.argv[1], float(sys.argv[2])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '?'))))
rows.sort()
t_end = rows[-1][1]
win = [r for r in rows if r[0] >= t_end - ms * 1e6]


def short(nm):
    if 'anonymous namespace' in nm and 'at::native' not in nm:
        return nm.split('::')[1].split('<')[0].split('(')[0]
    if 'rocclr' in nm:
        return nm.replace('__amd_rocclr_', 'rocclr:')
    if 'at::native' in nm:
        for key in ('FillFunctor', 'CUDAFunctor_add', 'direct_copy', 'bfloat16_copy', 'bfloat16tofloat32', 'CatArrayBatchedCopy', 'multi_tensor_apply', 'reduce_kernel', 'where', 'Compare', 'compare'):
            if key in nm:
                return 'torch:' + key
        return 'torch:' + nm[:60]
    return nm[:40]


def is_lib(nm):
    return 'anonymous namespace' in nm and 'at::native' not in nm or 'mg_zero_words' in nm


agg = collections.Counter()
for i, (s, e, nm, grid) in enumerate(win):
    if is_lib(nm):
        continue
    prev = next((short(win[j][2]) for j in range(i - 1, -1, -1) if is_lib(win[j][2])), '-')
    nxt = next((short(win[j][2]) for j in range(i + 1, len(win)) if is_lib(win[j][2])), '-')
    print('%6.1f us  grid %-9s %-28s after %-28s before %s' % ((e - s) / 1e3, grid, short(nm), prev, nxt))
    agg[short(nm)] += 1
print(sum(agg.values()), 'non-library launches in the step:', dict(agg))
