"""One graph-replayed step of a rocprofv3 --kernel-trace CSV as a launch list: duration and the gap to the previous kernel's end, for kernels
whose name contains one of the given substrings (all when none given); plus gap statistics of the whole step.
usage: python tools/step_kernels.py <kernel_trace.csv> <ms_per_step> [name ...]"""
import csv, sys

path, ms, names = sys.argv[1], float(sys.argv[2]), sys.argv[3:]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
t_end = rows[-1][1]
win = [r for r in rows if r[0] >= t_end - ms * 1e6]
gaps = [max(0, win[i][0] - win[i - 1][1]) for i in range(1, len(win))]
gs = sorted(gaps)
print('step: %d launches, busy %.3f ms, gaps %.3f ms (median %.2f us, p90 %.2f us, max %.1f us)' % (
    len(win), sum(e - s for s, e, _ in win) / 1e6, sum(gaps) / 1e6, gs[len(gs) // 2] / 1e3, gs[int(len(gs) * 0.9)] / 1e3, gs[-1] / 1e3))
for i, (s, e, n) in enumerate(win):
    if names and not any(k in n for k in names):
        continue
    short = n.replace('void ', '').replace('(anonymous namespace)::', '')
    short = short.split('<')[0].split('(')[0][-60:] if not short.startswith('at::') else short[:110]
    print('%5d %8.2f us  gap %6.2f us  %s' % (i, (e - s) / 1e3, (gaps[i - 1] / 1e3) if i else 0.0, short))
