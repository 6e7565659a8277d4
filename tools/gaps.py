"""GPU idle-gap analysis of a rocprofv3 --kernel-trace CSV: where does the device wait for the host?
usage: python tools/gaps.py <kernel_trace.csv> [steps_in_window] [ms_per_step]"""
import csv, sys, collections

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ms = float(sys.argv[3]) if len(sys.argv) > 3 else 36.0
t_end = rows[-1][1]
win = [r for r in rows if r[0] >= t_end - steps * ms * 1e6]
busy = sum(e - s for s, e, _ in win)
span = win[-1][1] - win[0][0]
print('window: %d kernels, span %.2f ms, busy %.2f ms (%.1f%%), per step: busy %.2f ms idle %.2f ms' % (
    len(win), span / 1e6, busy / 1e6, 100.0 * busy / span, busy / 1e6 / steps, (span - busy) / 1e6 / steps))
short = lambda n: n.split('(')[0][-60:]
by_next = collections.defaultdict(lambda: [0, 0])
hist = collections.Counter()
big = []
prev_end, prev_name = win[0][1], win[0][2]
for s, e, n in win[1:]:
    g = s - prev_end
    if g > 0:
        d = by_next[short(n)]
        d[0] += g; d[1] += 1
        b = 0 if g < 2000 else 1 if g < 5000 else 2 if g < 10000 else 3 if g < 20000 else 4 if g < 50000 else 5 if g < 200000 else 6
        hist[b] += g
        if g >= 50000:
            big.append((g, short(prev_name), short(n)))
    if e > prev_end:
        prev_end, prev_name = e, n
names = ['<2us', '2-5us', '5-10us', '10-20us', '20-50us', '50-200us', '>200us']
print('idle time by gap size (ms/step):', {names[k]: round(v / 1e6 / steps, 3) for k, v in sorted(hist.items())})
print('idle attributed to the kernel that arrived late (ms/step, count/step):')
for n, (g, c) in sorted(by_next.items(), key=lambda kv: -kv[1][0])[:25]:
    print('  %8.3f %6.1f  %s' % (g / 1e6 / steps, c / steps, n))
print('largest gaps (us): prev -> next')
for g, p, n in sorted(big, reverse=True)[:30]:
    print('  %8.1f  %s -> %s' % (g / 1e3, p, n))
