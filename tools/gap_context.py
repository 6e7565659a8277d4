"""Kernels around the largest GPU idle gaps of a rocprofv3 --kernel-trace CSV. usage: gap_context.py trace.csv [n_gaps] [context]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
n_gaps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rows = rows[len(rows) // 2:]                                      # second half: steady state
gaps = []
end = rows[0][1]
for i in range(1, len(rows)):
    g = rows[i][0] - end
    if g > 0:
        gaps.append((g, i))
    end = max(end, rows[i][1])
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').replace('at::native::', '')[:100]
for g, i in sorted(gaps, reverse=True)[:n_gaps]:
    print('==== gap %.1f us before kernel #%d' % (g / 1e3, i))
    for j in range(max(0, i - ctx), min(len(rows), i + ctx)):
        s, e, n = rows[j]
        print('%s %9.1f us  dur %7.1f  %s' % ('>>' if j == i else '  ', (s - rows[i][0]) / 1e3, (e - s) / 1e3, short(n)))
