"""Print the top rows of a rocprofv3 kernel_stats.csv with short kernel names: python tools/kstats.py <dir or csv> [n]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, '**', '*kernel_stats.csv'), recursive=True))[0]
for r in list(csv.DictReader(open(path)))[:top]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void at::native::', '').split('(')[0][:64]
    print('%-64s %6s  avg %8.2f us  min %7.2f  max %7.2f  %5s%%' % (n, r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3,
                                                                  float(r['MaxNs']) / 1e3, r['Percentage']))
