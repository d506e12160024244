#!/usr/bin/env python3
"""Joins the per-dispatch counters of tools/placement_pmc.sh with the per-buffer times the program printed.
Usage: python tools/placement_pmc_parse.py <dir with pass*/ and pass*.stdout>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for out in sorted(glob.glob(os.path.join(root, 'pass*.stdout'))):
    name = os.path.basename(out)[:-7]
    times = [float(line.split()[2]) for line in open(out) if line.startswith('buffer ')]
    files = glob.glob(os.path.join(root, name, '**', '*counter_collection.csv'), recursive=True)
    if not files or not times:
        print(name, ': no data (', open(os.path.join(root, name + '.stderr')).read()[-300:].strip(), ')')
        continue
    per = defaultdict(lambda: defaultdict(float))  # dispatch id -> counter -> value
    dur = {}                                       # dispatch id -> kernel duration in us (the profiler's own timestamps)
    for row in csv.DictReader(open(files[0])):
        if 'k3' not in row['Kernel_Name']:
            continue
        per[int(row['Dispatch_Id'])][row['Counter_Name']] += float(row['Counter_Value'])
        dur[int(row['Dispatch_Id'])] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
    ids = sorted(per)
    counters = sorted({c for d in per.values() for c in d})
    print('== %s: %d dispatches, %d buffers' % (name, len(ids), len(times)))
    print('%-8s %8s %8s  %s' % ('buffer', 'us(evt)', 'us(krn)', '  '.join('%s' % c for c in counters)))
    for b, t in enumerate(times):
        mine = ids[4 * b + 1:4 * b + 4]  # the three timed dispatches
        if len(mine) < 3:
            break
        vals = [sum(per[i][c] for i in mine) / len(mine) for c in counters]
        k = sum(dur[i] for i in mine) / len(mine)
        print('%-8d %8.1f %8.1f  %s  %s' % (b, t, k, '  '.join('%.4g' % v for v in vals), 'FAST' if k < 125 else ''))
