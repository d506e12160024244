#!/usr/bin/env python3
"""What the memory-side counters say about the rollout launch unthrottled, on a schedule, and on too short a schedule (run under
rocprofv3 --pmc by tools/pacing_pmc.sh; with `parse <dir>` it joins the passes).  One engine, the first trajectory allocation it
gets, 65 536 envs x 1000 steps; 12 launches per setting, fixed periods (no calibration: the dispatches line up across passes)."""
import csv
import glob
import os
import sys
from collections import defaultdict

SETTINGS = [('unthrottled', 0), ('schedule 172', 172), ('schedule 150 (too short)', 150)]
M = 12

if len(sys.argv) > 2 and sys.argv[1] == 'parse':
    root = sys.argv[2]
    for d in sorted(glob.glob(os.path.join(root, 'pass*'))):
        files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
        if not files:
            continue
        per, dur = defaultdict(lambda: defaultdict(float)), {}
        for row in csv.DictReader(open(files[0])):
            if 'gu_rollout_kernel<' not in row['Kernel_Name']:
                continue
            i = int(row['Dispatch_Id'])
            per[i][row['Counter_Name']] += float(row['Counter_Value'])
            dur[i] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
        ids = sorted(per)
        counters = sorted({c for v in per.values() for c in v})
        print('== %s: %d rollout dispatches' % (os.path.basename(d), len(ids)))
        print('%-28s %9s  %s' % ('setting', 'us(krn)', '  '.join(counters)))
        for k, (name, _) in enumerate(SETTINGS):
            mine = ids[k * M + 2:(k + 1) * M]  # (the first two of a setting settle)
            if not mine:
                continue
            print('%-28s %9.1f  %s' % (name, sum(dur[i] for i in mine) / len(mine),
                                       '  '.join('%*.4g' % (len(c), sum(per[i][c] for i in mine) / len(mine)) for c in counters)))
    sys.exit(0)

import random  # noqa: E402

import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd import _lib  # noqa: E402

random.seed(123)
np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
_lib.set_default_option('traj_candidates', 1)
eng = gua.Engine(65536, gua.GridSpec.from_env(env), seed=123)
eng.reset()
eng.reserve_trajectory(1000)
for name, period in SETTINGS:
    eng.set_option('rollout_pace', period)
    for _ in range(M):
        eng.rollout(1000, 'uniform', True, True)
    eng.sync()
eng.close()
