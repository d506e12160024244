#!/usr/bin/env python3
"""Triples layout on the general kernel against planes, under option sets given as 'name=value,...' arguments: number of obs / reward / done
words that differ after ONE launch from reset (config 3, 65 536 envs x 120 steps), several repetitions.  Debugging aid."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from griduniverse_amd import Engine, GridSpec, _lib  # noqa: E402
from benchlib.workloads import build_workload  # noqa: E402

spec = GridSpec.from_env(build_workload('c3')[0])
N, T = 65536, 120


def run(opts):
    with Engine(N, spec, seed=11, env_id0=5) as eng:
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(T, 'uniform', True, True)
        return eng.read_trajectory(0, T)


ref = run(dict(traj_layout=0, rollout_rows=0))
for arg in sys.argv[1:]:
    opts = dict(traj_layout=1, rollout_rows=0)
    opts.update({k: int(v) for k, v in (item.split('=') for item in arg.split(',') if item)})
    counts = []
    for rep in range(3):
        got = run(opts)
        counts.append(tuple(int((ref[k] != got[k]).sum()) for k in ('obs', 'reward', 'done')))
        if rep == 0 and counts[0][0]:
            w = np.argwhere(ref['obs'] != got['obs'])
            lanes = np.bincount(w[:, 1] % 64, minlength=64)
            print('   lanes (mod 64) of the wrong obs words:', {int(i): int(c) for i, c in enumerate(lanes) if c}, ' steps:', sorted(set(w[:, 0].tolist()))[:12])
    print('%-60s differing (obs, reward, done) words in 3 runs: %s' % (arg, counts), flush=True)
