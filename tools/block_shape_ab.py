#!/usr/bin/env python3
"""The headline launch (config 3, 65 536 envs x 1000 steps, int32 rows) by workgroup size of the general kernel, under the closed loop of
the store pacing (600 settle launches) and at fixed periods; interleaved repetitions on one box.  Usage (GPU box): python tools/block_shape_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402
from benchlib.workloads import build_workload  # noqa: E402

N, T, K = 65536, 1000, 40
spec = gua.GridSpec.from_env(build_workload('c3')[0])
engines = {}
for bs in (64, 128, 256, 512, 1024):
    eng = gua.Engine(N, spec, seed=123)
    eng.set_option('rollout_block', bs)
    eng.reset()
    eng.reserve_trajectory(T)
    engines[bs] = eng


def timed(eng, settle):
    for _ in range(settle):
        eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
    eng.sync()
    eng.timer_begin()
    for _ in range(K):
        eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
    return eng.timer_end() / K * 1e3


for rep in range(3):
    row = []
    for bs, eng in engines.items():
        eng.set_option('rollout_pace', None)
        loop = timed(eng, 600 if rep == 0 else 200)
        info = eng.rollout_pacing('uniform', True)
        fixed = {}
        for period in (150, 155, 160, 165, 170):
            eng.set_option('rollout_pace', period)
            fixed[period] = round(timed(eng, 20), 1)
        eng.set_option('rollout_pace', None)
        row.append('block %4d: loop %.1f (period %s) fixed %s' % (bs, loop, None if info is None else info['period'], fixed))
    print('\n'.join(row), flush=True)
    print()
for eng in engines.values():
    eng.close()
