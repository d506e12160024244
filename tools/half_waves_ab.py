#!/usr/bin/env python3
"""Int32-row launches of the transition-row kernel on the config-4 grid by batch size: whole waves against half waves (32 envs per
wave, twice the waves), planes against triples; us per 1000-step launch under the closed loop of the store pacing.
Usage (GPU box): python tools/half_waves_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402
from benchlib.workloads import build_workload  # noqa: E402

T, K = 1000, 20
spec = gua.GridSpec.from_env(build_workload('c4')[0])
for N in (16384, 24576, 32768, 49152, 65536):
    row = []
    for half, layout in ((0, 0), (1, 0), (0, 1), (1, 1)):
        eng = gua.Engine(N, spec, seed=4, env_id0=N)
        eng.set_option('rollout_rows', 1)
        eng.set_option('rollout_half_waves', half)
        eng.set_option('traj_layout', layout)
        eng.reset()
        eng.reserve_trajectory(T)
        best = 1e9
        for rep in range(2):
            for _ in range(200):
                eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
            eng.sync()
            eng.timer_begin()
            for _ in range(K):
                eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
            best = min(best, eng.timer_end() / K * 1e3)
        eng.close()
        row.append('half=%d layout=%d %.1f us (%.3f of peak)' % (half, layout, best, 12.0 * N * T / best / 1e6 / 8000.0))
    print('N = %6d  ' % N + '   '.join(row), flush=True)
