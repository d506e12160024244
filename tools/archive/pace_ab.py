#!/usr/bin/env python3
"""The rollout kernel with and without its store pacing on MANY trajectory buffers of one process -- each engine takes the first
allocation it gets (no placement search), so both write-rate classes show up.  Per buffer: the bare store probe, us per launch
unpaced (option rollout_pace = 0), with the calibrated schedule (the default), what the calibration found, and the launch time at
fixed periods (10 ns ticks per 16 steps).  PACE_AB_K = launches per figure (10), PACE_AB_BLOCK = workgroup size, PACE_AB_XCD = XCD-aware block order, PACE_AB_ROWS / PACE_AB_NO_ROWS force / forbid the
transition-row kernel, GU_LIB_PATH another build of the library (make variant ...).
    python tools/pace_ab.py [n_buffers] [envs] [fixed periods ...]"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd import _lib  # noqa: E402

n_buf = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
fixed = [int(x, 0) for x in sys.argv[3:]]
T, K = 1000, int(os.environ.get('PACE_AB_K', '10'))
random.seed(123)
np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
spec = gua.GridSpec.from_env(env)
_lib.set_default_option('traj_candidates', 1)
import os
if os.environ.get('PACE_AB_XCD'):
    _lib.set_default_option('rollout_xcd', 1)  # XCD-aware env-block order
if os.environ.get('PACE_AB_BLOCK'):
    _lib.set_default_option('rollout_block', int(os.environ['PACE_AB_BLOCK']))  # workgroup size of the general kernel
if os.environ.get('PACE_AB_ROWS'):
    _lib.set_default_option('rollout_rows', int(os.environ['PACE_AB_ROWS']))  # 1: the transition-row kernel wherever eligible, 2: without its pair tables
if os.environ.get('PACE_AB_NO_ROWS'):
    _lib.set_default_option('rollout_rows', 0)  # the general kernel also where the transition-row kernel would take the launch
engines = []
for b in range(n_buf):
    eng = gua.Engine(N, spec, seed=123)
    eng.reset()
    eng.reserve_trajectory(T)
    engines.append(eng)


def timed(eng):
    for _ in range(2):
        eng.rollout(T, 'uniform', True, True)
    eng.sync()
    eng.timer_begin()
    for _ in range(K):
        eng.rollout(T, 'uniform', True, True)
    return eng.timer_end() / K * 1e3


print('%-4s %8s %9s %9s  %s' % ('buf', 'probe', 'unpaced', 'paced', 'calibration') + ''.join(' %7s' % ('t=%d' % p) for p in fixed))
for b, eng in enumerate(engines):
    probe = eng.probe_trajectory() * 1e3
    eng.set_option('rollout_pace', 0)
    unpaced = timed(eng)
    eng.set_option('rollout_pace', -2)
    paced = timed(eng)
    info = eng.rollout_pacing()
    row = []
    for p in fixed:
        eng.set_option('rollout_pace', p)
        row.append(timed(eng))
    eng.set_option('rollout_pace', -2)
    print('%-4d %8.1f %9.1f %9.1f  %s' % (b, probe, unpaced, paced, info) + ''.join(' %7.1f' % v for v in row), flush=True)
# pacing never changes a result: same seed, same launch, with and without
keys = []
for eng in engines[:2]:
    for p in (0, None):
        eng.set_option('rollout_pace', p)
        eng.seed(123)
        eng.reset()
        eng.rollout(T, 'uniform', True, True)
        tr = eng.read_trajectory(T - 1, 1)
        keys.append(tuple(int(tr[k].astype(np.int64).sum()) for k in ('obs', 'reward', 'done')) + (int(eng.get_state()['episode'].sum()),))
assert len(set(keys)) == 1, keys
print('results identical with and without pacing')
