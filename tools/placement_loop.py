#!/usr/bin/env python3
"""Does it still matter WHERE the trajectory buffer lands, now that the store pacing is a closed loop?  n engines of the headline
shape, each on the FIRST allocation it gets (option traj_candidates = 1: no placement search), all alive at once so that the
allocations differ; per buffer: the bare store probe (ms per full write, the figure the placement search ranks by), and us per
launch under the closed loop after it has settled (interleaved rounds over the buffers).  A last column: one engine WITH the
default search, for comparison.
    python tools/placement_loop.py [n_buffers] [--json out]"""
import json
import random
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd import _lib  # noqa: E402

n_buf = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
out_path = sys.argv[sys.argv.index('--json') + 1] if '--json' in sys.argv else None
N, T = 65536, 1000
random.seed(123)
np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
spec = gua.GridSpec.from_env(env)


def wall_us(eng, n):
    eng.sync()
    eng.timer_begin()
    for _ in range(n):
        eng.rollout(T, 'uniform', True, True)
    return eng.timer_end() / n * 1e3


engines = []
_lib.set_default_option('traj_candidates', 1)
for b in range(n_buf):
    eng = gua.Engine(N, spec, seed=123)
    eng.reset()
    eng.reserve_trajectory(T)
    engines.append(eng)
_lib.set_default_option('traj_candidates', None)
searched = gua.Engine(N, spec, seed=123)
searched.reset()
searched.reserve_trajectory(T)
engines.append(searched)
probe = [e.probe_trajectory() for e in engines]
for e in engines:
    wall_us(e, 300)  # the loop settles
rounds = []
for r in range(5):
    rounds.append([wall_us(e, 58) for e in engines])
rounds = np.array(rounds)
med = np.median(rounds, axis=0)
print('%-9s %10s %14s %8s' % ('buffer', 'probe ms', 'loop us/launch', 'period'))
rows = []
for b, e in enumerate(engines):
    info = e.rollout_pacing()
    name = 'searched' if b == n_buf else 'first %d' % b
    print('%-9s %10.4f %14.2f %8d   placement %s' % (name, probe[b], med[b], info['period'], e.trajectory_placement()))
    rows.append(dict(buffer=name, probe_ms=probe[b], loop_us=float(med[b]), period=info['period']))
first = med[:n_buf]
print('first allocations: loop us/launch min %.2f median %.2f max %.2f (spread %.1f %%); probe ms min %.4f max %.4f (spread %.1f %%); searched buffer: %.2f us' % (
    first.min(), np.median(first), first.max(), (first.max() / first.min() - 1) * 100, min(probe[:n_buf]), max(probe[:n_buf]),
    (max(probe[:n_buf]) / min(probe[:n_buf]) - 1) * 100, med[n_buf]))
if out_path:
    json.dump(dict(rows=rows), open(out_path, 'w'))
for e in engines:
    e.close()
