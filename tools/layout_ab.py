#!/usr/bin/env python3
"""The int32 trajectory as three planes [T][N] (three 4-byte stores per lane and step) against one plane of triples [T][N][3]
(one 12-byte store; option traj_layout = 1), interleaved in one process, same seeds -- the rows that come back must be equal.
Workloads: config 2 (4096 envs, 8x8), a config-4 shard (32 768 envs, lava), config 3 (65 536 envs, maze) under the uniform
policy, config 3 under a sampled and a greedy policy.  Also rollout_rows = 3: the pair tables for int32 triples.
    python tools/layout_ab.py [--reps 5] [--launches 20] [--json out]"""
import argparse
import json
import random
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import griduniverse_amd as gua  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--launches', type=int, default=20)
ap.add_argument('--json', default=None)
ap.add_argument('--only', default=None)
ap.add_argument('--half', action='store_true')
ap.add_argument('--sizes', type=int, nargs='*', default=None, help='batch sizes for a crossover table on the config-2 and config-4 grids')
args = ap.parse_args()
T = 1000


def workload(name):
    random.seed(123)
    np.random.seed(123)
    if name == 'c2':
        return gua.GridUniverseEnv(grid_shape=(8, 8)), 4096
    if name == 'c4':
        return gua.GridUniverseEnv(grid_shape=(32, 32), lava_states=[16 + 32 * r for r in range(24)], goal_states=[1023]), 32768
    return gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True), 65536


if args.sizes:
    pass
cases = [('c2', 'uniform'), ('c4', 'uniform'), ('c3', 'uniform'), ('c3', 'sample'), ('c3', 'greedy'), ('c3', 'stream')]
variants = [('planes', dict(traj_layout=0, rollout_half_waves=0)), ('triples', dict(traj_layout=1, rollout_half_waves=0)),
            ('triples+pairs', dict(traj_layout=1, rollout_rows=3, rollout_half_waves=0)),
            ('planes general', dict(traj_layout=0, rollout_rows=0)), ('triples general', dict(traj_layout=1, rollout_rows=0))]
if args.half:  # the transition-row kernel with 32 envs per wave and twice the waves (option rollout_half_waves)
    variants = [('planes', dict(traj_layout=0, rollout_half_waves=0)), ('planes/half', dict(traj_layout=0, rollout_half_waves=1)),
                ('triples/half', dict(traj_layout=1, rollout_half_waves=1)), ('tri+pairs', dict(traj_layout=1, rollout_rows=3, rollout_half_waves=0)),
                ('tri+pairs/half', dict(traj_layout=1, rollout_rows=3, rollout_half_waves=1)), ('default', dict())]
out = {}
if args.sizes:
    cases = [(w + '@%d' % n, 'uniform') for w in ('c2', 'c4') for n in args.sizes]
for wname, policy in cases:
    if args.only and args.only != wname + ':' + policy:
        continue
    env, N = workload(wname.split('@')[0])
    if '@' in wname:
        N = int(wname.split('@')[1])
    spec = gua.GridSpec.from_env(env)
    S = spec.W * spec.H
    engines = {}
    for vname, opts in variants:
        eng = gua.Engine(N, spec, seed=5)
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.reset()
        eng.reserve_trajectory(T)
        if policy in ('sample', 'greedy'):
            eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
        if policy == 'stream':
            eng.upload_actions(np.random.RandomState(2).randint(0, 4, (T, N)).astype(np.int32))
        engines[vname] = eng
    # same rows from every variant
    ref = None
    for vname, eng in engines.items():
        eng.rollout(T, policy, True, True)
        tr = eng.read_trajectory(0, T)
        key = tuple(tr[k] for k in ('obs', 'reward', 'done'))
        if ref is None:
            ref = key
        else:
            assert all(np.array_equal(a, b) for a, b in zip(ref, key)), (wname, policy, vname)
    del ref, key, tr
    times = {v: [] for v, _ in variants}
    for rep in range(args.reps):
        for vname, eng in engines.items():
            for _ in range(3):
                eng.rollout(T, policy, True, True)
            eng.sync()
            eng.timer_begin()
            for _ in range(args.launches):
                eng.rollout(T, policy, True, True)
            times[vname].append(eng.timer_end() / args.launches * 1e3)
    line = {v: round(float(np.median(t)), 2) for v, t in times.items()}
    print('%-3s %-8s %6d envs: ' % (wname, policy, N) + '  '.join('%s %.2f us' % (v, line[v]) for v, _ in variants) +
          '   (min: ' + ' '.join('%.2f' % min(times[v]) for v, _ in variants) + ')', flush=True)
    out[wname + ':' + policy] = dict(median_us=line, all_us={v: [round(x, 2) for x in t] for v, t in times.items()})
    for eng in engines.values():
        eng.close()
print('rows identical in every variant')
if args.json:
    with open(args.json, 'w') as f:
        json.dump(out, f)
