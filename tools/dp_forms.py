#!/usr/bin/env python3
"""Tabular DP as ONE launch: rounds per second of value iteration (V1 + delta + V2 per round, stopping rule off) by grid size in
every one-launch form -- the per-XCD launch (the default wherever the grid fits it), the single workgroup (up to 4096 states), the chip-wide
workgroup cluster (beyond) -- and one launch per round.  Tables compared byte for byte between the forms.
Usage: python tools/dp_forms.py [out.json]"""
import hashlib
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd import _lib  # noqa: E402


def maze(w, h, k):
    random.seed(k)
    np.random.seed(k)
    return gua.GridUniverseEnv(grid_shape=(w, h), random_maze=True)


def main():
    out = {}
    rounds = 2000
    for w in [int(x) for x in sys.argv[1:]] or (8, 32, 40, 64, 101, 128):
        env = maze(w, w, 5)
        S = env.world.size
        row = {}
        ref = None
        for name, path, min_states in (('per_xcd', None, True), ('one_workgroup_or_chip_wide_cluster', 4, False), ('launch_per_round', 1, False)):
            _lib.set_default_option('vi_path', path)
            with gua.Engine(4, gua.GridSpec.from_env(env), seed=1) as eng:
                ts = []
                for rep in range(4):
                    eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
                    eng.sync()
                    t0 = time.perf_counter()
                    eng.vi_sweep(0.99, rounds if name != 'launch_per_round' else 400, greedy_update=True)
                    eng.sync()
                    if rep:
                        ts.append((time.perf_counter() - t0) / (rounds if name != 'launch_per_round' else 400))
                if name != 'launch_per_round':
                    v, pi = eng.vi_get()
                    dg = hashlib.sha256(v.tobytes() + pi.tobytes()).hexdigest()[:12]
                    ref = ref or dg
                    assert dg == ref, (w, name)
            row[name] = round(float(np.median(ts)) * 1e6, 3)
            if name != 'launch_per_round':  # with the stopping rule evaluated every round (value_iteration; a threshold that is never met)
                with gua.Engine(4, gua.GridSpec.from_env(env), seed=1) as eng:
                    ts = []
                    for rep in range(4):
                        eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
                        eng.sync()
                        t0 = time.perf_counter()
                        steps, _ = eng.vi_run(0.99, -1.0, rounds)
                        if rep:
                            ts.append((time.perf_counter() - t0) / rounds)
                    assert steps == rounds
                row[name + '_with_stopping_rule'] = round(float(np.median(ts)) * 1e6, 3)
        out['%dx%d (%d states)' % (w, w, S)] = row
        print(w, S, row, flush=True)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(json.dumps({'us_per_round': out, 'rounds_per_launch': rounds}, indent=1) + '\n')


if __name__ == '__main__':
    main()
