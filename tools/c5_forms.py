#!/usr/bin/env python3
"""Config 5 (65 536 envs, 64x64 maze, one V1+V2 sweep fused with one greedy env step per round) in every form of
gu_vi_sweep_step_run: one launch synchronised per XCD (workgroups of 256 / 512 / 1024 threads), one launch with a chip-wide
barrier per round, one launch per round.  us per round of each, interleaved over several repeats; every form's results are
compared byte for byte with the first one's, and the per-XCD form is re-run `--stress` times against its own first result.
Usage: python tools/c5_forms.py [--envs 65536] [--rounds 2000] [--repeats 5] [--stress 30] [out.json]"""
import argparse
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

FORMS = {  # name -> (vi_path, vi_xcd_block)
    'xcd_256': (None, 256), 'xcd_512': (None, 512), 'xcd_1024': (None, 1024), 'chip_wide': (4, None), 'per_launch': (1, None)}


def maze(w, h, k):
    random.seed(k)
    np.random.seed(k)
    return gua.GridUniverseEnv(grid_shape=(w, h), random_maze=True)


def digest(eng):
    v, pi = eng.vi_get()
    st = eng.get_state()
    h = hashlib.sha256()
    for x in (v, pi, st['pos'], st['done'], st['episode'], eng.read_outputs()[1]):
        h.update(np.ascontiguousarray(x).tobytes())
    return h.hexdigest()[:16]


def run(eng, S, rounds, gamma=1.0):
    eng.seed(5)
    eng.reset()
    eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
    eng.sync()
    t0 = time.perf_counter()
    deltas = eng.vi_sweep_step_run(gamma, rounds, True)
    dt = time.perf_counter() - t0
    return dt, hashlib.sha256(deltas.tobytes()).hexdigest()[:8] + digest(eng), eng.vi_last_form()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=2000)
    ap.add_argument('--repeats', type=int, default=5)
    ap.add_argument('--stress', type=int, default=30)
    ap.add_argument('out', nargs='?')
    a = ap.parse_args()
    env = maze(a.size, a.size, 5)
    S = env.world.size
    out = {'workload': '%d envs, %dx%d generator maze seed 5, gamma 1, %d rounds per launch' % (a.envs, a.size, a.size, a.rounds), 'forms': {}}
    eng = gua.Engine(a.envs, gua.GridSpec.from_env(env), seed=5)
    times = {k: [] for k in FORMS}
    ref = None
    for rep in range(a.repeats + 1):  # (the first pass warms every form up)
        for name, (path, block) in FORMS.items():
            _lib.set_default_option('vi_path', path)
            _lib.set_default_option('vi_xcd_block', block)
            rounds = a.rounds if name != 'per_launch' else min(a.rounds, 500)
            dt, dg, form = run(eng, S, rounds)
            if name != 'per_launch':
                ref = ref or dg
                assert dg == ref, (name, dg, ref)
            if rep:
                times[name].append(dt / rounds * 1e6)
            out['forms'].setdefault(name, {})['form_taken'] = form
            if form == 1:
                out['forms'][name]['workgroups_per_xcc'] = eng.vi_last_clusters()
    for name, ts in times.items():
        ts.sort()
        out['forms'][name].update(us_per_round_median=ts[len(ts) // 2], us_per_round_min=ts[0], us_per_round_max=ts[-1],
                                  env_steps_per_s=a.envs / (ts[len(ts) // 2] * 1e-6), state_updates_per_s=S / (ts[len(ts) // 2] * 1e-6))
    # launch overhead of the per-XCD form: the same launch with few rounds
    _lib.set_default_option('vi_path', None)
    _lib.set_default_option('vi_xcd_block', None)
    # (the default dispatch gives calls of one or two rounds to one launch per round; vi_path = 6: the per-XCD form always)
    for label, path in (('default_us_per_call_by_rounds', None), ('xcd_always_us_per_call_by_rounds', 6), ('launch_per_round_us_per_call_by_rounds', 1)):
        _lib.set_default_option('vi_path', path)
        short = {}
        for rounds in (1, 2, 3, 4, 10, 100):
            ts = sorted(run(eng, S, rounds)[0] for _ in range(5))
            short[rounds] = ts[2] * 1e6
        out[label] = short
    _lib.set_default_option('vi_path', None)
    first = None
    bad = 0
    for i in range(a.stress):
        dg = run(eng, S, 333)[1]
        first = first or dg
        bad += dg != first
    out['stress'] = {'runs': a.stress, 'rounds_each': 333, 'differing_from_first': bad}
    eng.close()
    text = json.dumps(out, indent=1)
    print(text)
    if a.out:
        open(a.out, 'w').write(text + '\n')
    assert bad == 0


if __name__ == '__main__':
    main()
