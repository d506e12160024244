#!/usr/bin/env python3
"""Where the closed loop of the store pacing sits against the best FIXED period of the same engine and buffer (config 3, 65 536 envs x
1000 steps): the loop for --loop launches, then fixed periods (each --each launches, the list run up and then down), then the loop again.
    python tools/pace_aim.py [--loop 1500] [--each 240] [--kind c3]"""
import argparse
import random
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import griduniverse_amd as gua  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--loop', type=int, default=1500)
ap.add_argument('--each', type=int, default=240)
ap.add_argument('--kind', default='c3')
ap.add_argument('--periods', type=int, nargs='*', default=None)
args = ap.parse_args()
T = 1000
random.seed(123)
np.random.seed(123)
if args.kind == 'c4':
    env, N = gua.GridUniverseEnv(grid_shape=(32, 32), lava_states=[16 + 32 * r for r in range(24)], goal_states=[1023]), 32768
else:
    env, N = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True), 65536
policy = 'sample' if args.kind == 'sample' else 'uniform'
spec = gua.GridSpec.from_env(env)
eng = gua.Engine(N, spec, seed=123)
eng.reset()
eng.reserve_trajectory(T)
if policy == 'sample':
    S = spec.W * spec.H
    eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))


def run(n):
    """n launches back to back in chunks of 60: wall us per launch of every chunk"""
    out = []
    done = 0
    while done < n:
        k = min(60, n - done)
        eng.sync()
        eng.timer_begin()
        for _ in range(k):
            eng.rollout(T, policy, True, True)
        out.append(eng.timer_end() / k * 1e3)
        done += k
    return out


def loop_phase(tag):
    eng.set_option('rollout_pace', None)
    chunks = run(args.loop)
    lg = eng.rollout_pace_log(policy, True)
    per = lg['period'][lg['period'] > 0]
    share = float(np.sum(lg['ended_late'])) / max(1.0, float(np.sum(lg['waves'])))
    print('%s: wall us per launch, chunks of 60: first third %.2f  middle %.2f  last third %.2f  (min %.2f max %.2f) | last 61 launches: period %.1f .. %.1f, waves behind %.3f, aim dec_q %s' % (
        tag, np.mean(chunks[:len(chunks) // 3]), np.mean(chunks[len(chunks) // 3:2 * len(chunks) // 3]), np.mean(chunks[2 * len(chunks) // 3:]), min(chunks), max(chunks),
        per.min(), per.max(), share, sorted(set(int(x) for x in lg['dec_q']))), flush=True)
    return chunks


print('placement', eng.trajectory_placement())
loop_phase('loop (fresh engine)')
model = eng.rollout_pacing(policy, True)['period']
periods = args.periods or list(range(model - 8, model + 10, 2))
res = {p: [] for p in periods}
for order in (periods, periods[::-1]):
    for p in order:
        eng.set_option('rollout_pace', p)
        run(20)
        res[p] += run(args.each)
print('fixed periods (ticks per 16 steps -> wall us per launch, mean of %d launches, two passes):' % (2 * args.each))
for p in periods:
    print('   %4d: %.2f   (chunks %s)' % (p, np.mean(res[p]), ' '.join('%.1f' % x for x in res[p])))
best = min(periods, key=lambda p: np.mean(res[p]))
print('best fixed period %d: %.2f us' % (best, np.mean(res[best])))
loop_phase('loop again')
eng.set_option('rollout_pace', 0)
print('no limiter: %.2f us' % np.mean(run(240)))
eng.close()
