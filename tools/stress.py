#!/usr/bin/env python3
"""Repetition stress of the paths that synchronise by hand: the workgroup-cluster DP kernel (device-wide barrier per round, bounded
spin), the completion word of small-batch gu_step, and back-to-back rollouts on three kernels sharing one engine.  Every repetition
must reproduce the first result byte for byte.  Usage: python tools/stress.py [seconds per part, default 20]"""
import hashlib
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


# ---- 1. cluster DP: value iteration to a fixed round count on grids of several workgroups
for shape in ((101, 101), (256, 256), (160, 90)):
    random.seed(7)
    np.random.seed(7)
    env = gua.GridUniverseEnv(grid_shape=shape, random_maze=True)
    S = env.world.size
    eng = gua.Engine(64, gua.GridSpec.from_env(env), seed=1)
    first, reps, t0 = None, 0, time.time()
    while time.time() - t0 < budget / 3:
        eng.vi_set(np.zeros(S), np.full((S, 4), 0.25))
        rounds, deltas = eng.vi_run(0.95, 1e-9, 300)
        v, pi = eng.vi_get()
        d = digest(v, pi, deltas)
        first = first or d
        assert d == first and rounds == 300, (shape, reps, d, first, rounds)
        reps += 1
    print('cluster DP %dx%d: %d x 300 rounds, all identical (%s)' % (shape + (reps, first)), flush=True)
    eng.close()

# ---- 2. small-batch gu_step: completion word published by the last workgroup to arrive
random.seed(3)
np.random.seed(3)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
spec = gua.GridSpec.from_env(env)
for N in (1, 64, 1000, 8192):
    eng = gua.Engine(N, spec, seed=5)
    ref = gua.Engine(N, spec, seed=5)
    eng.reset()
    ref.reset()
    rs = np.random.RandomState(N)
    calls, t0 = 0, time.time()
    while time.time() - t0 < budget / 4:
        acts = rs.randint(0, 4, (200, N)).astype(np.int32)
        ref.upload_actions(acts)
        for t in range(200):
            eng.pinned_actions[:] = acts[t]
            o, r, d = eng.step_pinned(auto_reset=True)
            ref.step_device(t, auto_reset=True)
            if t % 50 == 49:
                o2, r2, d2 = ref.read_outputs()
                assert np.array_equal(o, o2) and np.array_equal(r, r2) and np.array_equal(d, d2), (N, calls, t)
        calls += 200
    print('gu_step with completion word, %d envs: %d calls equal to the device-resident path' % (N, calls), flush=True)
    eng.close()
    ref.close()

# ---- 3. one engine, launches alternating between the general, row-table and K-step kernels
eng = gua.Engine(65536, spec, seed=9)
ref = gua.Engine(65536, spec, seed=9)
eng.reset()
ref.reset()
eng.reserve_trajectory(256)
ref.reserve_trajectory(256)
ref.set_option('rollout_rows', 0)   # `ref`: always the general kernel (an option of THAT engine; the library reads no environment)
ref.set_option('rollout_multi', 0)
modes = [dict(trajectory=True), dict(trajectory=False, stats=True), dict(trajectory='packed'), dict(trajectory=False)]
launches, t0 = 0, time.time()
while time.time() - t0 < budget:
    for i, T in enumerate((256, 97, 64, 1, 200, 33)):
        kw = modes[(launches + i) % len(modes)]
        eng.rollout(T, 'uniform', True, **kw)          # default dispatch: whichever kernel is preferred
        ref.rollout(T, 'uniform', True, **kw)          # always the general kernel
    launches += 6
    a, b = eng.get_state(), ref.get_state()
    assert all(np.array_equal(a[k], b[k]) for k in a), launches
print('mixed kernels on one engine: %d launches, state equal to the general kernel after every group of six' % launches, flush=True)
