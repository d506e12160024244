#!/usr/bin/env python3
"""A/B of two single-start auto-reset forms of the rollout kernel (GU_AUTO_FORM=1: reset candidate computed off
the dependent chain; 3: plain selects keyed on the TERM bit), stats-only and trajectory rollouts, two batch sizes.
Result (profiles/archive/r01e_auto_form_ab.txt): form 3 wins everywhere (61.9 vs 64.6-65.4 us stats-only at 65 536 envs) and
is the one the library keeps; the GU_AUTO_FORM switch was removed, so this script is a record of the experiment."""
import os
import random
import statistics
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1:  # child: one form (the env var is read once per process)
    import griduniverse_amd as gua
    random.seed(123)
    np.random.seed(123)
    env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
    for N, T in ((65536, 1000), (1 << 20, 250)):
        eng = gua.Engine(N, gua.GridSpec.from_env(env), seed=1)
        eng.reset()
        eng.reserve_trajectory(T)
        for traj in (False, True):
            times = []
            for rnd in range(5):
                for _ in range(2):
                    eng.rollout(T, 'uniform', True, traj, not traj)
                eng.sync()
                eng.timer_begin()
                for _ in range(10):
                    eng.rollout(T, 'uniform', True, traj, not traj)
                times.append(eng.timer_end() / 10 * 1e3)
            print('form %s N %8d traj %d  median %.1f us  min %.1f' % (sys.argv[1], N, traj, statistics.median(times), min(times)))
        eng.close()
else:
    for rnd in range(2):
        for form in ('1', '3'):
            subprocess.run([sys.executable, __file__, form], env=dict(os.environ, GU_AUTO_FORM=form))
