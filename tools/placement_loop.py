import os, sys, time, statistics, random
import numpy as np
sys.path.insert(0, os.getcwd())
import griduniverse_amd as gua
random.seed(123); np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
T, N = 1000, 65536
for far in ('0', '32'):
    os.environ['GU_TRAJ_FAR_CANDIDATES'] = far
    for i in range(4):
        eng = gua.Engine(N, gua.GridSpec.from_env(env), seed=1); eng.reset()
        t0 = time.perf_counter(); eng.reserve_trajectory(T); dt = time.perf_counter() - t0
        ts = []
        for r in range(3):
            for _ in range(3): eng.rollout(T, 'uniform', True, True)
            eng.sync(); eng.timer_begin()
            for _ in range(20): eng.rollout(T, 'uniform', True, True)
            ts.append(eng.timer_end() / 20 * 1e3)
        print('far', far, 'engine', i, 'placement', eng.trajectory_placement(), 'reserve %.1f ms' % (dt * 1e3), 'launch %.1f us' % statistics.median(ts), flush=True)
        # keep the engine alive so the next one sees a different neighbourhood
        globals().setdefault('keep', []).append(eng)
