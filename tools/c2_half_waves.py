import sys
sys.path.insert(0, '.')
import griduniverse_amd as gua
from benchlib.workloads import build_workload
spec = gua.GridSpec.from_env(build_workload('c2')[0])
T, K = 1000, 50
for N in (4096, 8192):
    for half in (0, 1):
        for layout in (-1, 0, 1):
            eng = gua.Engine(N, spec, seed=2)
            eng.set_option('rollout_half_waves', half)
            eng.set_option('traj_layout', layout)
            eng.reset(); eng.reserve_trajectory(T)
            best = 1e9
            for rep in range(3):
                for _ in range(20): eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
                eng.sync(); eng.timer_begin()
                for _ in range(K): eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
                best = min(best, eng.timer_end() / K * 1e3)
            eng.close()
            print('N %d half %d layout %2d: %.1f us' % (N, half, layout, best), flush=True)
