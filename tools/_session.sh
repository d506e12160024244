for r in 1 2 1 2; do echo "== 32768 envs, int32 rows, rollout_rows $r"; PACE_AB_ROWS=$r PACE_AB_K=30 timeout 600 python tools/pace_ab.py 4 32768 2>&1 | tail -6 | cut -c1-150; done
python - <<'PY'
import sys, random, numpy as np
sys.path.insert(0, '.')
import griduniverse_amd as gua
from griduniverse_amd import _lib
random.seed(123); np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
spec = gua.GridSpec.from_env(env)
for N in (16384, 32768, 65536):
    for rows in (2, 1, 2, 1):
        _lib.set_default_option('rollout_rows', rows)
        eng = gua.Engine(N, spec, seed=123); eng.reset(); eng.reserve_trajectory(1000)
        out = []
        for traj in ('packed', True):
            for _ in range(3): eng.rollout(1000, 'uniform', True, traj)
            eng.sync(); eng.timer_begin()
            for _ in range(30): eng.rollout(1000, 'uniform', True, traj)
            out.append(eng.timer_end() / 30 * 1e3)
        print('%6d envs, rollout_rows %d (%s): packed rows %.1f us, int32 rows %.1f us per launch' % (N, rows, 'pair tables' if rows == 1 else 'one-step table', out[0], out[1]), flush=True)
        eng.close()
PY
