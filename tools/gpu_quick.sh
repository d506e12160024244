# scratch session for gpurun (edited per experiment)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cat > /tmp/stats_ab.py <<'PY'
import sys, random, numpy as np
sys.path.insert(0, '.')
import griduniverse_amd as gua
random.seed(123); np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
spec = gua.GridSpec.from_env(env)
for N in (65536, 262144):
    eng = gua.Engine(N, spec, seed=5); eng.reset()
    for _ in range(20): eng.rollout(1000, 'uniform', True, False, stats=True)
    ts = []
    for _ in range(7):
        eng.sync(); eng.timer_begin()
        for _ in range(40): eng.rollout(1000, 'uniform', True, False, stats=True)
        ts.append(eng.timer_end() / 40 * 1e3)
    print('statistics only %6d envs: median %.2f us min %.2f' % (N, float(np.median(ts)), min(ts)), flush=True)
    eng.close()
PY
for i in 1 2 3; do
echo "== previous"; GU_ALLOW_STALE_LIB=1 GU_LIB_PATH=$PWD/griduniverse_amd/lib/libgu_prev.so python /tmp/stats_ab.py
echo "== this"; python /tmp/stats_ab.py
done
timeout 900 python -m pytest tests/test_gpu_kstep_kernel.py -q -m gpu -x 2>&1 | tail -2
