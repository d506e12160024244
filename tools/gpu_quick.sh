cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rows_kernel.py tests/test_gpu_mc.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r05zy_pytest.txt 2>&1; grep -E "passed|failed|Error" gpurun_out/r05zy_pytest.txt | tail -3
cat > /tmp/sample_ab.py <<'PY'
import sys, random, numpy as np
sys.path.insert(0, '.')
import griduniverse_amd as gua
random.seed(123); np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
spec = gua.GridSpec.from_env(env); S = spec.W * spec.H
for N, traj in ((65536, True), (65536, False), (32768, True), (65536, 'packed')):
    eng = gua.Engine(N, spec, seed=5); eng.reset(); eng.reserve_trajectory(1000)
    eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
    for _ in range(250): eng.rollout(1000, 'sample', True, traj)
    ts = []
    for _ in range(5):
        eng.sync(); eng.timer_begin()
        for _ in range(20): eng.rollout(1000, 'sample', True, traj)
        ts.append(eng.timer_end() / 20 * 1e3)
    print('sample %6d envs traj=%s: median %.2f us min %.2f' % (N, traj, float(np.median(ts)), min(ts)), flush=True)
    eng.close()
PY
for rep in 1 2; do
echo "== previous build"; GU_ALLOW_STALE_LIB=1 GU_LIB_PATH=$PWD/griduniverse_amd/lib/libgu_prev2.so python /tmp/sample_ab.py
echo "== this build"; python /tmp/sample_ab.py
done
