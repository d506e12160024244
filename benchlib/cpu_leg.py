"""The cpu_baseline leg of bench.py: oracle/ timed on the host cores of the GPU box.  The reference is pure
Python and cannot travel; oracle/ref_env.py is its per-instance port (1.02x the real reference step() on a common host,
BASELINE.md / tests/golden/calibrate_cpu.py)."""
import os
import time

def cpu_baseline(template, seed, T, budget_s=12.0):
    """The per-instance pure-Python restatement of the reference's step loop (same operation structure as
    core/envs/griduniverse_env.py:136-185; the reference itself cannot travel to the GPU box) on one host core, plus two
    stronger CPU baselines: the vectorised-numpy restatement and the scalar C oracle."""
    from oracle import c_oracle as C
    from oracle import gu_rng
    from oracle.np_env import NumpyBatchEnv
    from oracle.ref_env import OracleGridUniverseEnv

    n_inst = 64
    envs = []
    for _ in range(n_inst):
        e = OracleGridUniverseEnv(grid_shape=(template.x_max, template.y_max),
                                  initial_state=list(template.starting_states), goal_states=list(template.goal_states),
                                  lava_states=list(template.lava_states), walls=list(template.wall_indices))
        e.reset()
        envs.append(e)
    chunk = 256
    actions = gu_rng.action_stream(seed, range(n_inst), 0, chunk)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for t in range(chunk):
            row = actions[t]
            for j, e in enumerate(envs):
                if e.step(int(row[j]))[2]:
                    e.reset()
        steps += chunk * n_inst
    py_rate = steps / (time.perf_counter() - t0)

    n_np, t_np = 16384, 0
    batch = NumpyBatchEnv.from_env(template, n_np, seed)
    batch.reset()
    acts = gu_rng.action_stream(seed, range(n_np), 0, 64)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        for t in range(64):
            batch.step(acts[t], auto_reset=True)
        t_np += 64
    np_rate = n_np * t_np / (time.perf_counter() - t0)

    grid = C.Grid.from_env(template)
    n_c = 4096
    st = C.State(n_c)
    C.reset(grid, seed, st)
    t0 = time.perf_counter()
    C.rollout(grid, seed, st, T, True, trajectory=False)
    c_rate = n_c * T / (time.perf_counter() - t0)
    return dict(value=py_rate, unit='env-steps/s', cores=1, kind='port',
                sample='%d oracle/ref_env.py envs, round-robin, reset on done, %.0f s, same grid and action stream' % (n_inst, budget_s),
                numpy_vectorised_value=np_rate, numpy_vectorised_sample='%d envs as arrays (oracle/np_env.py), 3 s, 1 core' % n_np,
                c_oracle_value=c_rate, c_oracle_sample='%d envs x %d steps (oracle/gu_oracle.c), 1 core' % (n_c, T),
                host_cpu_count=os.cpu_count(), host_usable_cores=_usable_cores(), host_cpu_model=_cpu_model())


def _usable_cores():
    """Cores this process can actually run on: CPU affinity, capped by the cgroup CPU quota if there is one."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]  # cgroup v2
        if quota != 'max':
            cores = min(cores, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0:
                cores = min(cores, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return cores


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline_all_cores(template, seed, seconds=4.0):
    """Part of the cpu_baseline leg, run BEFORE the GPU is initialised (it forks): the same per-instance Python
    port (oracle/ref_env.py) on every core this process may use, one forked process per core, aggregate rate."""
    import multiprocessing as mp

    from oracle import gu_rng
    from oracle.ref_env import OracleGridUniverseEnv

    cores = _usable_cores()
    n_inst, chunk = 16, 256

    def worker(index, conn):
        envs = [OracleGridUniverseEnv(grid_shape=(template.x_max, template.y_max), initial_state=list(template.starting_states),
                                      goal_states=list(template.goal_states), lava_states=list(template.lava_states),
                                      walls=list(template.wall_indices)) for _ in range(n_inst)]
        for e in envs:
            e.reset()
        actions = gu_rng.action_stream(seed, range(index * n_inst, (index + 1) * n_inst), 0, chunk)
        steps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for t in range(chunk):
                row = actions[t]
                for j, e in enumerate(envs):
                    if e.step(int(row[j]))[2]:
                        e.reset()
            steps += chunk * n_inst
        conn.send((steps, time.perf_counter() - t0))
        conn.close()

    ctx = mp.get_context('fork')
    procs = []
    for i in range(cores):
        parent, child = ctx.Pipe(duplex=False)
        p = ctx.Process(target=worker, args=(i, child))
        p.start()
        procs.append((p, parent))
    rate = 0.0
    for p, parent in procs:
        steps, dt = parent.recv()
        rate += steps / dt
        p.join()
    return dict(value=rate, unit='env-steps/s', cores=cores, kind='port',
                sample='%d forked processes x %d oracle/ref_env.py envs, %.0f s each' % (cores, n_inst, seconds))

