#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused rollout kernel on N MI355X (one process per GPU).

    python bench.py                       # 1 GPU, defaults finish in well under a minute
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One bench "step" = ONE launch of the hot path over the whole batch: `gu_rollout` advancing
every env by T env-steps (uniform random actions from the per-env device RNG, harness
auto-reset, int32 (obs, reward, done) trajectory written to HBM).  Workload = BASELINE.json
config 3, the one the metric is quoted on: 65 536 envs per GPU on the 32x32 generator maze
(seed 123).  Weak scaling: every rank owns 65 536 envs, global env ids rank*65536.. (RNG
streams are keyed by global id); the data path has no collective.  Inputs (grid, state) are
resident in HBM before the timed region; nothing returns to the host inside it.

Timing: W untimed launches, barrier + device sync, K timed launches, device sync + barrier;
wall time per rank, MAX over ranks.  The kernel's own duration is measured with HIP events on
the engine's stream (gu_timer_begin/end) over the same K launches -> roofline.achieved.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd import _lib  # noqa: E402

METRIC = 'env-steps/sec at N_envs on 32×32 grid, 1/2/4/8 MI355X; bit-exact vs CPU'
BYTES_PER_ENV_STEP = 12       # fused rollout writing the int32 (obs, reward, done) trajectory
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def build_workload(name):
    """Returns (template env, description).  Grids are built by the product's own host code."""
    if name == 'c3':
        random.seed(123)
        np.random.seed(123)
        env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
        return env, '32x32 generator maze (random.seed(123); np.random.seed(123))'
    if name == 'c4':
        env = gua.GridUniverseEnv(grid_shape=(32, 32), lava_states=[16 + 32 * r for r in range(24)])
        return env, '32x32 open grid, start 0, goal 1023, lava column [16+32r, r<24]'
    if name == 'c2':
        return gua.GridUniverseEnv(grid_shape=(8, 8)), 'default 8x8 grid'
    raise SystemExit('unknown workload ' + name)


def cpu_baseline(template, seed, T, gpu_rows, budget_s=12.0):
    """The ONLY place bench.py touches oracle/: (1) times the per-instance pure-Python restatement of
    the reference's step loop (same operation structure as core/envs/griduniverse_env.py:136-185; the
    reference itself cannot travel to the GPU box) on one host core, (2) times the scalar C oracle,
    and (3) uses the C oracle as the checker for the first envs of the GPU's first timed launch."""
    from oracle import c_oracle as C
    from oracle import gu_rng
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

    grid = C.Grid.from_env(template)
    n_c = 4096
    st = C.State(n_c)
    C.reset(grid, seed, st)
    t0 = time.perf_counter()
    want = C.rollout(grid, seed, st, T, True)
    c_rate = n_c * T / (time.perf_counter() - t0)
    exact = None
    if gpu_rows is not None:
        exact = all(np.array_equal(gpu_rows[k], want[k][:, :gpu_rows[k].shape[1]]) for k in ('obs', 'reward', 'done'))
    return dict(value=py_rate, unit='env-steps/s', cores=1, kind='port',
                sample='%d per-instance Python envs (oracle/ref_env.py) stepped round-robin with reset-on-done for '
                       '%.0f s on one core, same grid and action stream as the GPU run; this port runs at 1.02x the real reference '
                       'step() on a common host (BASELINE.md, tests/golden/calibrate_cpu.py)' % (n_inst, budget_s),
                c_oracle_value=c_rate, c_oracle_sample='%d envs x %d steps, scalar C (oracle/gu_oracle.c), 1 core' % (n_c, T),
                host_cpu_count=os.cpu_count(), host_usable_cores=_usable_cores(), host_cpu_model=_cpu_model()), exact


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
                sample='%d forked processes x %d per-instance Python envs for %.0f s each' % (cores, n_inst, seconds))


def read_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/*.json), or None."""
    path = os.path.join(ROOT, 'profiles', 'rollout_pmc_latest.json')
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--envs', type=int, default=65536, help='envs per GPU')
    ap.add_argument('--T', type=int, default=1000, help='env-steps per launch')
    ap.add_argument('--workload', default='c3', choices=['c2', 'c3', 'c4'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--gather-view', action='store_true', help='after the timed region, exercise the RCCL gathered view')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)' % (args.gpus, world))

    dist = None
    if world > 1:  # torch only as the rendezvous / barrier / max-reduce plumbing (gloo, CPU tensors)
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()

    N, T, K, W = args.envs, args.T, args.steps, args.warmup
    seed = 123
    template, grid_desc = build_workload(args.workload)
    all_cores = None
    if world == 1 and not args.no_cpu_baseline:
        all_cores = cpu_baseline_all_cores(template, seed)  # forks: must precede any HIP call in this process
    n_dev = max(1, _lib.device_count())
    device = local_rank % n_dev  # identity on an N-GPU node; lets a 1-GPU box rehearse the N-process flow
    eng = gua.Engine(N, gua.GridSpec.from_env(template), device=device, env_id0=rank * N, seed=seed)
    eng.reset()
    eng.reserve_trajectory(T)

    first_rows = None
    for i in range(W):
        eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
        if i == 0 and rank == 0 and not args.no_cpu_baseline:
            eng.sync()
            first = eng.read_trajectory(0, T)
            first_rows = {k: v[:, :4096].copy() for k, v in first.items()}
            del first
    eng.sync()
    barrier()
    t0 = time.perf_counter()
    eng.timer_begin()
    for _ in range(K):
        eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
    kernel_ms = eng.timer_end()  # HIP events on the engine's stream; also drains it
    eng.sync()
    elapsed = time.perf_counter() - t0
    barrier()

    if dist is not None:
        import torch
        tmax = torch.tensor([elapsed, kernel_ms], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_max = float(tmax[0]), float(tmax[1])
    else:
        kernel_ms_max = kernel_ms

    if args.gather_view and world >= 1:
        gather_view_demo(eng, dist, rank, world)

    if rank == 0:
        total_steps = float(world) * N * T * K
        launch_s = kernel_ms_max / 1e3 / K
        achieved = BYTES_PER_ENV_STEP * N * T / launch_s / 1e9
        traffic = read_traffic()
        line = {
            'metric': METRIC, 'value': total_steps / elapsed, 'unit': 'env-steps/s', 'n_gpus': world,
            'steps': K, 'warmup': W, 'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'int32', 'data': 'synthetic',
            'config': {'workload': '%s: %d envs per GPU on the %s, uniform random actions from the per-env device RNG, '
                                   'auto-reset on done, one launch = %d env-steps per env, int32 (obs,reward,done) '
                                   'trajectory written to HBM' % (args.workload, N, grid_desc, T),
                       'envs_per_gpu': N, 'env_steps_per_launch': T, 'global_envs': world * N,
                       'parallelism': 'env-index shards, no data-path collective'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS,
                         'traffic': None if traffic is None else traffic.get('hbm_bytes_per_launch'),
                         'kernel': 'gu_rollout_kernel<UNIFORM,TRAJ,LDS>', 'launch_ms': launch_s * 1e3,
                         'algorithmic_bytes_per_launch': BYTES_PER_ENV_STEP * N * T,
                         'traffic_source': None if traffic is None else traffic.get('source')},
        }
        if world == 1 and not args.no_cpu_baseline:
            base, exact = cpu_baseline(template, seed, T, first_rows)
            base['all_cores'] = all_cores
            line['cpu_baseline'] = base
            line['bit_exact_vs_oracle'] = exact
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


def gather_view_demo(eng, dist, rank, world):
    """Optional: the single-array (obs, reward, done) view over RCCL, outside the timed region."""
    if world > 1:
        import torch
        ident = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            ident = torch.frombuffer(bytearray(gua.Engine.comm_unique_id()), dtype=torch.uint8).clone()
        dist.broadcast(ident, src=0)
        uid = bytes(ident.numpy().tobytes())
    else:
        uid = gua.Engine.comm_unique_id()
    eng.comm_init(world, rank, uid)
    t0 = time.perf_counter()
    obs, rew, don = eng.allgather_view()
    dt = time.perf_counter() - t0
    own = eng.read_outputs()
    ok = np.array_equal(obs[rank * eng.N:(rank + 1) * eng.N], own[0])
    print('[rank %d] gathered view of %d envs in %.3f ms, own shard matches: %s' % (rank, obs.size, dt * 1e3, ok),
          file=sys.stderr, flush=True)
    eng.comm_destroy()


if __name__ == '__main__':
    main()
