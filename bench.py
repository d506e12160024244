#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused rollout kernel on N MI355X (one process per GPU).

    python bench.py                       # 1 GPU, defaults finish in about a minute
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W      # one rank per GPU (what the driver runs)
    python bench.py --gpus N              # started plainly: spawns the N ranks itself (before anything touches a GPU)
    python bench.py --gpus N --single-process   # ONE process, one engine per device, launches issued device after device

The ranks talk to each other on the host only through griduniverse_amd/rendezvous.py (a socket per rank: barrier, max-reduce,
RCCL's 128-byte id) -- no PyTorch is imported anywhere, whichever way the script is started.

One bench "step" = ONE launch of the hot path over the whole batch: `gu_rollout` advancing every env by T env-steps
(uniform random actions from the per-env device RNG, harness auto-reset, int32 (obs, reward, done) trajectory written to
HBM).  Workload = BASELINE.json config 3, the one the metric is quoted on: 65 536 envs per GPU on the 32x32 generator maze
(seed 123).  Weak scaling: every rank owns 65 536 envs, global env ids rank*65536.. (RNG streams are keyed by global id);
the data path has no collective.  Inputs (grid, state) are resident in HBM before the timed region; nothing returns to
the host inside it.

Timing.  W untimed launches, then BLOCKS of exactly K launches, each block bracketed by barrier + device sync on both
sides, repeated until at least --min-seconds (0.5 s) of timed launches have run.  Per block: wall time (MAX over ranks)
and the kernels' own duration from HIP events on the engine's stream (gu_timer_begin/end).  `ms_per_step` / `value` are
the MEDIAN block; the spread is reported next to them.  roofline.achieved = algorithmic bytes per launch / (median block's
HIP-event time / K).

Checks in the same run (rank 0): the FIRST launch is hashed in full and compared with the sha256 the REFERENCE produced
for exactly this run (tests/golden/digests.json, 65.5 M steps of the reference's own step()); after the timed region the
final (pos, done, episode, step count) of a sample of envs is compared with the C oracle advanced by every step launched.

N > 1: after the timed region the RCCL gathered view runs once and is checked against every rank's own shard ("rccl"),
and config 4 -- 262 144 envs on the lava grid in total, split over the ranks -- is timed as a strong-scaling line
("strong_c4"), its result checked against the C oracle on every rank (and, on one GPU, against the reference digest).
"""
import argparse
import contextlib
import ctypes
import glob
import hashlib
import json
import os
import random
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

# RCCL between processes needs dmabuf IPC on this driver stack (the pool exports this already; harmless when it is set)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd import _lib  # noqa: E402

METRIC = 'env-steps/sec at N_envs on 32×32 grid, 1/2/4/8 MI355X; bit-exact vs CPU'
BYTES_PER_ENV_STEP = 12       # fused rollout writing the int32 (obs, reward, done) trajectory
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBPS = 6290.0
C4_TOTAL_ENVS = 262144        # BASELINE.json config 4
WORKLOAD_SEED = {'c2': 2, 'c3': 123, 'c4': 4, 'c5': 5}
REFERENCE_DIGEST = {'c2': 'c2_open8x8_4096x1000', 'c3': 'c3_maze32_65536x1000', 'c4': 'c4_lava32_262144x250'}


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
    if name == 'c5':
        random.seed(5)
        np.random.seed(5)
        return gua.GridUniverseEnv(grid_shape=(64, 64), random_maze=True), '64x64 generator maze (random.seed(5); np.random.seed(5))'
    raise SystemExit('unknown workload ' + name)


# --------------------------------------------------------------------------------------- CPU baseline leg (oracle/)
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
                sample='%d per-instance Python envs (oracle/ref_env.py) stepped round-robin with reset-on-done for '
                       '%.0f s on one core, same grid and action stream as the GPU run; this port runs at 1.02x the real reference '
                       'step() on a common host (BASELINE.md, tests/golden/calibrate_cpu.py)' % (n_inst, budget_s),
                numpy_vectorised_value=np_rate,
                numpy_vectorised_sample='%d envs stepped as arrays (oracle/np_env.py: the reference\'s tests on numpy arrays, no '
                                        'precomputed table) for 3 s, 1 core' % n_np,
                c_oracle_value=c_rate, c_oracle_sample='%d envs x %d steps, scalar C (oracle/gu_oracle.c), 1 core' % (n_c, T),
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
                sample='%d forked processes x %d per-instance Python envs for %.0f s each' % (cores, n_inst, seconds))


# --------------------------------------------------------------------------------------- checks
# (the cpu_baseline_check_* functions use oracle/ as the CHECKER of what the GPU produced -- never as the thing measured)
def sha256_triplet(traj):
    """sha256 over obs | reward | done, each int32 little-endian [T, N] -- tests/golden/make_golden.py: digest()."""
    h = hashlib.sha256()
    for k in ('obs', 'reward', 'done'):
        h.update(np.ascontiguousarray(traj[k], dtype='<i4').tobytes())
    return h.hexdigest()


def reference_digest(workload, template, seed, N, T, env_id0):
    """The sha256 the REFERENCE's own step() produced for this very run, if this run is the one that was captured
    (tests/golden/digests.json: same grid, seed, batch, length, env ids 0..N-1, from reset, auto-reset)."""
    try:
        entry = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'digests.json')))[REFERENCE_DIGEST[workload]]
    except (OSError, KeyError, ValueError):
        return None
    same = (env_id0 == 0 and entry['N'] == N and entry['T'] == T and entry['seed'] == seed and entry['auto_reset']
            and (entry['W'], entry['H']) == (template.x_max, template.y_max)
            and entry['starts'] == [int(s) for s in template.starting_states]
            and entry['goals'] == [int(s) for s in template.goal_states]
            and entry['lava'] == [int(s) for s in template.lava_states]
            and entry['walls'] == [int(s) for s in template.wall_indices])
    return entry['sha256'] if same else None


def cpu_baseline_check_prefix(template, seed, env_id0, traj, n_check=4096):
    """First `n_check` envs of a from-reset launch against the C oracle (used when no reference digest covers the run)."""
    from oracle import c_oracle as C
    T, N = traj['obs'].shape
    n = min(n_check, N)
    grid = C.Grid.from_env(template)
    st = C.State(n, env_id0)
    C.reset(grid, seed, st)
    want = C.rollout(grid, seed, st, T, True)
    return all(np.array_equal(traj[k][:, :n], want[k]) for k in ('obs', 'reward', 'done'))


def cpu_baseline_check_final_state(template, seed, env_id0, N, total_steps, state, budget_steps=4.0e8):
    """After ALL launches of the run (checked one, warm-up, probe, timed, instrumented): the final pos / done / episode /
    step count of a sample of envs -- the first and the last ones of the shard -- against the C oracle advanced by the same
    number of steps.  The whole batch would take the scalar oracle about an hour; the sample is sized to seconds."""
    from oracle import c_oracle as C
    per_block = int(max(1, min(N // 2, budget_steps // max(1, total_steps) // 2)))
    grid = C.Grid.from_env(template)
    ok, checked = True, 0
    for lo in sorted({0, N - per_block}):
        st = C.State(per_block, env_id0 + lo)
        C.reset(grid, seed, st)
        C.rollout(grid, seed, st, total_steps, True, trajectory=False)
        sl = slice(lo, lo + per_block)
        ok = ok and all(np.array_equal(state[k][sl], getattr(st, k)) for k in ('pos', 'done', 'episode', 'tcount'))
        checked += per_block
    return dict(equal=bool(ok), envs_checked=checked, env_steps_each=int(total_steps),
                fields='pos, done, episode, tcount', checker='oracle/gu_oracle.c')


def cpu_baseline_check_stats(template, seed, env_id0, T, ret, episodes, n_check=2048):
    """Per-env return and episode count of a from-reset, statistics-only launch against the C oracle (first `n_check` envs)."""
    from oracle import c_oracle as C
    n = min(n_check, ret.size)
    grid = C.Grid.from_env(template)
    st = C.State(n, env_id0)
    C.reset(grid, seed, st)
    want = C.rollout(grid, seed, st, T, True, trajectory=False, stats=True)
    return bool(np.array_equal(ret[:n], want['ret']) and np.array_equal(episodes[:n], want['episodes']))


def cpu_baseline_check_c5(template, seed, gamma, rounds, v, pi, state, rewards):
    """Config 5 against the C oracle: `rounds` x { V1 + V2 sweep (value_iteration_step, itself pinned to the reference's
    value-iteration trace by tests/test_oracle_c.py); every env steps greedily on the updated policy (np.argmax of its row,
    examples/griduniverse_alg_examples.py:76), lazy reset first } from reset with zero values and the uniform policy --
    tables as raw bytes, every env's position / done flag / episode count and last reward."""
    from oracle import c_oracle as C
    grid = C.Grid.from_env(template)
    S, N = template.world.size, state['pos'].size
    st = C.State(N)
    C.reset(grid, seed, st)
    v_o, pi_o = np.zeros(S), np.ones((S, 4)) / 4
    want = None
    for _ in range(rounds):
        v_o, pi_o, _ = C.value_iteration_step(grid, gamma, pi_o, v_o)
        acts = np.argmax(pi_o, axis=1).astype(np.int32)
        if st.done.any():
            C.reset(grid, seed, st, mask=st.done.astype(bool))
        want = C.rollout(grid, seed, st, 1, False, actions=acts[st.pos][None, :])
    return bool(v.tobytes() == v_o.tobytes() and pi.tobytes() == pi_o.tobytes() and np.array_equal(state['pos'], st.pos)
                and np.array_equal(state['done'], st.done) and np.array_equal(state['episode'], st.episode)
                and np.array_equal(rewards, want['reward'][0]))


# --------------------------------------------------------------------------------------- the other BASELINE configs
def baseline_configs(engine_cls, device, K, check):
    """BASELINE.json configs 2, 4 (one shard of eight) and 5 on this GPU, beside the headline (config 3) -- never as `value`.
    Each entry: workload, us per launch (or per round), env-steps/s, the bound it claims with a stated floor, its own parity bit."""
    out = {}
    T = 1000
    clock_ghz = None
    try:
        clock_ghz = float(engine_cls.device_info(device).get('sclk_khz', 0)) / 1e6 or None
    except Exception:  # noqa: BLE001 -- reporting only
        pass

    def rows(eng, policy='uniform'):
        for _ in range(settle_launches(eng, T, policy, trajectory=True)):
            eng.rollout(T, policy, auto_reset=True, trajectory=True)
        eng.sync()
        eng.timer_begin()
        for _ in range(K):
            eng.rollout(T, policy, auto_reset=True, trajectory=True)
        return eng.timer_end() / K

    # ---- config 2: 4096 envs, default 8x8 grid
    template, desc = build_workload('c2')
    N, seed = 4096, WORKLOAD_SEED['c2']
    eng = engine_cls(N, gua.GridSpec.from_env(template), device=device, env_id0=0, seed=seed)
    try:
        eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
        ok = None
        if check:
            got = eng.read_trajectory(0, T)
            ref = reference_digest('c2', template, seed, N, T, 0)
            ok = sha256_triplet(got) == ref if ref is not None else bool(cpu_baseline_check_prefix(template, seed, 0, got))
            del got
        ms = rows(eng)
    finally:
        eng.close()
    cyc = None if clock_ghz is None else ms * 1e3 * clock_ghz * 1e3 / T
    out['c2'] = dict(workload='c2: %d envs on the %s, seed %d, uniform device-RNG actions, auto-reset, int32 trajectory, %d env-steps per launch' % (N, desc, seed, T),
                     us_per_launch=ms * 1e3, env_steps_per_s=float(N) * T / ms * 1e3, hbm_gbps=BYTES_PER_ENV_STEP * N * T / ms / 1e6,
                     frac_of_hbm_peak=BYTES_PER_ENV_STEP * N * T / ms / 1e6 / HBM_PEAK_GBPS,
                     bound='latency / issue of one wave: 64 waves on 1024 SIMDs, each a chain of %d dependent steps.  Since round 5 the rows of a '
                           'batch this small are one plane of (obs, reward, done) triples and the steps go through the pair tables: one LDS round '
                           'trip per TWO steps and two 12-byte-per-lane stores per pair where round 4 issued six 4-byte ones (three 256-byte '
                           'stores per step at ~25 clocks each: 49 us per launch then) -- gu_rollout_rows_kernel<UNIFORM, triples, pairs>; the '
                           'HBM stream is 49 MB per launch' % T,
                     cycles_per_step=cyc, floor_us='%d pairs x ~85 shader clocks (ds_read_b64 issue -> use, MI355X_MICROARCH.md) = %.0f us at the clock '
                                                   'HIP reports, + ~5 us of table staging and the first step' % (T // 2, T // 2 * 85 / ((clock_ghz or 2.4) * 1e3)),
                     bit_exact=ok, check='first launch from reset == the reference digest c2_open8x8_4096x1000 (tests/golden/digests.json)')

    # ---- config 4, one shard of eight: 32 768 envs with global ids 32768 .. 65535 on the lava grid
    template, desc = build_workload('c4')
    N, seed = C4_TOTAL_ENVS // 8, WORKLOAD_SEED['c4']
    eng = engine_cls(N, gua.GridSpec.from_env(template), device=device, env_id0=N, seed=seed)
    try:
        eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(250, 'uniform', auto_reset=True, trajectory=True)
        ok = bool(cpu_baseline_check_prefix(template, seed, N, eng.read_trajectory(0, 250), n_check=N)) if check else None
        ms = rows(eng)
    finally:
        eng.close()
    out['c4_shard'] = dict(workload='c4, shard 1 of 8: %d envs (global ids %d ..) on the %s, seed %d, uniform device-RNG actions, auto-reset, int32 '
                                    'trajectory, %d env-steps per launch' % (N, N, desc, seed, T),
                           us_per_launch=ms * 1e3, env_steps_per_s=float(N) * T / ms * 1e3, hbm_gbps=BYTES_PER_ENV_STEP * N * T / ms / 1e6,
                           frac_of_hbm_peak=BYTES_PER_ENV_STEP * N * T / ms / 1e6 / HBM_PEAK_GBPS,
                           bound='between the dependent chain (512 waves: half a wave per SIMD) and the HBM write stream (393 MB per launch)',
                           floor_us='393 MB / 8 TB/s = 49 us (HBM); %d steps x ~85 clocks = %.0f us (chain)' % (T, T * 85 / ((clock_ghz or 2.4) * 1e3)),
                           eight_shards_env_steps_per_s_if_scaling_were_perfect=8.0 * N * T / ms * 1e3,
                           bit_exact=ok, check='first launch (250 steps from reset): the whole shard == C oracle at its global env ids')

    # ---- config 5: 65 536 envs, 64x64 maze, one V1 + V2 sweep fused with one greedy env step per round
    if hasattr(engine_cls, 'vi_sweep_step_run'):
        template, desc = build_workload('c5')
        N, seed, gamma, S = 65536, WORKLOAD_SEED['c5'], 1.0, 64 * 64
        eng = engine_cls(N, gua.GridSpec.from_env(template), device=device, env_id0=0, seed=seed)
        try:
            ok, n_check = None, 12
            if check:
                eng.reset()
                eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
                eng.vi_sweep_step_run(gamma, n_check, auto_reset=True)
                v, pi = eng.vi_get()
                ok = cpu_baseline_check_c5(template, seed, gamma, n_check, v, pi, eng.get_state(), eng.read_outputs()[1])
            rounds, per_round = 2000, []
            for rep in range(4):
                eng.reset()
                eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
                eng.sync()
                t0 = time.perf_counter()
                eng.vi_sweep_step_run(gamma, rounds, auto_reset=True)
                if rep:
                    per_round.append((time.perf_counter() - t0) / rounds)
            form = eng.vi_last_form()
        finally:
            eng.close()
        us = float(np.median(per_round)) * 1e6
        out['c5'] = dict(workload='c5: %d envs on the %s, gamma %.1f: per round one V1 + V2 sweep of the %d-state tables (float64, bit-exact) fused with one '
                                  'greedy env step of every env; %d rounds in ONE launch' % (N, desc, gamma, S, rounds),
                         us_per_round=us, env_steps_per_s=N / us * 1e6, state_updates_per_s=S / us * 1e6,
                         form={1: 'one launch synchronised per XCD (self-tagged granules, no barrier between workgroups)',
                               2: 'one launch, chip-wide barrier per round', 3: 'one launch per round'}.get(form, str(form)),
                         timing='host wall time of one gu_vi_sweep_step_run call / rounds (snapshot, launch and read-back of the deltas included), median of 3',
                         bound='latency: per round one store -> L2 -> load hop inside the XCD and two dependent float64 chains (V1, V2: ~35-clock '
                               'dependent-issue latency per float64 operation on gfx950); the round moves 64 KB of value granules per XCD, nothing '
                               'near any bandwidth limit',
                         floor_us='V1 (6 dependent float64 operations) + V2 (8) at ~38 clocks each + one L2 store-to-load hop (~600 clocks) + two workgroup '
                                  'barriers = ~1300 clocks = 0.55 us at 2.4 GHz',
                         bit_exact=ok, check='%d rounds from reset (zero values, uniform policy): tables as raw bytes and every env\'s position / done / '
                                             'episode / reward == C oracle (value_iteration_step + greedy step)' % n_check)
    return out


def topology_block(engine_cls):
    """What the node looks like, for the first run on more than one GPU to be self-diagnosing: HIP's device count, every
    device's PCI id, the xGMI link matrix as sysfs (or rocm-smi) shows it, the RCCL library the gathered view would load."""
    out = {}
    try:
        n = _lib.device_count() if engine_cls is gua.Engine else 1
        out['hip_device_count'] = n
        out['devices'] = []
        for d in range(n):
            info = engine_cls.device_info(d) if hasattr(engine_cls, 'device_info') else {}
            out['devices'].append({k: info.get(k) for k in ('name', 'arch', 'pci', 'cus') if k in info})
    except Exception as err:  # noqa: BLE001 -- reporting only
        out['error'] = str(err)
    links = {}
    for path in sorted(glob.glob('/sys/class/kfd/kfd/topology/nodes/*/io_links/*/properties')):
        try:
            props = dict(line.split(None, 1) for line in open(path).read().splitlines() if ' ' in line)
        except OSError:
            continue
        if props.get('type', '').strip() == '11':  # HSA_IOLINK_TYPE_XGMI
            node = path.split('/nodes/')[1].split('/')[0]
            links.setdefault(node, []).append(dict(to=props.get('node_to', '').strip(), weight=props.get('weight', '').strip(),
                                                   max_bandwidth=props.get('max_bandwidth', '').strip()))
    out['xgmi_links_by_kfd_node'] = links or None
    out['xgmi_hives'] = sorted({open(p).read().strip() for p in glob.glob('/sys/class/drm/card*/device/xgmi_hive_info/xgmi_hive_id')
                                if os.access(p, os.R_OK)}) or None
    rccl = os.environ.get('GU_RCCL_LIB') or '/opt/rocm/lib/librccl.so'
    out['rccl_library'] = os.path.realpath(rccl) if os.path.exists(rccl) else None
    out['visible_devices_env'] = {k: os.environ[k] for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES') if k in os.environ}
    return out


# --------------------------------------------------------------------------------------- timing
class Ranks(object):
    """Host channel between the ranks: griduniverse_amd.rendezvous (one socket per rank to rank 0; torchrun-style environment,
    no PyTorch).  A no-op for one process."""

    def __init__(self, rank, world):
        from griduniverse_amd.rendezvous import Rendezvous
        self.rank, self.world = rank, world
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        self.rdzv = Rendezvous(rank, world, join_timeout=float(os.environ.get('GU_RDZV_JOIN_TIMEOUT', '600')))
        self.rdzv.barrier()

    def barrier(self):
        self.rdzv.barrier()

    def reduce(self, values, op):
        """Element-wise MAX / MIN over ranks of a list of floats."""
        return self.rdzv.reduce(values, op)

    def gather(self, values):
        """[world][len] of every rank's list of floats."""
        return self.rdzv.gather(values)

    def gather_bytes(self, payload):
        return self.rdzv.gather_bytes(payload)

    def broadcast_bytes(self, payload, src=0):
        return self.rdzv.broadcast_bytes(payload, src)

    def close(self):
        self.rdzv.close()


def timed_block(eng, ranks, T, K):
    """EXACTLY K launches between barrier + device sync pairs.  Returns (wall seconds, HIP-event ms) of this rank."""
    eng.sync()
    ranks.barrier()
    t0 = time.perf_counter()
    eng.timer_begin()
    for _ in range(K):
        eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
    kernel_ms = eng.timer_end()  # HIP events on the engine's stream; also drains it
    eng.sync()
    elapsed = time.perf_counter() - t0
    ranks.barrier()
    return elapsed, kernel_ms


def timed_region(eng, ranks, T, K, min_seconds, max_blocks=4000):
    """One untimed probe block sizes the region (identically on every rank: its time is max-reduced), then B timed blocks.
    Returns per-block wall seconds and HIP-event ms (each MAX over ranks), this rank's own per-block wall seconds, and the
    number of launches issued."""
    probe = ranks.reduce([timed_block(eng, ranks, T, K)[0]], 'MAX')[0]
    blocks = int(min(max_blocks, max(3, np.ceil(min_seconds / max(probe, 1e-6)))))
    wall, kern = [], []
    for _ in range(blocks):
        e, k = timed_block(eng, ranks, T, K)
        wall.append(e)
        kern.append(k)
    both = ranks.reduce(wall + kern, 'MAX')
    return both[:blocks], both[blocks:], wall, (blocks + 1) * K


def other_modes(eng, template, seed, env_id0, N, T, K, check):
    """The same workload in the two launch forms that do not stream 12 bytes per env-step (reported beside `value`, never as
    it): per-env statistics only (return, episodes finished: no HBM stream at all, bound by the LDS round trip of the
    K-step transition table), and one packed uint32 per env-step (4 B)."""
    def launch_ms(**kw):
        for _ in range(settle_launches(eng, T, 'uniform', **kw)):
            eng.rollout(T, 'uniform', auto_reset=True, **kw)
        eng.sync()
        eng.timer_begin()
        for _ in range(K):
            eng.rollout(T, 'uniform', auto_reset=True, **kw)
        return eng.timer_end() / K

    out = {}
    eng.seed(seed)
    eng.reset()
    eng.rollout(T, 'uniform', auto_reset=True, trajectory=False, stats=True)
    ret, episodes = eng.read_stats()
    ok = cpu_baseline_check_stats(template, seed, env_id0, T, ret, episodes) if check else None
    ms = launch_ms(trajectory=False, stats=True)
    out['stats_only'] = {'ms_per_launch': ms, 'value': float(N) * T / ms * 1e3, 'unit': 'env-steps/s (this rank)',
                         'returns_vs_oracle': ok, 'mean_return_per_env': float(np.mean(ret)),
                         'is': 'per-env return and episodes finished instead of the trajectory; first launch from reset checked '
                               'against oracle/gu_oracle.c'}
    if hasattr(eng, 'vi_set'):  # the sampled table policy (the producer of Monte-Carlo evaluation): actions ~ pi[s] by inverse CDF on RNG stream 2
        S = template.world.size
        eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))

        def sample_ms(**kw):
            for _ in range(settle_launches(eng, T, 'sample', **kw)):
                eng.rollout(T, 'sample', auto_reset=True, **kw)
            eng.sync()
            eng.timer_begin()
            for _ in range(K):
                eng.rollout(T, 'sample', auto_reset=True, **kw)
            return eng.timer_end() / K
        ms = sample_ms(trajectory=True)
        out['rollout_sample_policy_traj'] = {'ms_per_launch': ms, 'value': float(N) * T / ms * 1e3, 'unit': 'env-steps/s (this rank)',
                                             'bytes_per_env_step': BYTES_PER_ENV_STEP, 'achieved_GBps': BYTES_PER_ENV_STEP * float(N) * T / ms / 1e6,
                                             'frac_of_hbm_peak': BYTES_PER_ENV_STEP * float(N) * T / ms / 1e6 / HBM_PEAK_GBPS,
                                             'is': 'actions sampled from a random stochastic policy table (Dirichlet(1) rows) instead of uniform; int32 rows'}
        ms = sample_ms(trajectory=False, stats=True)
        out['rollout_sample_policy_stats_only'] = {'ms_per_launch': ms, 'value': float(N) * T / ms * 1e3, 'unit': 'env-steps/s (this rank)'}
    if hasattr(eng, 'read_trajectory_packed'):
        ms = launch_ms(trajectory='packed')
        out['packed_rows'] = {'ms_per_launch': ms, 'value': float(N) * T / ms * 1e3, 'unit': 'env-steps/s (this rank)',
                              'bytes_per_env_step': 4, 'achieved_GBps': 4.0 * N * T / ms / 1e6,
                              'frac_of_hbm_peak': 4.0 * N * T / ms / 1e6 / HBM_PEAK_GBPS,
                              'bound': 'the dependent chain of one wave per SIMD, not the memory (the closed loop of the store pacing finds the limiter '
                                       'useless for this kind and switches it off): %d pairs of steps x (one ds_read_b64 round trip ~85 clocks + the issue of '
                                       'two 256-byte stores at ~25 clocks each) = ~%d clocks, + ~5 us of table staging and the first step' % (T // 2, T // 2 * 135),
                              'floor_us': T // 2 * 135 / 2.4e3 + 5.0,
                              'is': 'obs | reward << 16 | done << 24 in one uint32 per env-step'}
    return out


@contextlib.contextmanager
def native_stdout_to_stderr():
    """RCCL prints a banner (ROCm version, hostname, library path) to the C-level stdout when a communicator comes up, and C
    stdio flushes it whenever it likes -- after the JSON line, when stdout is a pipe.  The driver reads ONE JSON line from
    stdout, so everything native code prints inside this block goes to stderr instead."""
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def spread(values):
    v = np.sort(np.asarray(values, dtype=np.float64))
    return float(v[0]), float(np.median(v)), float(v[-1])


def read_traffic(mode, launch_ms=None):
    """HBM bytes per launch of bench mode `mode` ('headline', 'strong_c4', 'packed_rows', 'stats_only') from the committed
    rocprofv3 --pmc passes over THIS script (tools/gpu_profile.sh -> profiles/rollout_pmc_latest.json), with the tag and date of
    the profile and its own kernel duration -- and a note when that duration and this run's differ by more than 5 %.
    Counters cannot be read inside an unprofiled run: the figure is a property of the kernel and its launch shape, re-measured
    by every profile pass, and is labelled as coming from a file."""
    path = os.path.join(ROOT, 'profiles', 'rollout_pmc_latest.json')
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    entry = table.get('modes', {}).get(mode) if 'modes' in table else (table if mode == 'headline' else None)
    if not entry:
        return None
    out = dict(entry)
    out.setdefault('tag', table.get('tag'))
    out.setdefault('date', table.get('date'))
    prof_us = out.get('kernel_avg_us')
    if prof_us and launch_ms:
        ratio = launch_ms * 1e3 / prof_us
        out['this_run_over_profile_duration'] = ratio
        if abs(ratio - 1.0) > 0.05:
            out['note'] = 'kernel duration differs from the profiled run by %+.1f %% (profile %.1f us, this run %.1f us): the traffic ' \
                          'figure is per launch and does not depend on it, the achieved rate does' % ((ratio - 1.0) * 100, prof_us, launch_ms * 1e3)
    return out


def under_a_profiler():
    """True when this process already runs under rocprofv3 (tools/gpu_profile.sh): no nested counter passes then."""
    return 'rocprof' in os.environ.get('LD_PRELOAD', '') or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)


def live_traffic(args, N, T, budget_s=150):
    """HBM bytes per launch of the headline kernel MEASURED FOR THIS RUN: two short child runs of this very script under
    `rocprofv3 --pmc` -- WRITE_SIZE and FETCH_SIZE in separate passes, counters only (no trace domain), as
    MI355X_MICROARCH.md's HBM section prescribes -- on the same device, right after the timed region.  Each child launches the
    bench kernel a few times on the bench workload (`--pmc-child`); the counter rows of `gu_rollout_kernel<...>` dispatches of
    this launch size are averaged (the first launch, with cold caches, excluded).  bytes = WRITE_SIZE * 1024 + 2 * FETCH_SIZE *
    1024 (both counters are in KiB; on gfx950 FETCH_SIZE reports half of a coalesced read stream).  None when rocprofv3 is not
    there, takes too long or reports nothing -- the committed profile's figure is used then, and labelled so."""
    import csv
    tool = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if tool is None or under_a_profiler():
        return None
    work = tempfile.mkdtemp(prefix='gu_pmc_', dir='/tmp')
    t0 = time.time()
    sums = {}
    try:
        for counter in ('WRITE_SIZE', 'FETCH_SIZE'):
            out = os.path.join(work, counter)
            cmd = [tool, '--pmc', counter, '--output-format', 'csv', '-d', out, '--', sys.executable, os.path.abspath(__file__),
                   '--pmc-child', '--envs', str(N), '--T', str(T), '--workload', args.workload]
            left = budget_s - (time.time() - t0)
            if left < 20:
                return None
            # (its own session: if rocprofv3 spawns the program instead of exec'ing it, a timeout must take the whole group down)
            child = subprocess.Popen(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                     start_new_session=True)
            try:
                child.wait(timeout=left)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(child.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
                child.wait()
                return None
            proc = child
            values = []
            for path in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
                with open(path, newline='') as f:
                    for row in csv.DictReader(f):
                        if 'gu_rollout_kernel<' in row['Kernel_Name'] and int(row['Grid_Size']) == N and row['Counter_Name'] == counter:
                            values.append((int(row['Dispatch_Id']), float(row['Counter_Value'])))
            values = [v for _, v in sorted(values)][1:]  # (the first launch writes into cold caches)
            if proc.returncode != 0 or not values:
                return None
            sums[counter] = (sum(values) / len(values), len(values))
    except (OSError, subprocess.SubprocessError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)
    wr, rd = sums['WRITE_SIZE'][0] * 1024.0, 2.0 * sums['FETCH_SIZE'][0] * 1024.0
    return dict(hbm_bytes_per_launch=wr + rd, write_bytes=wr, read_bytes_corrected=rd, dispatches_counted=sums['WRITE_SIZE'][1],
                seconds=time.time() - t0,
                source='rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE: two child runs of this script (bench.py --pmc-child: the bench kernel on the '
                       'bench workload, %d launches counted each) on this device right after the timed region; bytes = WRITE_SIZE*1024 + '
                       '2*FETCH_SIZE*1024 (FETCH_SIZE x 2: on gfx950 the counter tallies 128-byte read requests at 64 bytes, MI355X_MICROARCH.md "HBM")' % sums['WRITE_SIZE'][1])


def pmc_child(args):
    """`bench.py --pmc-child` (started by live_traffic under rocprofv3 --pmc): the bench kernel, nine launches, nothing else."""
    template, _ = build_workload(args.workload)
    eng = gua.Engine(args.envs, gua.GridSpec.from_env(template), device=0, env_id0=0, seed=WORKLOAD_SEED[args.workload])
    eng.set_option('traj_candidates', 1)  # (no placement search under the profiler: every probe launch would be counted too)
    eng.reset()
    eng.reserve_trajectory(args.T)
    for _ in range(9):
        eng.rollout(args.T, 'uniform', auto_reset=True, trajectory=True)
    eng.sync()
    eng.close()


def device_block(engine_cls, device):
    """What the device looked like during the run: gu_device_info (name, arch, CUs, clocks as HIP reports them) plus the sysfs
    view of the same PCI function -- current sclk / mclk, power cap, memory and compute partition -- so that a slow run can be
    told from a differently configured box."""
    if not hasattr(engine_cls, 'device_info'):
        return None
    try:
        info = dict(engine_cls.device_info(device))
    except Exception as err:  # noqa: BLE001 -- reporting only
        return {'error': str(err)}
    pci = str(info.get('pci', '')).lower()
    base = '/sys/bus/pci/devices/' + pci
    sysfs = {}

    def read(rel):
        try:
            with open(os.path.join(base, rel)) as f:
                return f.read().strip()
        except OSError:
            return None

    if pci and os.path.isdir(base):
        for key, rel in (('memory_partition', 'current_memory_partition'), ('compute_partition', 'current_compute_partition'),
                         ('perf_level', 'power_dpm_force_performance_level'), ('vbios', 'vbios_version'),
                         ('gpu_busy_percent', 'gpu_busy_percent'), ('mem_busy_percent', 'mem_busy_percent')):
            v = read(rel)
            if v is not None:
                sysfs[key] = v
        for key, rel in (('sclk', 'pp_dpm_sclk'), ('mclk', 'pp_dpm_mclk'), ('fclk', 'pp_dpm_fclk')):
            v = read(rel)
            if v is not None:
                levels = [ln.strip() for ln in v.splitlines() if ln.strip()]
                sysfs[key + '_levels'] = levels
                sysfs[key + '_current'] = next((ln.rstrip(' *').split(':', 1)[-1].strip() for ln in levels if ln.endswith('*')), None)
        for hw in glob.glob(os.path.join(base, 'hwmon', 'hwmon*')):
            for key, rel in (('power_cap_uW', 'power1_cap'), ('power_cap_max_uW', 'power1_cap_max'), ('power_average_uW', 'power1_average'),
                             ('power_input_uW', 'power1_input'), ('temp_edge_mC', 'temp1_input'), ('temp_hbm_mC', 'temp3_input')):
                try:
                    with open(os.path.join(hw, rel)) as f:
                        sysfs[key] = int(f.read().strip())
                except (OSError, ValueError):
                    pass
    info['sysfs'] = sysfs or None
    return info


def settle_launches(eng, T, policy, **kw):
    """Untimed launches in front of a timed block of a launch kind: 3 -- or, for a kind whose rows keep a schedule, the few hundred
    the closed loop of the store pacing takes to come down from its model period (it runs inside the launches themselves: nothing
    else is asked of the engine; 200 launches = 20 ms at the headline size)."""
    eng.rollout(T, policy, auto_reset=True, **kw)
    paced = hasattr(eng, 'rollout_pacing') and eng.rollout_pacing(policy, True, packed=kw.get('trajectory') == 'packed') is not None
    return 200 if paced else 3


def pacing_block(eng):
    """roofline.store_pacing: where the closed loop of the rollout kernel's rate limiter stands for the bench launch (the waves'
    schedule: 10 ns ticks per 16 steps), and the records of its last launches."""
    if not hasattr(eng, 'rollout_pacing'):
        return None
    info = eng.rollout_pacing('uniform', True)
    totals = eng.rollout_pacing_totals() if hasattr(eng, 'rollout_pacing_totals') else None
    if info is None:
        return {'paced': False, 'totals': totals}
    info['paced'] = True
    info['requested'] = 'nothing: bench.py only launches (rounds 3 and 4 called gu_rollout_calibrate before the warm-up); totals.launches_spent ' \
                        'counts launches the engine issued for itself'
    info['totals'] = totals
    if hasattr(eng, 'rollout_pace_log'):
        lg = eng.rollout_pace_log('uniform', True)
        iv = lg['interval'][lg['interval'] > 0]
        info['last_launches'] = {
            'launches_of_the_kind': int(lg['launches']), 'periods': [round(float(x), 2) for x in lg['period'][-16:]],
            'phase': [int(x) for x in lg['phase'][-16:]],
            'launches_in_log': int(len(lg['seq'])), 'launches_behind_in_log': int((lg['verdict'] == 2).sum()),
            'waves_behind_share_in_log': float(lg['ended_late'].sum()) / max(1, int(lg['waves'].sum())),
            'start_to_start_us_median': float(np.median(iv)) / 100.0 if len(iv) else None,
            'is': 'the kind\'s ring of launch records on the device: the period each of the last launches ran with (0 = without the limiter), '
                  'how many waves reported more than two periods behind their schedule, the device-clock time from one launch\'s start to the next'}
    info['is'] = 'the HBM write path collapses when it is over-driven (5.7 TB/s on most allocations): every wave keeps a schedule -- its ' \
                 'next 16 steps begin no earlier than `period` ticks of 10 ns after the last ones were due, late waves do not wait.  The ' \
                 'period is chosen by the launches themselves, closed loop, on the device: every wave reports whether it fell behind, the ' \
                 'first wave of the next launch sums the reports and moves the period of the launch after it (up by the share of waves ' \
                 'behind, down by a quarter tick per launch), and every 1024 launches three launches run without the limiter to see ' \
                 'whether it pays at all (DESIGN.md section 6; gu_rollout.hpp: GuPacer)'
    return info


def placement_block(eng, post_probe_ms, launch_ms):
    """roofline.trajectory_placement: what gu_reserve_trajectory's candidate search did for the bench buffer, per candidate, what
    it cost, and the SAME store probe run once more on the kept buffer right after the timed region (so that "the probe said
    0.132 ms, the kernel took 0.140" can be split into drift of the device and cost of the kernel)."""
    if not hasattr(eng, 'trajectory_placement'):
        return None
    n, best, worst = eng.trajectory_placement()
    out = {'candidates_probed': n, 'probe_ms_kept': best, 'probe_ms_slowest': worst}
    if hasattr(eng, 'trajectory_placement_detail'):
        out.update(eng.trajectory_placement_detail())
    out['probe_ms_kept_after_timed_region'] = post_probe_ms
    if post_probe_ms and best:
        out['probe_drift'] = post_probe_ms / best
    if post_probe_ms and launch_ms:
        out['kernel_over_probe_after'] = launch_ms / post_probe_ms
    out['is'] = 'gu_reserve_trajectory writes candidate allocations once in the rollout\'s store shape and keeps the fastest (where a ' \
                'buffer lands in HBM changes its write rate by ~15 %, DESIGN.md section 6); probe_ms = one full write of the buffer ' \
                'by a bare store loop; peak_bytes = most memory the search held; the search stops after the back-to-back candidates ' \
                'when they are within 6 % of each other'
    return out


# --------------------------------------------------------------------------------------- RCCL gathered view
def rccl_view_check(eng, engine_cls, ranks):
    """The single-array (obs, reward, done) view over RCCL, outside the timed region: one ncclAllGather of every rank's
    packed int32[3N] block.  Proves the collective saw `world` ranks: every rank's own shard digest travels over gloo and
    is compared with the digest of that rank's slice of the RCCL view."""
    world, rank = ranks.world, ranks.rank
    if hasattr(engine_cls, 'host_channel'):  # (the oracle-backed stub of the CPU tests gathers over the host channel)
        engine_cls.host_channel = ranks.rdzv
    uid = engine_cls.comm_unique_id() if rank == 0 else bytes(_lib.COMM_ID_BYTES)
    uid = ranks.broadcast_bytes(uid, 0)
    ranks.barrier()
    t0 = time.perf_counter()
    eng.comm_init(world, rank, uid)
    init_ms = (time.perf_counter() - t0) * 1e3
    view = eng.allgather_view()  # first call: untimed (lazy connection set-up)
    laps = []
    for _ in range(5):
        ranks.barrier()
        t0 = time.perf_counter()
        view = eng.allgather_view()
        laps.append((time.perf_counter() - t0) * 1e3)
    own = eng.read_outputs()
    n = own[0].size
    digest = hashlib.sha256(b''.join(np.ascontiguousarray(a, dtype='<i4').tobytes() for a in own)).digest()
    shard_digests = ranks.gather_bytes(digest)
    equal = all(v.size == world * n for v in view)
    for r in range(world):
        got = hashlib.sha256(b''.join(np.ascontiguousarray(v[r * n:(r + 1) * n], dtype='<i4').tobytes() for v in view)).digest()
        equal = equal and got == shard_digests[r]
    equal = ranks.reduce([1.0 if equal else 0.0], 'MIN')[0] == 1.0  # every rank checked every slice of ITS copy of the view
    lap = ranks.reduce([float(np.median(laps))], 'MAX')[0]
    eng.comm_destroy()
    return dict(nranks=world, comm_init_ms=ranks.reduce([init_ms], 'MAX')[0], allgather_ms=lap, bytes_per_rank=3 * n * 4,
                view_envs=world * n, view_equals_shards=bool(equal),
                note='ncclAllGather of the packed (obs|reward|done) int32[3N] block per rank + D2H of the view; '
                     'median of 5 calls, max over ranks; compared slice by slice with every rank\'s own shard')


# --------------------------------------------------------------------------------------- config 4, strong scaling
def strong_c4(args, ranks, engine_cls, device):
    """BASELINE.json config 4: 262 144 envs on the 32x32 lava grid IN TOTAL, sharded over the ranks by env index (strong
    scaling: 262 144 / world envs per GPU), seed 4.  One checked launch (250 steps from reset: every rank compares its shard
    with the C oracle; on one GPU the whole batch is also hashed against the reference's digest), then timed blocks of K
    launches of T steps."""
    world, rank = ranks.world, ranks.rank
    total = args.c4_envs
    if total % world:
        return dict(skipped='%d envs do not divide over %d ranks' % (total, world))
    n, seed, T_check = total // world, WORKLOAD_SEED['c4'], 250
    template, desc = build_workload('c4')
    eng = engine_cls(n, gua.GridSpec.from_env(template), device=device, env_id0=rank * n, seed=seed)
    try:
        eng.reset()
        eng.reserve_trajectory(max(args.T, T_check))
        eng.rollout(T_check, 'uniform', auto_reset=True, trajectory=True)
        eng.sync()
        got = eng.read_trajectory(0, T_check)
        shard_ok = cpu_baseline_check_prefix(template, seed, rank * n, got, n_check=n)
        shards_ok = ranks.reduce([1.0 if shard_ok else 0.0], 'MIN')[0] == 1.0
        ref = reference_digest('c4', template, seed, n, T_check, rank * n) if world == 1 else None
        ref_ok = None if ref is None else sha256_triplet(got) == ref
        del got
        for _ in range(max(args.warmup, settle_launches(eng, args.T, 'uniform', trajectory=True))):
            eng.rollout(args.T, 'uniform', auto_reset=True, trajectory=True)
        wall, kern, _, _ = timed_region(eng, ranks, args.T, args.steps, args.min_seconds / 2)
        pacing = pacing_block(eng)
        if pacing:
            pacing.pop('is', None)  # (explained once, in roofline.store_pacing)
    finally:
        eng.close()
    w_min, w_med, w_max = spread(wall)
    k_med = spread(kern)[1]
    K = args.steps
    return dict(value=float(total) * args.T * K / w_med, unit='env-steps/s', scaling='strong', total_envs=total, envs_per_gpu=n,
                n_gpus=world, env_steps_per_launch=args.T, steps=K, blocks=len(wall), ms_per_step=w_med / K * 1e3,
                ms_per_step_min=w_min / K * 1e3, ms_per_step_max=w_max / K * 1e3, launch_ms=k_med / K,
                hbm_gbps_per_gpu=BYTES_PER_ENV_STEP * n * args.T / (k_med / K / 1e3) / 1e9,
                workload='c4: %s, seed %d, uniform device-RNG actions, auto-reset, int32 trajectory' % (desc, seed),
                shards_equal_oracle=bool(shards_ok), bit_exact_vs_reference_digest=ref_ok, store_pacing=pacing,
                check='first launch (250 steps from reset): every rank\'s full shard trajectory == C oracle'
                      + ('; whole batch sha256 == reference digest c4_lava32_262144x250' if ref is not None else ''))


# --------------------------------------------------------------------------------------- plumbing
def ensure_library_is_current(engine_cls, local_rank=0):
    """A checkout whose libgu.so is missing or older than its sources: build it (one rank per node) rather than measure
    nothing.  _lib.is_stale() reads the hash from the file's bytes, so looking never maps the library."""
    if engine_cls is not gua.Engine or not _lib.is_stale():
        return
    if local_rank == 0:
        with native_stdout_to_stderr():  # (make's and hipcc's chatter belongs on stderr: stdout carries the one JSON line)
            _lib.build()
    else:
        deadline = time.time() + 900
        while _lib.is_stale() and time.time() < deadline:
            time.sleep(2)


def spawn_ranks(args, argv, engine_cls=None, script=None):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE in the environment): this process becomes a launcher.  It
    starts N fresh children -- one rank each, torchrun-style environment, each the leader of its own process group -- BEFORE
    anything here has touched a GPU or loaded libgu.so, relays rank 0's JSON line, and returns the worst exit code.  (Never an
    exec of a process that has initialised the GPU: the children are ordinary subprocesses and this parent never calls into HIP.)
    EVERY child is watched: the first one that dies with an error takes the others down with it at once -- the survivors would
    otherwise sit in the rendezvous until its timeout, silently, holding their GPUs -- and SIGTERM / SIGINT to the launcher
    (an outer `timeout`) are passed on to all of them."""
    import signal
    import threading

    ensure_library_is_current(engine_cls or gua.Engine)  # a subprocess `make`: no HIP call in this process
    with socket.socket() as sck:
        sck.bind(('127.0.0.1', 0))
        port = sck.getsockname()[1]
    script = script or os.path.abspath(__file__)
    token = os.urandom(16).hex()  # the ranks of THIS launch (griduniverse_amd/rendezvous.py turns away anyone else)
    procs = []

    def kill_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, _frame):
        kill_all(signal.SIGTERM)
        time.sleep(0.5)
        kill_all(signal.SIGKILL)
        sys.exit(128 + signum)

    previous = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    out = []
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), GU_RDZV_JOIN_TIMEOUT=os.environ.get('GU_RDZV_JOIN_TIMEOUT', '120'), GU_RDZV_TOKEN=token)
            procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, start_new_session=True,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
        reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        failed = None
        while any(p.poll() is None for p in procs):
            failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)), None)
            if failed is not None:
                sys.stderr.write('bench.py: rank %d exited with code %d: stopping the other ranks\n' % failed)
                kill_all(signal.SIGTERM)
                deadline = time.time() + 5
                while time.time() < deadline and any(p.poll() is None for p in procs):
                    time.sleep(0.05)
                kill_all(signal.SIGKILL)
                break
            time.sleep(0.05)
        codes = [p.wait() for p in procs]
        reader.join(timeout=5)
    finally:
        kill_all(signal.SIGKILL)
        for sig, handler in previous.items():
            signal.signal(sig, handler)
    if out and failed is None:
        sys.stdout.write(out[0].decode('utf-8', 'replace'))
        sys.stdout.flush()
    worst = failed[1] if failed is not None else next((c for c in codes if c != 0), 0)
    if worst:
        sys.stderr.write('bench.py: rank exit codes %r\n' % (codes,))
    return worst


# --------------------------------------------------------------------------------------- one process, N devices
def run_single_process(args, engine_cls=None, emit=print):
    """SURVEY.md 8(e)'s form: ONE host process, one engine (handle + HIP stream) per device, contiguous env-index shards with
    global env ids g * N .., every launch enqueued device after device so that the GPUs run concurrently; the single-array view
    through ncclCommInitAll + one grouped ncclAllGather (gu_comm_init_all / gu_allgather_view_all).  Timed like the
    multi-process form: blocks of exactly K launches PER DEVICE between device syncs of all devices; `value` = all devices'
    env-steps / median block wall time; per_rank = every device's own HIP-event time."""
    engine_cls = engine_cls or gua.Engine
    ensure_library_is_current(engine_cls)
    G, N, T, K, W = args.gpus, args.envs, args.T, args.steps, args.warmup
    seed = WORKLOAD_SEED[args.workload]
    template, grid_desc = build_workload(args.workload)
    want_cpu = G == 1 and not args.no_cpu_baseline
    all_cores = cpu_baseline_all_cores(template, seed) if want_cpu else None  # forks: must precede any HIP call here
    n_dev = max(1, _lib.device_count()) if engine_cls is gua.Engine else 1
    devices = [g % n_dev for g in range(G)]  # identity on a G-GPU node; a smaller box rehearses the flow with shared devices
    spec = gua.GridSpec.from_env(template)
    engines = [engine_cls(N, spec, device=devices[g], env_id0=g * N, seed=seed) for g in range(G)]

    def launch_all():
        for e in engines:
            e.rollout(T, 'uniform', auto_reset=True, trajectory=True)

    def block():
        for e in engines:
            e.sync()
        t0 = time.perf_counter()
        for e in engines:
            e.timer_begin()
        for _ in range(K):
            launch_all()
        kernel_ms = [e.timer_end() for e in engines]  # (each waits for its own device)
        return time.perf_counter() - t0, kernel_ms

    try:
        for e in engines:
            e.reset()
            e.reserve_trajectory(T)
        launches = 1
        for e in engines:  # (the first launch from reset: checked in full below)
            e.rollout(T, 'uniform', auto_reset=True, trajectory=True)
        checks = {}
        if not args.no_checks:
            first = engines[0].read_trajectory(0, T)
            ref = reference_digest(args.workload, template, seed, N, T, 0)
            checks['bit_exact_vs_reference_digest'] = None if ref is None else sha256_triplet(first) == ref
            ok = bool(cpu_baseline_check_prefix(template, seed, 0, first))
            del first
            for g, e in enumerate(engines[1:], start=1):  # every other shard: its first envs against the oracle at ITS global ids
                ok = ok and bool(cpu_baseline_check_prefix(template, seed, g * N, e.read_trajectory(0, T), n_check=512))
            checks['bit_exact_vs_oracle'] = ok
        for _ in range(W):
            launch_all()
        launches += W
        probe = block()[0]
        blocks = int(min(4000, max(3, np.ceil(args.min_seconds / max(probe, 1e-6)))))
        wall, kern = [], []
        for _ in range(blocks):
            w, k = block()
            wall.append(w)
            kern.append(k)
        launches += (blocks + 1) * K
        dev_info = [device_block(engine_cls, d) for d in sorted(set(devices))]
        post_probe = [e.probe_trajectory() if hasattr(e, 'probe_trajectory') else None for e in engines]
        if not args.no_checks:
            for e in engines:
                e.sync()
            checks['final_state_vs_oracle'] = cpu_baseline_check_final_state(template, seed, 0, N, launches * T, engines[0].get_state())
            checks['final_state_vs_oracle']['launches'] = launches
        # ---- the gathered view: one communicator over all devices of this process, one grouped all-gather
        rccl = None
        if G > 1 or args.gather_view:
            try:
                with native_stdout_to_stderr():
                    t0 = time.perf_counter()
                    engine_cls.comm_init_all(engines)
                    init_ms = (time.perf_counter() - t0) * 1e3
                    view = engine_cls.allgather_view_all(engines)  # first call: untimed (lazy connection set-up)
                    laps = []
                    for _ in range(5):
                        t0 = time.perf_counter()
                        view = engine_cls.allgather_view_all(engines)
                        laps.append((time.perf_counter() - t0) * 1e3)
                equal = all(v.size == G * N for v in view)
                for g, e in enumerate(engines):
                    own = e.read_outputs()
                    equal = equal and all(np.array_equal(view[k][g * N:(g + 1) * N], own[k]) for k in range(3))
                rccl = dict(nranks=G, comm_init_ms=init_ms, allgather_ms=float(np.median(laps)), bytes_per_rank=3 * N * 4, view_envs=G * N,
                            view_equals_shards=bool(equal), form='ncclCommInitAll + one grouped ncclAllGather from one process',
                            note='packed (obs|reward|done) int32[3N] block per device; view read from the first device; median of 5 '
                                 'calls; compared slice by slice with every device\'s own shard')
            except gua.GuError as err:  # reported, not fatal (a box with fewer devices than ranks: RCCL wants one device per rank)
                rccl = dict(nranks=G, view_equals_shards=None, error=str(err))
        placement = [placement_block(e, post_probe[g], float(np.median([k[g] for k in kern])) / K) for g, e in enumerate(engines)]
        pacing = [pacing_block(e) for e in engines]
    finally:
        for e in engines:
            e.close()

    # ---- config 4, strong scaling, same form
    c4 = None
    if not args.no_strong_c4 and args.c4_envs % G == 0:
        n, seed4, T_check = args.c4_envs // G, WORKLOAD_SEED['c4'], 250
        template4, desc4 = build_workload('c4')
        spec4 = gua.GridSpec.from_env(template4)
        engines = [engine_cls(n, spec4, device=devices[g], env_id0=g * n, seed=seed4) for g in range(G)]
        try:
            for e in engines:
                e.reset()
                e.reserve_trajectory(max(T, T_check))
            for e in engines:
                e.rollout(T_check, 'uniform', auto_reset=True, trajectory=True)
            shards_ok = all(bool(cpu_baseline_check_prefix(template4, seed4, g * n, e.read_trajectory(0, T_check), n_check=n))
                            for g, e in enumerate(engines))
            ref = reference_digest('c4', template4, seed4, n, T_check, 0) if G == 1 else None
            ref_ok = None if ref is None else sha256_triplet(engines[0].read_trajectory(0, T_check)) == ref
            for _ in range(W):
                launch_all()
            probe4 = block()[0]
            blocks4 = int(min(4000, max(3, np.ceil(args.min_seconds / 2 / max(probe4, 1e-6)))))
            wall4, kern4 = [], []
            for _ in range(blocks4):
                w, k = block()
                wall4.append(w)
                kern4.append(max(k))
        finally:
            for e in engines:
                e.close()
        w_min, w_med, w_max = spread(wall4)
        k_med = spread(kern4)[1]
        c4 = dict(value=float(args.c4_envs) * T * K / w_med, unit='env-steps/s', scaling='strong', total_envs=args.c4_envs, envs_per_gpu=n,
                  n_gpus=G, env_steps_per_launch=T, steps=K, blocks=len(wall4), ms_per_step=w_med / K * 1e3, ms_per_step_min=w_min / K * 1e3,
                  ms_per_step_max=w_max / K * 1e3, launch_ms=k_med / K, hbm_gbps_per_gpu=BYTES_PER_ENV_STEP * n * T / (k_med / K / 1e3) / 1e9,
                  workload='c4: %s, seed %d, uniform device-RNG actions, auto-reset, int32 trajectory' % (desc4, seed4),
                  shards_equal_oracle=bool(shards_ok), bit_exact_vs_reference_digest=ref_ok,
                  check='first launch (250 steps from reset): every device\'s full shard trajectory == C oracle')
        c4['traffic'] = read_traffic('strong_c4', c4['launch_ms'])
    elif not args.no_strong_c4:
        c4 = dict(skipped='%d envs do not divide over %d devices' % (args.c4_envs, G))

    w_min, w_med, w_max = spread(wall)
    per_dev_ms = [float(np.median([k[g] for k in kern])) / K for g in range(G)]
    worst = [max(k) for k in kern]
    k_min, k_med, k_max = spread(worst)
    launch_s = k_med / 1e3 / K
    achieved = BYTES_PER_ENV_STEP * N * T / launch_s / 1e9
    traffic = read_traffic('headline', launch_s * 1e3)
    steps_per_block = float(G) * N * T * K
    line = {
        'metric': METRIC, 'value': steps_per_block / w_med, 'unit': 'env-steps/s', 'n_gpus': G, 'steps': K, 'warmup': W,
        'ms_per_step': w_med / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int32',
        'data': 'synthetic',
        'config': {'workload': '%s: %d envs per GPU on the %s, uniform random actions from the per-env device RNG, auto-reset on done, '
                               'one launch = %d env-steps per env, int32 (obs,reward,done) trajectory written to HBM' % (args.workload, N, grid_desc, T),
                   'envs_per_gpu': N, 'env_steps_per_launch': T, 'global_envs': G * N,
                   'parallelism': 'env-index shards, no data-path collective; ONE host process, one engine per device, launches '
                                  'enqueued device after device', 'devices': devices},
        'mode': 'single-process',
        'timing': {'blocks': blocks, 'launches_per_block': K, 'timed_seconds': float(np.sum(wall)),
                   'value_is': 'median block (each block = exactly K launches per device between syncs of every device)',
                   'ms_per_step_min': w_min / K * 1e3, 'ms_per_step_median': w_med / K * 1e3, 'ms_per_step_max': w_max / K * 1e3,
                   'value_min': steps_per_block / w_max, 'value_max': steps_per_block / w_min,
                   'launch_ms_min': k_min / K, 'launch_ms_median': k_med / K, 'launch_ms_max': k_max / K,
                   'launch_ms_is': 'HIP-event time of a block / K on the slowest device, per block', 'launches_total': launches * G},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS,
                     'traffic': None if traffic is None else traffic.get('hbm_bytes_per_launch'), 'traffic_measured_by_child_runs': False,
                     'kernel': 'gu_rollout_kernel<UNIFORM,TRAJ,LDS>', 'launch_ms': launch_s * 1e3,
                     'algorithmic_bytes_per_launch': BYTES_PER_ENV_STEP * N * T,
                     'traffic_source': None if traffic is None else traffic.get('source'),
                     'vs_measured_copy_rate': achieved / HBM_COPY_GBPS, 'is': 'per device (the slowest one)',
                     'store_pacing': pacing, 'trajectory_placement': placement},
        'device': dev_info,
        'engine': engine_cls.__module__ + '.' + engine_cls.__name__,
        'per_rank': {'ms_per_step': per_dev_ms, 'value': [float(N) * T / (ms / 1e3) for ms in per_dev_ms],
                     'is': 'every device\'s own HIP-event time per launch (median block)'},
        'rccl': rccl, 'strong_c4': c4, 'other_modes': None,
    }
    line.update(checks)
    if want_cpu:
        base = cpu_baseline(template, seed, T)
        base['all_cores'] = all_cores
        line['cpu_baseline'] = base
    ctypes.CDLL(None).fflush(None)
    emit(json.dumps(line))
    sys.stdout.flush()


# --------------------------------------------------------------------------------------- the run
def run(args, engine_cls=None, emit=print):
    """`engine_cls` exists for the CPU tests (tests/_bench_stub.py passes tests/_oracle_engine.py, which stands in for the
    device); this script's own command line can only measure griduniverse_amd.Engine, and the JSON line names the class that ran."""
    engine_cls = engine_cls or gua.Engine
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: start the script plainly (it spawns its ranks) or under '
                         'torch.distributed.run --nproc-per-node %d' % (args.gpus, world, args.gpus))
    ranks = Ranks(rank, world)
    ensure_library_is_current(engine_cls, local_rank)

    N, T, K, W = args.envs, args.T, args.steps, args.warmup
    seed = WORKLOAD_SEED[args.workload]
    template, grid_desc = build_workload(args.workload)
    want_cpu = world == 1 and not args.no_cpu_baseline
    all_cores = cpu_baseline_all_cores(template, seed) if want_cpu else None  # forks: must precede any HIP call here
    n_dev = max(1, _lib.device_count()) if engine_cls is gua.Engine else 1
    device = local_rank % n_dev  # identity on an N-GPU node; lets a 1-GPU box rehearse the N-process flow
    eng = engine_cls(N, gua.GridSpec.from_env(template), device=device, env_id0=rank * N, seed=seed)
    eng.reset()
    eng.reserve_trajectory(T)

    # ---- launch 1, from reset: checked in full.  (Nothing is asked of the engine but the launches themselves: the store pacing of
    # the rollout kernel is a closed loop that runs inside them from the first one on -- rounds 3 and 4 asked for a search here.)
    launches = 1
    eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
    eng.sync()
    checks = {}
    if rank == 0 and not args.no_checks:
        first = eng.read_trajectory(0, T)
        ref = reference_digest(args.workload, template, seed, N, T, rank * N)
        checks['bit_exact_vs_reference_digest'] = None if ref is None else sha256_triplet(first) == ref
        checks['reference_digest'] = None if ref is None else REFERENCE_DIGEST[args.workload] + ' (tests/golden/digests.json: sha256 of ' \
            'the (obs, reward, done) streams the reference\'s own step() produced for this grid, seed, batch and length)'
        checks['bit_exact_vs_oracle'] = bool(cpu_baseline_check_prefix(template, seed, rank * N, first))
        del first
    for _ in range(W):
        eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
    launches += W

    # ---- timed region: blocks of exactly K launches until >= min_seconds
    wall, kern, own_wall, n_launched = timed_region(eng, ranks, T, K, args.min_seconds)
    launches += n_launched
    blocks = len(wall)
    dev_info = device_block(engine_cls, device) if rank == 0 else None  # (clocks as they are right behind the timed launches)

    # ---- one instrumented block: an event after every launch (not part of `value`)
    eng.sync()
    eng.timer_mark()
    for _ in range(K):
        eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
        eng.timer_mark()
    per_launch = eng.timer_laps()
    launches += K
    # ---- the bare store probe once more on the kept buffer, right after the timed launches (overwrites the rows, not the state)
    post_probe_ms = eng.probe_trajectory() if hasattr(eng, 'probe_trajectory') else None

    if rank == 0 and not args.no_checks:
        eng.sync()
        checks['final_state_vs_oracle'] = cpu_baseline_check_final_state(template, seed, rank * N, N, launches * T, eng.get_state())
        checks['final_state_vs_oracle']['launches'] = launches

    others = None
    if not args.no_other_modes and hasattr(eng, 'read_stats'):
        others = other_modes(eng, template, seed, rank * N, N, T, K, rank == 0 and not args.no_checks)

    rccl = None
    if world > 1 or args.gather_view:
        try:
            with native_stdout_to_stderr():
                rccl = rccl_view_check(eng, engine_cls, ranks)
        except gua.GuError as err:  # reported, not fatal: the throughput line above does not depend on the collective
            rccl = dict(nranks=world, view_equals_shards=None, error=str(err))
    per_rank = ranks.gather([float(np.median(own_wall))])
    placement = placement_block(eng, post_probe_ms, float(np.median(kern)) / K)
    pacing = pacing_block(eng)
    eng.close()
    c4 = None if args.no_strong_c4 else strong_c4(args, ranks, engine_cls, device)
    configs = None
    if world == 1 and not args.no_configs:
        configs = baseline_configs(engine_cls, device, K, not args.no_checks)

    if rank == 0:
        w_min, w_med, w_max = spread(wall)
        k_min, k_med, k_max = spread(kern)
        launch_s = k_med / 1e3 / K
        achieved = BYTES_PER_ENV_STEP * N * T / launch_s / 1e9
        traffic = read_traffic('headline', launch_s * 1e3)
        measured = None
        if world == 1 and engine_cls is gua.Engine and not args.no_live_traffic:
            measured = live_traffic(args, N, T)
        if c4 and 'launch_ms' in c4:
            c4['traffic'] = read_traffic('strong_c4', c4['launch_ms'])
        for mode in (others or {}):
            others[mode]['traffic'] = read_traffic(mode, others[mode]['ms_per_launch'])
        steps_per_block = float(world) * N * T * K
        line = {
            'metric': METRIC, 'value': steps_per_block / w_med, 'unit': 'env-steps/s', 'n_gpus': world,
            'steps': K, 'warmup': W, 'ms_per_step': w_med / K * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'int32', 'data': 'synthetic',
            'config': {'workload': '%s: %d envs per GPU on the %s, uniform random actions from the per-env device RNG, '
                                   'auto-reset on done, one launch = %d env-steps per env, int32 (obs,reward,done) '
                                   'trajectory written to HBM' % (args.workload, N, grid_desc, T),
                       'envs_per_gpu': N, 'env_steps_per_launch': T, 'global_envs': world * N,
                       'parallelism': 'env-index shards, no data-path collective'},
            'timing': {'blocks': blocks, 'launches_per_block': K, 'timed_seconds': float(np.sum(wall)),
                       'value_is': 'median block (each block = exactly K launches between barrier + device sync pairs, max over ranks)',
                       'ms_per_step_min': w_min / K * 1e3, 'ms_per_step_median': w_med / K * 1e3, 'ms_per_step_max': w_max / K * 1e3,
                       'value_min': steps_per_block / w_max, 'value_max': steps_per_block / w_min,
                       'launch_ms_min': k_min / K, 'launch_ms_median': k_med / K, 'launch_ms_max': k_max / K,
                       'launch_ms_is': 'HIP-event time of a block / K, per block',
                       'per_launch_ms_min': float(per_launch.min()), 'per_launch_ms_median': float(np.median(per_launch)),
                       'per_launch_ms_max': float(per_launch.max()),
                       'per_launch_ms_is': 'one extra, untimed block with an event after every launch (%d launches)' % per_launch.size,
                       'launches_total': launches},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS,
                         'frac_wall': BYTES_PER_ENV_STEP * N * T / (w_med / K) / 1e9 / HBM_PEAK_GBPS,
                         'frac_is': 'frac: algorithmic bytes / the kernels\' own HIP-event time per launch; frac_wall: / ms_per_step, the wall '
                                    'time the driver\'s clock sees (barriers and launch overhead included)',
                         'traffic': measured['hbm_bytes_per_launch'] if measured else None if traffic is None else traffic.get('hbm_bytes_per_launch'),
                         'traffic_measured_by_child_runs': bool(measured),
                         'traffic_over_algorithmic': (measured['hbm_bytes_per_launch'] if measured else (traffic or {}).get('hbm_bytes_per_launch', 0.0))
                         / float(BYTES_PER_ENV_STEP * N * T) or None,
                         'kernel': 'gu_rollout_kernel<UNIFORM,TRAJ,LDS>', 'launch_ms': launch_s * 1e3,
                         'algorithmic_bytes_per_launch': BYTES_PER_ENV_STEP * N * T,
                         'traffic_source': measured['source'] if measured else None if traffic is None else traffic.get('source'),
                         'traffic_live': measured,
                         'traffic_profile': None if traffic is None else {k: traffic.get(k) for k in
                                                                          ('tag', 'date', 'kernel', 'kernel_avg_us', 'this_run_over_profile_duration', 'note')},
                         'vs_measured_copy_rate': achieved / HBM_COPY_GBPS,
                         'store_pacing': pacing,
                         'trajectory_placement': placement},
            'device': dev_info,
            'engine': engine_cls.__module__ + '.' + engine_cls.__name__,
            'per_rank': {'ms_per_step': [v[0] / K * 1e3 for v in per_rank],
                         'value': [float(N) * T * K / v[0] for v in per_rank],
                         'is': 'every rank\'s own median block (the N = 1 run of this script reports exactly this figure as `value`)'},
            'rccl': rccl, 'strong_c4': c4, 'other_modes': others, 'configs': configs,
            'topology': topology_block(engine_cls),
        }
        line.update(checks)
        if want_cpu:
            base = cpu_baseline(template, seed, T)
            base['all_cores'] = all_cores
            line['cpu_baseline'] = base
        ctypes.CDLL(None).fflush(None)  # whatever native code still holds in its stdout buffer comes BEFORE the line, never after
        emit(json.dumps(line))
        sys.stdout.flush()
    ranks.close()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50, help='launches per timed block')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--envs', type=int, default=65536, help='envs per GPU')
    ap.add_argument('--T', type=int, default=1000, help='env-steps per launch')
    ap.add_argument('--workload', default='c3', choices=['c2', 'c3', 'c4'])
    ap.add_argument('--min-seconds', type=float, default=0.5, help='repeat the K-launch block until this much time has been timed')
    ap.add_argument('--c4-envs', type=int, default=C4_TOTAL_ENVS, help='total envs of the strong-scaling config-4 line')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-checks', action='store_true', help='skip the reference-digest / oracle checks of this run')
    ap.add_argument('--no-strong-c4', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip BASELINE configs 2, 4 (one shard) and 5 reported beside `value` (1 GPU only)')
    ap.add_argument('--no-other-modes', action='store_true', help='skip the statistics-only / packed-row launches reported beside `value`')
    ap.add_argument('--gather-view', action='store_true', help='exercise the RCCL gathered view with one rank too')
    ap.add_argument('--single-process', action='store_true',
                    help='ONE process driving --gpus devices (one engine per device, gu_comm_init_all for the view) instead of one rank per GPU')
    ap.add_argument('--no-live-traffic', action='store_true',
                    help='do not measure roofline.traffic with two short rocprofv3 --pmc child runs (1 GPU only); use the committed profile')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def main(argv=None, engine_cls=None, script=None):
    """`engine_cls` / `script`: the CPU tests' entry (tests/_bench_stub.py) runs this very flow on the oracle-backed stub engine;
    `script` is what a plain --gpus N start re-launches as its ranks."""
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.pmc_child:
        pmc_child(args)
        return 0
    if args.single_process:
        run_single_process(args, engine_cls)
        return 0
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return spawn_ranks(args, argv, engine_cls, script)  # (before any HIP call and before libgu.so is loaded)
    run(args, engine_cls)
    return 0


if __name__ == '__main__':
    sys.exit(main())
