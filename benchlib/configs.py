"""What bench.py reports BESIDE `value`, never as it: BASELINE.json configs 2, 4 (one shard of eight) and 5, the launch
forms that do not stream 12 bytes per env-step, config 4 as a strong-scaling line and the RCCL gathered view.
Every entry is numbers plus its own parity bit; the bounds each one sits under are argued in DESIGN.md section 5."""
import hashlib
import time

import numpy as np

import griduniverse_amd as gua
from griduniverse_amd import _lib

from . import checks
from .timing import launch_ms, settle_launches, spread, timed_region
from .workloads import BYTES_PER_ENV_STEP, C4_TOTAL_ENVS, HBM_PEAK_GBPS, WORKLOAD_SEED, build_workload


def _rows_entry(N, T, ms, bytes_per_step=BYTES_PER_ENV_STEP):
    gbps = bytes_per_step * float(N) * T / ms / 1e6
    return dict(us_per_launch=ms * 1e3, env_steps_per_s=float(N) * T / ms * 1e3, hbm_gbps=gbps, frac_of_hbm_peak=gbps / HBM_PEAK_GBPS)


def baseline_configs(engine_cls, device, K, check, only=None):
    """Configs 2, 4 (shard 1 of 8) and 5 on this GPU: us per launch (or per round), env-steps/s, fraction of the HBM peak
    where rows are written, and `bit_exact` (the config's own check against the reference digest or the C oracle)."""
    out = {}
    T = 1000
    if only:  # (tools: a subset of the entries, by prefix)
        full = baseline_configs(engine_cls, device, K, check) if not all(o.startswith('c3_distinct') for o in only) else _distinct_grids(engine_cls, device, K, check, T)
        return {k: v for k, v in full.items() if any(k.startswith(o) for o in only)}

    # ---- config 2: 4096 envs, default 8x8 grid.  Latency-bound: 64 waves on 1024 SIMDs, a chain of T dependent steps.
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
            ref = checks.reference_digest('c2', template, seed, N, T, 0)
            ok = checks.sha256_triplet(got) == ref if ref is not None else bool(checks.cpu_baseline_check_prefix(template, seed, 0, got))
            del got
        ms = launch_ms(eng, T, K, trajectory=True)
    finally:
        eng.close()
    out['c2'] = dict(_rows_entry(N, T, ms), workload='c2: %d envs, %s, seed %d, %d steps per launch' % (N, desc, seed, T),
                     bound='latency', bit_exact=ok, check='first launch == reference digest c2_open8x8_4096x1000')

    # ---- config 4, one shard of eight: 32 768 envs with global ids 32768 .. 65535 on the lava grid
    template, desc = build_workload('c4')
    N, seed = C4_TOTAL_ENVS // 8, WORKLOAD_SEED['c4']
    eng = engine_cls(N, gua.GridSpec.from_env(template), device=device, env_id0=N, seed=seed)
    try:
        eng.reset()
        eng.reserve_trajectory(4 * T)
        eng.rollout(250, 'uniform', auto_reset=True, trajectory=True)
        ok = bool(checks.cpu_baseline_check_prefix(template, seed, N, eng.read_trajectory(0, 250), n_check=N)) if check else None
        ms = launch_ms(eng, T, K, trajectory=True)
        # fixed cost and slope of a launch: T against 4 T, the two lengths alternating, the best of three each (the store pacing
        # starts over whenever the length changes by more than a factor of two: launch_ms settles it every time)
        short, long_ = [ms], []
        for rep in range(3):
            long_.append(launch_ms(eng, 4 * T, max(4, K // 2), trajectory=True))
            if rep < 2:
                short.append(launch_ms(eng, T, K, trajectory=True))
        ms_long, ms_short = min(long_), min(short)
    finally:
        eng.close()
    slope_us = (ms_long - ms_short) * 1e3 / (3 * T)  # us per env-step row of the shard, fixed cost of a launch removed
    out['c4_shard'] = dict(_rows_entry(N, T, ms), workload='c4 shard 1 of 8: %d envs (ids %d..), %s, seed %d' % (N, N, desc, seed),
                           bound='hbm + fixed cost per launch', bit_exact=ok, check='first launch (250 steps): whole shard == C oracle',
                           us_per_launch_4000_steps=ms_long * 1e3, fixed_us_per_launch=ms_short * 1e3 - slope_us * T,
                           asymptote_frac_of_hbm_peak=BYTES_PER_ENV_STEP * N / slope_us / 1e3 / HBM_PEAK_GBPS if slope_us > 0 else None)

    out.update(_distinct_grids(engine_cls, device, K, check, T))

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
                ok = checks.cpu_baseline_check_c5(template, seed, gamma, n_check, v, pi, eng.get_state(), eng.read_outputs()[1])
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
        out['c5'] = dict(workload='c5: %d envs, %s, gamma %.1f, %d rounds of (V1 + V2 sweep, float64) + one greedy step in ONE launch'
                                  % (N, desc, gamma, rounds),
                         us_per_round=us, env_steps_per_s=N / us * 1e6, state_updates_per_s=S / us * 1e6,
                         form={1: 'per-XCD', 2: 'chip-wide barrier', 3: 'launch per round'}.get(form, str(form)), bound='latency',
                         timing='host wall time of one call / rounds, median of 3', bit_exact=ok,
                         check='%d rounds from reset: tables as raw bytes + every env == C oracle' % n_check)
    return out


def _distinct_grids(engine_cls, device, K, check, T):
    # ---- config 3 with DISTINCT grids (SURVEY.md 8(d) C3 variant; N x GridUniverseEnv(random_maze=True), griduniverse_env.py:318-321):
    # 65 536 envs on G device-generated 32x32 mazes (8(f3)), grid = env // (N / G).  G = 65 536 is one maze per env: every lane keeps
    # its own grid in LDS at four bits per cell (gu_rollout.hpp, MAP 5).
    out = {}
    if hasattr(engine_cls, 'generate_mazes'):
        N, seed, maze_seed = 65536, WORKLOAD_SEED['c3'], 2026
        for G in (1024, 65536):
            eng = engine_cls(N, gua.GridSpec(32, 32, [0], [1023], [], []), device=device, env_id0=0, seed=seed)
            try:
                eng.generate_mazes(G, 32, 32, maze_seed)
                first = eng.reset()
                eng.reserve_trajectory(T)
                eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
                ok = None
                if check:  # a sample of grids, each against the C oracle on ITS maze (oracle/gu_oracle.c: the same build RNG)
                    ok = checks.cpu_baseline_check_distinct_grids(eng.read_trajectory(0, T), first, N, G, T, seed, maze_seed)
                ms = launch_ms(eng, T, K, trajectory=True)
            finally:
                eng.close()
            out['c3_distinct_%d' % G] = dict(_rows_entry(N, T, ms), workload='c3 on %d distinct device-generated 32x32 mazes (%d envs each), seed %d'
                                             % (G, N // G, seed), bound='hbm', bit_exact=ok,
                                             check='first launch: envs of six grids == C oracle on the same mazes')
    return out


def other_modes(eng, template, seed, env_id0, N, T, K, check):
    """The headline workload in the launch forms that do not write int32 rows from uniform actions: per-env statistics only
    (no HBM stream), the sampled table policy (rows / statistics), one packed uint32 per env-step (4 B)."""
    out = {}
    eng.seed(seed)
    eng.reset()
    eng.rollout(T, 'uniform', auto_reset=True, trajectory=False, stats=True)
    ret, episodes = eng.read_stats()
    ok = checks.cpu_baseline_check_stats(template, seed, env_id0, T, ret, episodes) if check else None
    ms = launch_ms(eng, T, K, trajectory=False, stats=True)
    out['stats_only'] = {'ms_per_launch': ms, 'value': float(N) * T / ms * 1e3, 'unit': 'env-steps/s (this rank)',
                         'returns_vs_oracle': ok, 'mean_return_per_env': float(np.mean(ret))}
    if hasattr(eng, 'vi_set'):  # the sampled table policy (the producer of Monte-Carlo evaluation): actions ~ pi[s] on RNG stream 2
        S = template.world.size
        eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
        ms = launch_ms(eng, T, K, 'sample', trajectory=True)
        gbps = BYTES_PER_ENV_STEP * float(N) * T / ms / 1e6
        out['rollout_sample_policy_traj'] = {'ms_per_launch': ms, 'value': float(N) * T / ms * 1e3, 'unit': 'env-steps/s (this rank)',
                                             'bytes_per_env_step': BYTES_PER_ENV_STEP, 'achieved_GBps': gbps,
                                             'frac_of_hbm_peak': gbps / HBM_PEAK_GBPS}
        ms = launch_ms(eng, T, K, 'sample', trajectory=False, stats=True)
        out['rollout_sample_policy_stats_only'] = {'ms_per_launch': ms, 'value': float(N) * T / ms * 1e3, 'unit': 'env-steps/s (this rank)'}
    if hasattr(eng, 'read_trajectory_packed'):
        ms = launch_ms(eng, T, K, trajectory='packed')
        out['packed_rows'] = {'ms_per_launch': ms, 'value': float(N) * T / ms * 1e3, 'unit': 'env-steps/s (this rank)',
                              'bytes_per_env_step': 4, 'achieved_GBps': 4.0 * N * T / ms / 1e6,
                              'frac_of_hbm_peak': 4.0 * N * T / ms / 1e6 / HBM_PEAK_GBPS}
    return out


def rccl_view_check(eng, engine_cls, ranks):
    """The single-array (obs, reward, done) view over RCCL, outside the timed region: one ncclAllGather of every rank's
    packed int32[3N] block.  Every rank's own shard digest travels over the host channel and is compared with the digest of
    that rank's slice of the RCCL view, on every rank."""
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
    equal = ranks.reduce([1.0 if equal else 0.0], 'MIN')[0] == 1.0
    lap = ranks.reduce([float(np.median(laps))], 'MAX')[0]
    eng.comm_destroy()
    return dict(nranks=world, comm_init_ms=ranks.reduce([init_ms], 'MAX')[0], allgather_ms=lap, bytes_per_rank=3 * n * 4,
                view_envs=world * n, view_equals_shards=bool(equal), form='one rank per process: ncclCommInitRank + ncclAllGather')


def strong_c4(args, ranks, engine_cls, device, pacing_block=None):
    """BASELINE.json config 4: 262 144 envs on the 32x32 lava grid IN TOTAL, sharded over the ranks by env index (strong
    scaling), seed 4.  One checked launch (250 steps from reset: every rank compares its shard with the C oracle; on one GPU
    the whole batch is also hashed against the reference's digest), then timed blocks of K launches of T steps."""
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
        shard_ok = checks.cpu_baseline_check_prefix(template, seed, rank * n, got, n_check=n)
        shards_ok = ranks.reduce([1.0 if shard_ok else 0.0], 'MIN')[0] == 1.0
        ref = checks.reference_digest('c4', template, seed, n, T_check, rank * n) if world == 1 else None
        ref_ok = None if ref is None else checks.sha256_triplet(got) == ref
        del got
        for _ in range(max(args.warmup, settle_launches(eng, args.T, 'uniform', trajectory=True))):
            eng.rollout(args.T, 'uniform', auto_reset=True, trajectory=True)
        wall, kern, _, _ = timed_region(eng, ranks, args.T, args.steps, args.min_seconds / 2)
        pacing = pacing_block(eng) if pacing_block else None
    finally:
        eng.close()
    return strong_c4_entry(total, n, world, args.T, args.steps, wall, kern, desc, seed, shards_ok, ref_ok, pacing)


def strong_c4_entry(total, n, world, T, K, wall, kern, desc, seed, shards_ok, ref_ok, pacing=None):
    w_min, w_med, w_max = spread(wall)
    k_med = spread(kern)[1]
    return dict(value=float(total) * T * K / w_med, unit='env-steps/s', scaling='strong', total_envs=total, envs_per_gpu=n,
                n_gpus=world, env_steps_per_launch=T, steps=K, blocks=len(wall), ms_per_step=w_med / K * 1e3,
                ms_per_step_min=w_min / K * 1e3, ms_per_step_max=w_max / K * 1e3, launch_ms=k_med / K,
                hbm_gbps_per_gpu=BYTES_PER_ENV_STEP * n * T / (k_med / K / 1e3) / 1e9,
                workload='c4: %s, seed %d, uniform device-RNG actions, auto-reset, int32 rows' % (desc, seed),
                shards_equal_oracle=bool(shards_ok), bit_exact_vs_reference_digest=ref_ok, store_pacing=pacing,
                check='first launch (250 steps from reset): every shard == C oracle'
                      + ('; whole batch sha256 == reference digest c4_lava32_262144x250' if ref_ok is not None else ''))
