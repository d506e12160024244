"""bench.py --single-process: SURVEY.md 8(e)'s form.  ONE host process, one engine (handle + HIP stream) per device,
contiguous env-index shards with global env ids g * N .., every launch enqueued device after device so that the GPUs run
concurrently; the single-array view through ncclCommInitAll + one grouped ncclAllGather (gu_comm_init_all /
gu_allgather_view_all).  Timed like the multi-process form: blocks of exactly K launches PER DEVICE between device syncs of
all devices; `value` = all devices' env-steps / median block wall time."""
import time

import numpy as np

import griduniverse_amd as gua
from griduniverse_amd import _lib

from . import checks
from .configs import strong_c4_entry
from .cpu_leg import cpu_baseline, cpu_baseline_all_cores
from .launcher import ensure_library_is_current
from .report import device_block, emit_report, pacing_block, placement_block
from .timing import native_stdout_to_stderr, spread
from .traffic import read_traffic
from .workloads import BYTES_PER_ENV_STEP, HBM_COPY_GBPS, HBM_PEAK_GBPS, METRIC, WORKLOAD_SEED, build_workload, workload_line


def run_single_process(args, engine_cls=None, emit=print):
    """See the module docstring.  per_rank = every device's own HIP-event time."""
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
        found = {}
        if not args.no_checks:
            first = engines[0].read_trajectory(0, T)
            ref = checks.reference_digest(args.workload, template, seed, N, T, 0)
            found['bit_exact_vs_reference_digest'] = None if ref is None else checks.sha256_triplet(first) == ref
            ok = bool(checks.cpu_baseline_check_prefix(template, seed, 0, first))
            del first
            for g, e in enumerate(engines[1:], start=1):  # every other shard: its first envs against the oracle at ITS global ids
                ok = ok and bool(checks.cpu_baseline_check_prefix(template, seed, g * N, e.read_trajectory(0, T), n_check=512))
            found['bit_exact_vs_oracle'] = ok
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
            found['final_state_vs_oracle'] = checks.cpu_baseline_check_final_state(template, seed, 0, N, launches * T, engines[0].get_state())
            found['final_state_vs_oracle']['launches'] = launches
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
                            view_equals_shards=bool(equal), form='one process: ncclCommInitAll + one grouped ncclAllGather')
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
            shards_ok = all(bool(checks.cpu_baseline_check_prefix(template4, seed4, g * n, e.read_trajectory(0, T_check), n_check=n))
                            for g, e in enumerate(engines))
            ref = checks.reference_digest('c4', template4, seed4, n, T_check, 0) if G == 1 else None
            ref_ok = None if ref is None else checks.sha256_triplet(engines[0].read_trajectory(0, T_check)) == ref
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
        c4 = strong_c4_entry(args.c4_envs, n, G, T, K, wall4, kern4, desc4, seed4, shards_ok, ref_ok)
    elif not args.no_strong_c4:
        c4 = dict(skipped='%d envs do not divide over %d devices' % (args.c4_envs, G))

    w_min, w_med, w_max = spread(wall)
    per_dev_ms = [float(np.median([k[g] for k in kern])) / K for g in range(G)]
    worst = [max(k) for k in kern]
    k_min, k_med, k_max = spread(worst)
    launch_s = k_med / 1e3 / K
    achieved = BYTES_PER_ENV_STEP * N * T / launch_s / 1e9
    traffic = read_traffic('headline', launch_s * 1e3) if (N, T, args.workload) == (65536, 1000, 'c3') else None
    steps_per_block = float(G) * N * T * K
    detail = {
        'metric': METRIC, 'value': steps_per_block / w_med, 'unit': 'env-steps/s', 'n_gpus': G, 'steps': K, 'warmup': W,
        'ms_per_step': w_med / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int32',
        'data': 'synthetic',
        'config': {'workload': workload_line(args.workload, N, grid_desc, T), 'envs_per_gpu': N, 'env_steps_per_launch': T,
                   'global_envs': G * N, 'parallelism': 'env-index shards, no data-path collective; one host process, one engine per device',
                   'devices': devices},
        'mode': 'single-process',
        'timing': {'blocks': blocks, 'launches_per_block': K, 'timed_seconds': float(np.sum(wall)),
                   'ms_per_step_min': w_min / K * 1e3, 'ms_per_step_median': w_med / K * 1e3, 'ms_per_step_max': w_max / K * 1e3,
                   'value_min': steps_per_block / w_max, 'value_max': steps_per_block / w_min,
                   'launch_ms_min': k_min / K, 'launch_ms_median': k_med / K, 'launch_ms_max': k_max / K,
                   'launches_total': launches * G},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS,
                     'frac_wall': BYTES_PER_ENV_STEP * N * T / (w_med / K) / 1e9 / HBM_PEAK_GBPS,
                     'traffic': None if traffic is None else traffic.get('hbm_bytes_per_launch'), 'traffic_measured_by_child_runs': False,
                     'traffic_over_algorithmic': None if traffic is None else traffic.get('hbm_bytes_per_launch', 0.0) / float(BYTES_PER_ENV_STEP * N * T),
                     'kernel': 'gu_rollout_kernel<UNIFORM,TRAJ,LDS>', 'launch_ms': launch_s * 1e3,
                     'algorithmic_bytes_per_launch': BYTES_PER_ENV_STEP * N * T,
                     'traffic_source': None if traffic is None else traffic.get('source'),
                     'vs_measured_copy_rate': achieved / HBM_COPY_GBPS,
                     'store_pacing': pacing, 'trajectory_placement': placement},
        'device': dev_info,
        'engine': engine_cls.__module__ + '.' + engine_cls.__name__,
        'per_rank': {'ms_per_step': per_dev_ms, 'value': [float(N) * T / (ms / 1e3) for ms in per_dev_ms]},
        'rccl': rccl, 'strong_c4': c4, 'other_modes': None,
    }
    detail.update(found)
    if want_cpu:
        base = cpu_baseline(template, seed, T)
        base['all_cores'] = all_cores
        detail['cpu_baseline'] = base
    emit_report(detail, emit, args.detail)

