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

Output: ONE JSON line on stdout, under 4 KB (benchlib/report.py: contract keys, config, roofline, cpu_baseline, parity
bits, other figures as numbers); everything else goes to the side file --detail names (bench_detail.json in the cwd).

N > 1: after the timed region the RCCL gathered view runs once and is checked against every rank's own shard ("rccl"),
and config 4 -- 262 144 envs on the lava grid in total, split over the ranks -- is timed as a strong-scaling line
("strong_c4"), its result checked against the C oracle on every rank (and, on one GPU, against the reference digest).
"""
import argparse
import os
import sys
import threading

import numpy as np

# RCCL between processes needs dmabuf IPC on this driver stack (the pool exports this already; harmless when it is set)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd import _lib  # noqa: E402
from benchlib import checks  # noqa: E402
from benchlib.checks import reference_digest  # noqa: E402,F401  (tests/test_multiprocess.py reads these through `bench`)
from benchlib.configs import baseline_configs, other_modes, rccl_view_check, strong_c4  # noqa: E402
from benchlib.cpu_leg import cpu_baseline, cpu_baseline_all_cores  # noqa: E402
from benchlib.launcher import ensure_library_is_current, spawn_ranks  # noqa: E402
from benchlib.report import device_block, emit_report, pacing_block, placement_block, topology_block  # noqa: E402
from benchlib.single_process import run_single_process  # noqa: E402
from benchlib.timing import Ranks, native_stdout_to_stderr, spread, timed_region  # noqa: E402,F401
from benchlib.traffic import live_traffic, pmc_child, read_traffic  # noqa: E402
from benchlib.workloads import (BYTES_PER_ENV_STEP, C4_TOTAL_ENVS, HBM_COPY_GBPS, HBM_PEAK_GBPS, METRIC,  # noqa: E402,F401
                                REFERENCE_DIGEST, WORKLOAD_SEED, build_workload, workload_line)


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
    found = {}
    if rank == 0 and not args.no_checks:
        first = eng.read_trajectory(0, T)
        ref = checks.reference_digest(args.workload, template, seed, N, T, rank * N)
        found['bit_exact_vs_reference_digest'] = None if ref is None else checks.sha256_triplet(first) == ref
        found['reference_digest'] = None if ref is None else REFERENCE_DIGEST[args.workload] + ' (tests/golden/digests.json)'
        found['bit_exact_vs_oracle'] = bool(checks.cpu_baseline_check_prefix(template, seed, rank * N, first))
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
        found['final_state_vs_oracle'] = checks.cpu_baseline_check_final_state(template, seed, rank * N, N, launches * T, eng.get_state())
        found['final_state_vs_oracle']['launches'] = launches

    others = None
    if not args.no_other_modes and hasattr(eng, 'read_stats'):
        others = other_modes(eng, template, seed, rank * N, N, T, K, rank == 0 and not args.no_checks)

    per_rank = ranks.gather([float(np.median(own_wall))])
    placement = placement_block(eng, post_probe_ms, float(np.median(kern)) / K)
    pacing = pacing_block(eng)
    c4 = None if args.no_strong_c4 else strong_c4(args, ranks, engine_cls, device, pacing_block)
    configs = None
    if world == 1 and not args.no_configs:
        configs = baseline_configs(engine_cls, device, K, not args.no_checks)

    if rank == 0:
        w_min, w_med, w_max = spread(wall)
        k_min, k_med, k_max = spread(kern)
        launch_s = k_med / 1e3 / K
        achieved = BYTES_PER_ENV_STEP * N * T / launch_s / 1e9
        profiled_size = (N, T, args.workload) == (65536, 1000, 'c3')  # (the committed PMC profile is of the default launch only)
        traffic = read_traffic('headline', launch_s * 1e3) if profiled_size else None
        measured = None
        if world == 1 and engine_cls is gua.Engine and not args.no_live_traffic:
            measured = live_traffic(args, N, T)
        if c4 and 'launch_ms' in c4 and profiled_size and args.c4_envs == C4_TOTAL_ENVS and world == 1:
            c4['traffic'] = read_traffic('strong_c4', c4['launch_ms'])
        for mode in (others or {}) if profiled_size else ():
            others[mode]['traffic'] = read_traffic(mode, others[mode]['ms_per_launch'])
        steps_per_block = float(world) * N * T * K
        hbm_bytes = measured['hbm_bytes_per_launch'] if measured else None if traffic is None else traffic.get('hbm_bytes_per_launch')
        detail = {
            'metric': METRIC, 'value': steps_per_block / w_med, 'unit': 'env-steps/s', 'n_gpus': world,
            'steps': K, 'warmup': W, 'ms_per_step': w_med / K * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'int32', 'data': 'synthetic',
            'config': {'workload': workload_line(args.workload, N, grid_desc, T), 'envs_per_gpu': N, 'env_steps_per_launch': T,
                       'global_envs': world * N, 'parallelism': 'env-index shards, no data-path collective'},
            # value = median block; a block = exactly K launches between barrier + device sync pairs, max over ranks
            'timing': {'blocks': blocks, 'launches_per_block': K, 'timed_seconds': float(np.sum(wall)),
                       'ms_per_step_min': w_min / K * 1e3, 'ms_per_step_median': w_med / K * 1e3, 'ms_per_step_max': w_max / K * 1e3,
                       'value_min': steps_per_block / w_max, 'value_max': steps_per_block / w_min,
                       'launch_ms_min': k_min / K, 'launch_ms_median': k_med / K, 'launch_ms_max': k_max / K,
                       'per_launch_ms_min': float(per_launch.min()), 'per_launch_ms_median': float(np.median(per_launch)),
                       'per_launch_ms_max': float(per_launch.max()), 'launches_total': launches},
            # frac: algorithmic bytes / the kernels' own HIP-event time per launch; frac_wall: / ms_per_step (the driver's clock)
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS,
                         'frac_wall': BYTES_PER_ENV_STEP * N * T / (w_med / K) / 1e9 / HBM_PEAK_GBPS,
                         'traffic': hbm_bytes, 'traffic_measured_by_child_runs': bool(measured),
                         'traffic_over_algorithmic': hbm_bytes / float(BYTES_PER_ENV_STEP * N * T) if hbm_bytes else None,
                         'kernel': 'gu_rollout_kernel<UNIFORM,TRAJ,LDS>', 'launch_ms': launch_s * 1e3,
                         # the same kernel's average duration in the committed rocprofv3 --kernel-trace summary (profiles/), for comparison
                         'kernel_avg_us': None if traffic is None else traffic.get('kernel_avg_us'),
                         'kernel_avg_us_profile': None if traffic is None else traffic.get('tag'),
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
            # every rank's own median block (the N = 1 run of this script reports exactly this figure as `value`)
            'per_rank': {'ms_per_step': [v[0] / K * 1e3 for v in per_rank], 'value': [float(N) * T * K / v[0] for v in per_rank]},
            'rccl': None, 'strong_c4': c4, 'other_modes': others, 'configs': configs,
            'topology': topology_block(engine_cls),
        }
        detail.update(found)
        if want_cpu:
            base = cpu_baseline(template, seed, T)
            base['all_cores'] = all_cores
            detail['cpu_baseline'] = base
    # ---- LAST: the RCCL gathered view (never run with more than one rank before the driver's own multi-GPU run).  Nothing follows
    # it but the line, and a watchdog stands behind it: a collective that does not come back costs the `rccl` object, not the line.
    if world > 1 or args.gather_view:
        box = {}

        def check():
            try:
                box['rccl'] = rccl_view_check(eng, engine_cls, ranks)
            except gua.GuError as err:  # reported, not fatal: the throughput line does not depend on the collective
                box['rccl'] = dict(nranks=world, view_equals_shards=None, error=str(err))
            except Exception as err:  # noqa: BLE001 -- (a peer that left the host channel while this rank was still in the check)
                box['rccl'] = dict(nranks=world, view_equals_shards=None, error='%s: %s' % (type(err).__name__, err))

        limit = float(os.environ.get('GU_RCCL_CHECK_TIMEOUT', '180'))
        worker = threading.Thread(target=check, daemon=True)
        with native_stdout_to_stderr():  # (RCCL's banner; restored HERE, by this thread, whatever becomes of the worker)
            worker.start()
            worker.join(limit)
        if worker.is_alive():  # the thread sits in native code: this process can only report and leave
            if rank == 0:
                detail['rccl'] = dict(nranks=world, view_equals_shards=None, error='the RCCL view check did not come back within %.0f s' % limit)
                emit_report(detail, emit, args.detail)
            sys.stdout.flush()
            sys.stderr.write('bench.py: rank %d: the RCCL view check did not come back within %.0f s; leaving\n' % (rank, limit))
            os._exit(0)
        if rank == 0:
            detail['rccl'] = box.get('rccl')
    if rank == 0:
        emit_report(detail, emit, args.detail)
    eng.close()
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
    ap.add_argument('--detail', default='bench_detail.json', help="side file for everything that is not on the line ('' = none)")
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
        return spawn_ranks(args, argv, script or os.path.abspath(__file__), engine_cls)  # (before any HIP call and before libgu.so is loaded)
    run(args, engine_cls)
    return 0


if __name__ == '__main__':
    sys.exit(main())
