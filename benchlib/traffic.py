"""roofline.traffic of bench.py: HBM bytes per launch from rocprofv3 --pmc passes (MI355X_MICROARCH.md, HBM section:
WRITE_SIZE and FETCH_SIZE in separate passes, counters only; both in KiB; FETCH_SIZE x 2 on gfx950)."""
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import griduniverse_amd as gua

from .workloads import WORKLOAD_SEED, build_workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_SCRIPT = os.path.join(ROOT, 'bench.py')


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
            out['note'] = 'kernel duration differs from the profiled run by %+.1f %%' % ((ratio - 1.0) * 100)
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
            cmd = [tool, '--pmc', counter, '--output-format', 'csv', '-d', out, '--', sys.executable, BENCH_SCRIPT,
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
                source='rocprofv3 --pmc WRITE_SIZE, FETCH_SIZE: two child runs of bench.py --pmc-child on this device after the timed region')


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

