"""Round 3 on the device: per-engine options, the capped / process-aware trajectory placement search and what it reports,
gu_probe_trajectory, gu_device_info, several engines of one process on one device (LDS limits per device), and bench.py started
plainly with --gpus 2 and with --single-process (two ranks / two engines sharing the one GPU of the box)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import griduniverse_amd as gua
from griduniverse_amd import Engine, GridSpec, _lib
from oracle import c_oracle as C
from tests import _golden as G

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def spec_of(meta):
    return GridSpec(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'])


def test_options_are_per_engine_with_a_process_default(gu_option):
    """Two engines of one process on different kernel paths at the same time (the environment switches of rounds 1 / 2 could not do
    that), identical results; an engine's own value wins over the process default, None hands it back."""
    meta, _ = G.load_traj('c3_maze32')
    N, T = 4096, 300
    with Engine(N, spec_of(meta), seed=11) as a, Engine(N, spec_of(meta), seed=11) as b:
        assert a.get_option('rollout_rows') == -1 and a.get_option('rollout_block') == 256
        a.set_option('rollout_rows', 0)
        a.set_option('rollout_multi', 0)      # a: the general kernel for everything
        b.set_option('rollout_rows', 1)       # b: the transition-row kernel wherever eligible
        b.set_option('rollout_block', 512)
        gu_option('rollout_block', 128)       # process default: a uses it, b keeps its own 512
        assert a.get_option('rollout_block') == 128 and b.get_option('rollout_block') == 512
        outs = []
        for e in (a, b):
            e.reset()
            e.reserve_trajectory(T)
            e.rollout(T, 'uniform', True, True, stats=True)
            outs.append((e.read_trajectory(0, T), e.read_stats(), e.get_state()))
        for k in ('obs', 'reward', 'done'):
            assert np.array_equal(outs[0][0][k], outs[1][0][k]), k
        assert all(np.array_equal(x, y) for x, y in zip(outs[0][1], outs[1][1]))
        assert all(np.array_equal(outs[0][2][k], outs[1][2][k]) for k in outs[0][2])
        grid, st = C.Grid.from_lists(**meta), C.State(N)
        C.reset(grid, 11, st)
        want = C.rollout(grid, 11, st, T, True)
        assert all(np.array_equal(outs[0][0][k], want[k]) for k in want)
        b.set_option('rollout_block', None)
        assert b.get_option('rollout_block') == 128
        with pytest.raises(gua.GuError):
            a.set_option('x_traj_uncached', 1)  # compiled out of the product library


def test_placement_search_reports_every_candidate_and_later_engines_hold_less(gu_option):
    """gu_trajectory_placement_detail: per-candidate probe time and address, the kept index, the wall time and the peak bytes
    of the search; the search gives up after the back-to-back candidates when they are alike (no far phase then); a second
    engine of the process on the device probes at most four candidates and holds at most an eighth of the free memory; results
    never depend on the choice; gu_probe_trajectory re-times the kept buffer."""
    meta, _ = G.load_traj('c3_maze32')
    N, T = 65536, 400  # 3 x 105 MB planes = 315 MB: far candidates allowed (>= 256 MiB)
    grid = C.Grid.from_lists(**meta)
    gu_option('traj_candidates', 3)
    gu_option('traj_far_candidates', 2)
    gu_option('traj_stride_mib', 512)
    with Engine(N, spec_of(meta), seed=2) as first:
        first.reset()
        first.reserve_trajectory(T)
        d = first.trajectory_placement_detail()
        n, best, worst = first.trajectory_placement()
        assert n == len(d['probe_ms']) == len(d['address']) and 1 <= n <= 5 and 0 <= d['kept'] < n
        assert abs(d['probe_ms'][d['kept']] - best) < 1e-6 and abs(max(d['probe_ms']) - worst) < 1e-6 and best == min(d['probe_ms'])
        assert len(set(d['address'])) == n and d['search_ms'] > 0
        bytes_one = 3 * N * T * 4
        assert bytes_one <= d['peak_bytes'] <= 3 * bytes_one + 2 * (512 << 20) + 2 * bytes_one
        spread = (worst - best) / worst
        if n > 3:
            assert spread >= 0.06 - 1e-6 or best <= 0.86 * worst  # the far phase runs only where the first candidates showed two classes
        again = first.probe_trajectory()
        assert 0.5 * best < again < 2.0 * best
        first.rollout(T, 'uniform', True, True)
        st = C.State(2048)
        C.reset(grid, 2, st)
        want = C.rollout(grid, 2, st, T, True)
        got = first.read_trajectory(T - 1, 1)
        assert all(np.array_equal(got[k][0, :2048], want[k][T - 1]) for k in got)
        # a second engine of the same process on the same device: restricted search
        gu_option('traj_candidates', 12)
        gu_option('traj_far_candidates', 32)
        free = _lib.device_info(0)['hbm_free']
        with Engine(N, spec_of(meta), seed=2) as second:
            second.reset()
            second.reserve_trajectory(T)
            d2 = second.trajectory_placement_detail()
            assert 1 <= len(d2['probe_ms']) <= 4 and d2['peak_bytes'] <= max(bytes_one, free // 8 + bytes_one)
            second.rollout(T, 'uniform', True, True)
            got2 = second.read_trajectory(T - 1, 1)
            assert all(np.array_equal(got2[k], got[k]) for k in got)
    with Engine(64, spec_of(meta)) as small:  # small buffers are simply allocated
        small.reserve_trajectory(16)
        d = small.trajectory_placement_detail()
        assert d['probe_ms'] == [] and d['kept'] == -1 and d['peak_bytes'] == 3 * 64 * 16 * 4


def test_device_info_names_the_device():
    info = _lib.device_info(0)
    assert str(info['arch']).startswith('gfx950') and info['cus'] >= 1 and info['lds_per_cu'] >= 65536
    assert info['hbm_bytes'] > 2 ** 34 and 0 < info['hbm_free'] <= info['hbm_bytes'] and len(str(info['pci'])) >= 7
    assert Engine.device_info(0)['name'] == info['name']


def test_several_engines_of_one_process_with_large_lds_tables():
    """parallel.MultiDeviceVecGridUniverse on [0, 0, 0] (the box has one GPU; on a node the same code puts each engine on its own
    device, where the dynamic-LDS limit of a kernel has to be raised once PER DEVICE -- the latch is a per-device mask now):
    statistics-only launches on a 32x32 grid need 80 KiB of LDS for the K = 2 table; all shards against the oracle."""
    from griduniverse_amd.parallel import MultiDeviceVecGridUniverse
    meta, _ = G.load_traj('c3_maze32')
    total, T = 3 * 1024, 500
    env = MultiDeviceVecGridUniverse(total, [0, 0, 0], seed=5, auto_reset=True, template=spec_of(meta))
    try:
        env.reset()
        out = env.rollout(T, trajectory=False, stats=True)
        grid, st = C.Grid.from_lists(**meta), C.State(total)
        C.reset(grid, 5, st)
        want = C.rollout(grid, 5, st, T, True, trajectory=False, stats=True)
        assert np.array_equal(out['ret'], want['ret']) and np.array_equal(out['episodes'], want['episodes'])
        view = env.view()
        assert np.array_equal(view[0], st.pos) and np.array_equal(view[2], st.done)
    finally:
        env.close()


def _bench(*args):
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), cwd=ROOT, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=1200)
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


SMALL = ['--envs', '16384', '--T', '200', '--steps', '3', '--warmup', '1', '--min-seconds', '0.05', '--c4-envs', '16384', '--no-cpu-baseline', '--no-live-traffic']


def test_bench_started_plainly_with_two_ranks_on_the_one_gpu():
    """`python bench.py --gpus 2`, no launcher: two rank processes (sharing device 0 here), rank 0's one line; RCCL itself refuses
    two ranks on one device, which the line reports instead of dying."""
    line = _bench('--gpus', '2', *SMALL)
    assert line['n_gpus'] == 2 and len(line['per_rank']['value']) == 2 and line['engine'] == 'griduniverse_amd.engine.Engine'
    assert line['bit_exact_vs_oracle'] is True and line['final_state_vs_oracle']['equal'] is True
    assert line['rccl']['nranks'] == 2 and (line['rccl'].get('view_equals_shards') is True or 'error' in line['rccl'])
    assert line['strong_c4']['n_gpus'] == 2 and line['strong_c4']['shards_equal_oracle'] is True
    assert line['device']['arch'].startswith('gfx950') and line['roofline']['trajectory_placement'] is not None


def test_bench_single_process_form_on_the_one_gpu():
    """--single-process --gpus 2: one process, two engines (both on device 0 here), launches enqueued engine after engine; the
    gu_comm_init_all view needs one device per engine and is reported as refused on this box."""
    line = _bench('--gpus', '2', '--single-process', *SMALL)
    assert line['n_gpus'] == 2 and line['mode'] == 'single-process' and line['config']['devices'] == [0, 0]
    assert len(line['per_rank']['value']) == 2 and line['bit_exact_vs_oracle'] is True and line['final_state_vs_oracle']['equal'] is True
    assert line['rccl']['nranks'] == 2 and ('error' in line['rccl'] or line['rccl']['view_equals_shards'] is True)
    assert line['strong_c4']['shards_equal_oracle'] is True and len(line['roofline']['trajectory_placement']) == 2
    line = _bench('--gpus', '1', '--single-process', '--gather-view', *SMALL)
    assert line['n_gpus'] == 1 and line['rccl']['view_equals_shards'] is True and line['rccl']['nranks'] == 1


def test_store_pacing_is_calibrated_on_the_engines_own_state_and_never_changes_a_result(gu_option):
    """The rollout kernel rate-limits its int32-row store stream (every wave keeps a schedule on the 100 MHz clock; DESIGN.md
    section 6).  The period is calibrated by timing the kernel itself on the engine's own state, which is snapshot and put back: an
    engine that calibrated continues exactly where one that never did would -- trajectory, state, episode counters, done list;
    fixed periods (one the waves never meet, one they always wait for) give the same bytes too; launches too small to be bound by
    HBM are neither paced nor calibrated."""
    meta, _ = G.load_traj('c3_maze32')
    N, T = 65536, 300  # 236 MB of rows per launch: paced
    outs = {}
    for pace in (0, None, 20, 400):
        gu_option('rollout_pace', pace)
        with Engine(N, spec_of(meta), seed=9) as eng:
            eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T // 3, 'uniform', True, True)       # a shorter launch first (under the 128 MB bar: never paced)
            assert eng.rollout_pacing() is None
            eng.rollout(T, 'uniform', True, True, stats=True)  # (calibrates when pace is None)
            info = eng.rollout_pacing()
            if pace is None:
                assert info is not None and info['evaluated'] >= 8 and info['ms_unpaced'] > 0 and 0 < info['ms_paced'] <= info['ms_unpaced'] * 1.001
                assert 0 <= info['period'] <= 4000 and info['calibration_ms'] > 0
            else:
                assert info is None  # a fixed amount: nothing to calibrate
            eng.rollout(T, 'uniform', True, True, stats=True)
            tr = eng.read_trajectory(0, T)
            st = eng.get_state()
            outs[pace] = (tr['obs'], tr['reward'], tr['done'], st['pos'], st['done'], st['episode'], st['tcount'], eng.read_stats()[0], eng.done_indices())
    for pace in (None, 20, 400):
        assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[pace])), pace
    grid, st = C.Grid.from_lists(**meta), C.State(2048)
    C.reset(grid, 9, st)
    C.rollout(grid, 9, st, T // 3 + T, True, trajectory=False)
    want = C.rollout(grid, 9, st, T, True)
    assert all(np.array_equal(outs[None][i][:, :2048], want[k]) for i, k in enumerate(('obs', 'reward', 'done')))
    # a batch whose last workgroup is ragged, with and without a schedule
    N2 = 65536 + 100
    ragged = {}
    for pace in (0, 150, 60):
        gu_option('rollout_pace', pace)
        with Engine(N2, spec_of(meta), seed=9, env_id0=1000) as eng:
            eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T, 'uniform', True, True, stats=True)
            eng.rollout(T // 2, 'uniform', True, True, stats=True)
            tr = eng.read_trajectory(0, T // 2)
            st = eng.get_state()
            ragged[pace] = (tr['obs'], tr['reward'], tr['done'], st['pos'], st['episode'], eng.read_stats()[0], eng.read_stats()[1], eng.done_indices())
    for pace in (150, 60):
        assert all(np.array_equal(a, b) for a, b in zip(ragged[0], ragged[pace])), pace
    # launches of fewer than 64 steps and batches of more than four waves per SIMD keep no schedule (nothing to gain there)
    cus = Engine.device_info(0)['cus']
    gu_option('rollout_pace', None)
    for n_big, t_big in ((65536, 48), (cus * 1024 + 256, 64)):
        with Engine(n_big, spec_of(meta), seed=9) as eng:
            eng.reset()
            eng.reserve_trajectory(t_big)
            eng.rollout(t_big, 'uniform', True, True)
            assert eng.rollout_pacing() is None, (n_big, t_big)
    # a caller-supplied stream and a table policy are calibrated as launch kinds of their own
    gu_option('rollout_pace', None)
    with Engine(N, spec_of(meta), seed=9) as eng:
        eng.reset()
        eng.reserve_trajectory(T)
        S = meta['W'] * meta['H']
        eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
        eng.rollout(T, 'sample', False, True)
        assert eng.rollout_pacing('sample', False) is not None and eng.rollout_pacing('uniform', True) is None


def test_pair_tables_leave_the_same_rows_as_the_one_step_table_and_the_general_kernel(gu_option):
    """gu_rollout_rows.hip's pair tables (two env-steps per LDS round trip; uniform policy and caller-supplied streams, launches that
    write rows, one workgroup per CU at most): every launch shape through the pair tables (option rollout_rows = 1), the one-step
    table (= 2) and the general kernel (= 0) -- int32 and packed rows, with and without auto-reset, step counts that leave heads and
    tails around the 16-step action words, two launches in a row (the second starts inside a word), per-env statistics, final state."""
    meta, _ = G.load_traj('c4_lava32')
    rs = np.random.RandomState(5)
    cases = [(4096, 1000, 'uniform'), (4100, 777, 'uniform'), (2048, 17, 'uniform'), (3000, 33, 'stream'), (4096, 250, 'stream')]
    for N, T, policy in cases:
        acts = rs.randint(0, 4, size=(T, N)).astype(np.int32) if policy == 'stream' else None
        for auto in (True, False):
            for traj in (True, 'packed'):
                outs = {}
                for rows in (0, 1, 2):
                    gu_option('rollout_rows', rows)
                    with Engine(N, spec_of(meta), seed=21, env_id0=7) as eng:
                        eng.reset()
                        if acts is not None:
                            eng.upload_actions(acts)
                        eng.reserve_trajectory(T)
                        eng.rollout(T // 3 + 1, policy, auto, trajectory=traj, stats=True)  # the next launch starts inside an action word
                        if acts is not None:
                            eng.upload_actions(acts)
                        eng.rollout(T, policy, auto, trajectory=traj, stats=True)
                        tr = eng.read_trajectory(0, T) if traj is True else eng.read_trajectory_packed(0, T)
                        st = eng.get_state()
                        outs[rows] = [tr[k] for k in sorted(tr)] + [st[k] for k in sorted(st)] + list(eng.read_stats()) + [eng.done_indices()]
                for rows in (1, 2):
                    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[rows])), (N, T, policy, auto, traj, rows)
    gu_option('rollout_rows', None)
    # the default dispatch at a config-4 shard (32 768 envs, int32 rows) and for packed rows at 65 536 envs is the pair path: oracle
    grid = C.Grid.from_lists(**meta)
    for N, traj in ((32768, True), (65536, 'packed')):
        T = 200
        with Engine(N, spec_of(meta), seed=4) as eng:
            eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T, 'uniform', True, trajectory=traj)
            got = eng.read_trajectory(0, T) if traj is True else eng.read_trajectory_packed(0, T, unpack=True)
        st = C.State(2048, N - 2048)
        C.reset(grid, 4, st)
        want = C.rollout(grid, 4, st, T, True)
        for k in ('obs', 'reward', 'done'):
            assert np.array_equal(got[k][:, N - 2048:], want[k]), (N, traj, k)
