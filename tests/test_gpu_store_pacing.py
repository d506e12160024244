"""The rate limiter of the trajectory store stream (gu_rollout.hpp: GuPacer): calibration on the engine's own state, when the search runs, the process-wide cache of periods."""
import numpy as np
import pytest

from griduniverse_amd import Engine, GridSpec, _lib
from oracle import c_oracle as C
from tests import _golden as G

pytestmark = pytest.mark.gpu

def spec_of(meta):
    return GridSpec(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'])


def test_store_pacing_is_calibrated_on_the_engines_own_state_and_never_changes_a_result(gu_option):
    """The rollout kernel rate-limits its int32-row store stream (every wave keeps a schedule on the 100 MHz clock; DESIGN.md
    section 6).  The period is calibrated by timing the kernel itself on the engine's own state, which is snapshot and put back: an
    engine that calibrated continues exactly where one that never did would -- trajectory, state, episode counters, done list;
    fixed periods (one the waves never meet, one they always wait for) give the same bytes too; launches too small to be bound by
    HBM are neither paced nor calibrated."""
    meta, _ = G.load_traj('c3_maze32')
    N, T = 65536, 300  # 236 MB of rows per launch: paced
    outs = {}
    for pace in (0, 'lazy', 'eager', 'explicit', 20, 400):
        gu_option('rollout_pace', {'lazy': None, 'explicit': None, 'eager': -2}.get(pace, pace))
        with Engine(N, spec_of(meta), seed=9) as eng:
            eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T // 3, 'uniform', True, True)       # a shorter launch first (under the 128 MB bar: never paced)
            assert eng.rollout_pacing() is None
            if pace == 'explicit':
                eng.calibrate_rollout(T, 'uniform', True, True, stats=True)  # = rollout, with the search made now
            else:
                eng.rollout(T, 'uniform', True, True, stats=True)  # (searches when pace is -2: round 3's behaviour)
            info = eng.rollout_pacing()
            totals = eng.rollout_pacing_totals()
            if pace in ('eager', 'explicit'):
                # (the first of the two searches; the second engine of this shape only CHECKS the period the first one found)
                assert info is not None and info['ms_unpaced'] > 0 and 0 < info['ms_paced'] <= info['ms_unpaced'] * 1.001
                assert 0 <= info['period'] <= 4000 and info['calibration_ms'] > 0 and totals['kinds_paced'] == 1
                assert totals['launches_spent'] >= 6 and totals['kinds_from_cache'] in (0, 1)
            else:
                assert info is None  # a fixed amount, or the default: no limiter until 1024 launches of the kind have been issued
                assert totals['launches_spent'] == 0 and totals['kinds_paced'] == 0 and totals['kinds_waiting'] == (1 if pace == 'lazy' else 0)
            eng.rollout(T, 'uniform', True, True, stats=True)
            tr = eng.read_trajectory(0, T)
            st = eng.get_state()
            outs[pace] = (tr['obs'], tr['reward'], tr['done'], st['pos'], st['done'], st['episode'], st['tcount'], eng.read_stats()[0], eng.done_indices())
    for pace in ('lazy', 'eager', 'explicit', 20, 400):
        assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[pace])), pace
    outs[None] = outs['eager']
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
    gu_option('rollout_pace', -2)
    for n_big, t_big in ((65536, 48), (cus * 1024 + 256, 64)):
        with Engine(n_big, spec_of(meta), seed=9) as eng:
            eng.reset()
            eng.reserve_trajectory(t_big)
            eng.rollout(t_big, 'uniform', True, True)
            assert eng.rollout_pacing() is None, (n_big, t_big)
    # a caller-supplied stream and a table policy are calibrated as launch kinds of their own
    gu_option('rollout_pace', -2)
    with Engine(N, spec_of(meta), seed=9) as eng:
        eng.reset()
        eng.reserve_trajectory(T)
        S = meta['W'] * meta['H']
        eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
        eng.rollout(T, 'sample', False, True)
        assert eng.rollout_pacing('sample', False) is not None and eng.rollout_pacing('uniform', True) is None


def test_the_pacing_search_is_paid_once_per_process_and_only_when_it_can_pay(gu_option):
    """Round 3 charged every engine ~100 full-size launches the first time a launch kind ran.  Now: a fresh engine's first rollout
    at the headline size costs a kernel, not a search; the search runs on request (calibrate_rollout) and its result is kept for
    the process: the second engine of the same shape CHECKS the period with six launches in a few milliseconds."""
    import time
    meta, _ = G.load_traj('c3_maze32')
    N, T = 65536, 1000
    gu_option('rollout_pace', None)
    with Engine(N, spec_of(meta), seed=3) as eng:
        eng.reset()
        eng.reserve_trajectory(T)
        eng.sync()
        t0 = time.perf_counter()
        eng.rollout(T, 'uniform', True, True)
        eng.sync()
        first_ms = (time.perf_counter() - t0) * 1e3
        assert first_ms < 3.0 and eng.rollout_pacing_totals()['launches_spent'] == 0  # (a 0.12 ms kernel + what a first launch of a kernel costs)
        eng.calibrate_rollout(T, 'uniform', True, True)
        found = eng.rollout_pacing()
        totals = eng.rollout_pacing_totals()
        assert found is not None and totals['launches_spent'] >= 6 and totals['kinds_paced'] == 1
    for _ in range(2):  # later engines of the same shape: six launches, < 5 ms, the same period
        with Engine(N, spec_of(meta), seed=4) as eng:
            eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T, 'uniform', True, True)
            totals = eng.rollout_pacing_totals()
            info = eng.rollout_pacing()
            assert totals['launches_spent'] == 6 and totals['calibration_ms'] < 5.0, totals
            if info is not None:  # (kept: it still beat no limiter on this engine's buffer)
                assert info['period'] == found['period'] and totals['kinds_from_cache'] == 1
            else:
                assert totals['kinds_waiting'] == 1
