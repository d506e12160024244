"""The rate limiter of the trajectory store stream (gu_rollout.hpp: GuPacer): the launches of a kind choose their period themselves, closed
loop, on the device -- from the first launch on, without a search, a dedicated launch or a stall, and as fast as the best FIXED period around
its model; results never depend on any of it."""
import time

import numpy as np
import pytest

from griduniverse_amd import Engine, GridSpec
from oracle import c_oracle as C
from tests import _golden as G

import os

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(bool(os.environ.get('GU_TEST_OPTIONS')), reason='about the DEFAULT dispatch and pacing: not under forced launch-shape options')]


def spec_of(meta):
    return GridSpec(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'])


def test_store_pacing_never_changes_a_result(gu_option):
    """The rollout kernel rate-limits its int32-row store stream (every wave keeps a schedule on the 100 MHz clock; DESIGN.md
    section 6).  Whatever the period -- the closed loop's own (the default), none, a fixed one the waves never meet, one they always
    wait for, the loop with its look at the other side every four launches -- trajectory, state, episode counters, statistics and
    done list are the same bytes, and equal the oracle's; launches too small to be bound by HBM keep no schedule."""
    meta, _ = G.load_traj('c3_maze32')
    N, T = 65536, 300  # 236 MB of rows per launch: paced
    outs = {}
    for pace in (0, None, 'probing', 20, 400):
        gu_option('rollout_pace', None if pace == 'probing' else pace)
        gu_option('pace_probe_every', 4 if pace == 'probing' else None)
        with Engine(N, spec_of(meta), seed=9) as eng:
            eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T // 3, 'uniform', True, True)       # a shorter launch first (under the 128 MB bar: never paced)
            assert eng.rollout_pacing() is None
            for _ in range(12 if pace == 'probing' else 1):
                eng.rollout(T, 'uniform', True, True, stats=True)
            info = eng.rollout_pacing()
            totals = eng.rollout_pacing_totals()
            assert totals['launches_spent'] == 0 and totals['calibration_ms'] == 0.0
            if pace == 0:
                assert info is None and totals['kinds_paced'] == 0
            else:
                assert info is not None and totals['kinds_paced'] == 1 and info['evaluated'] == (12 if pace == 'probing' else 1)
                if pace in (20, 400):
                    assert info['period'] == pace
                else:  # (the model: 175 ticks; the loop is clamped to 3/4 .. 2 x of it -- its QUALITY is the next test's subject)
                    assert 0.75 * 174 <= info['period'] <= 2 * 176, info
            if pace == 'probing':
                lg = eng.rollout_pace_log()
                assert set(lg['phase'].tolist()) - {0} and (lg['period'] == 0).any(), lg  # it has been running without the limiter in between
                # start over for the comparison: the same launches as the others, on a loop that is in the middle of its cycle
                eng.seed(9)
                eng.reset()
                eng.rollout(T // 3, 'uniform', True, True)
                eng.rollout(T, 'uniform', True, True, stats=True)
            eng.rollout(T, 'uniform', True, True, stats=True)
            tr = eng.read_trajectory(0, T)
            st = eng.get_state()
            outs[pace] = (tr['obs'], tr['reward'], tr['done'], st['pos'], st['done'], st['episode'], st['tcount'], eng.read_stats()[0], eng.done_indices())
    for pace in (None, 'probing', 20, 400):
        assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[pace])), pace
    grid, st = C.Grid.from_lists(**meta), C.State(2048)
    C.reset(grid, 9, st)
    C.rollout(grid, 9, st, T // 3 + T, True, trajectory=False)
    want = C.rollout(grid, 9, st, T, True)
    assert all(np.array_equal(outs[None][i][:, :2048], want[k]) for i, k in enumerate(('obs', 'reward', 'done')))
    gu_option('pace_probe_every', None)
    # a batch whose last workgroup is ragged, with and without a schedule
    N2 = 65536 + 100
    ragged = {}
    for pace in (0, None, 150, 60):
        gu_option('rollout_pace', pace)
        with Engine(N2, spec_of(meta), seed=9, env_id0=1000) as eng:
            eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T, 'uniform', True, True, stats=True)
            eng.rollout(T // 2, 'uniform', True, True, stats=True)
            tr = eng.read_trajectory(0, T // 2)
            st = eng.get_state()
            ragged[pace] = (tr['obs'], tr['reward'], tr['done'], st['pos'], st['episode'], eng.read_stats()[0], eng.read_stats()[1], eng.done_indices())
    for pace in (None, 150, 60):
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
    # a caller-supplied stream and a table policy keep schedules of their own
    with Engine(N, spec_of(meta), seed=9) as eng:
        eng.reset()
        eng.reserve_trajectory(T)
        S = meta['W'] * meta['H']
        eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
        eng.rollout(T, 'sample', False, True)
        assert eng.rollout_pacing('sample', False) is not None and eng.rollout_pacing('uniform', True) is None


def wall_us(eng, n, T, policy, traj):
    eng.sync()
    eng.timer_begin()
    for _ in range(n):
        eng.rollout(T, policy, True, traj)
    return eng.timer_end() / n * 1e3


@pytest.mark.parametrize('kind', ['int32 rows', 'packed rows'])
def test_an_engine_that_is_simply_used_gets_the_paced_rate(kind, gu_option):
    """Rounds 3 and 4 needed gu_rollout_calibrate (a search of a few hundred launches, 40 .. 60 ms) to reach the paced rate; an engine
    left alone ran without a limiter for 1024 launches and then stalled for the search.  Now: no launch of a fresh engine is ever
    spent on anything but the caller's work (launches_spent stays 0, the first launch costs a kernel), no launch takes more than
    twice the median, and launches 200 .. 400 of the fresh engine run within 4 % (int32 rows; measured: -2 .. +3 %, profiles/archive/r05h_matrix.txt,
    r06*_pace_quality.txt; packed rows: 10 %, see below) of the best of (a) no limiter and (b) FIVE FIXED PERIODS around the model (0.92 .. 1.08 x the rows of 16
    steps at 7.2 TB/s), each held on the same engine and buffer -- for the headline launch, where the limiter is worth 10 %, and
    for packed rows at one wave per SIMD, where it is worth nothing and the loop must find that out and switch it off."""
    meta, _ = G.load_traj('c3_maze32')
    N, T = 65536, 1000
    traj = 'packed' if kind == 'packed rows' else True
    packed = traj == 'packed'
    gu_option('rollout_pace', None)
    with Engine(N, spec_of(meta), seed=3) as eng:
        eng.reset()
        eng.reserve_trajectory(T)
        eng.sync()
        t0 = time.perf_counter()
        eng.rollout(T, 'uniform', True, traj)
        eng.sync()
        first_ms = (time.perf_counter() - t0) * 1e3
        assert first_ms < 10.0, first_ms  # (a 0.12 ms kernel + what a first launch of a kernel costs: ~1 ms; the searches of rounds 3 and 4: 40 .. 60)
        # launches 2 .. 59 back to back: start-to-start intervals from the device's own clock (the kind's ring of launch records)
        for _ in range(57):
            eng.rollout(T, 'uniform', True, traj)
        lg = eng.rollout_pace_log('uniform', True, packed=packed)
        iv = lg['interval'][(lg['interval'] > 0) & (lg['seq'] >= 2)] / 100.0  # us (launch 1 was waited for by the host)
        assert len(iv) >= 50 and iv.max() < 2.0 * np.median(iv), (np.median(iv), iv.max())  # (measured: < 1.35 x)
        early_us = float(np.mean(iv[5:32]))
        for _ in range(3):  # launches 59 .. 232
            wall_us(eng, 58, T, 'uniform', traj)
        loop_us = min(wall_us(eng, 56, T, 'uniform', traj) for _ in range(3))  # launches 233 .. 400
        for _ in range(8):  # ... and once its slow adjustment of the aim (blocks of 192 launches) has had 1600 more
            wall_us(eng, 200, T, 'uniform', traj)
        late_us = min(wall_us(eng, 58, T, 'uniform', traj) for _ in range(3))  # launches 2001 .. 2174
        lg = eng.rollout_pace_log('uniform', True, packed=packed)
        assert eng.rollout_pacing_totals()['launches_spent'] == 0
        # (a) no limiter at all, (b) five fixed periods around the model, each run up and held
        eng.set_option('rollout_pace', 0)
        wall_us(eng, 10, T, 'uniform', traj)
        fixed = {0: min(wall_us(eng, 58, T, 'uniform', traj) for _ in range(3))}
        model = 16.0 * N * (4 if packed else 12) / 7200e9 * 1e8  # ticks of 10 ns per 16 steps
        for scale in (0.92, 0.96, 1.0, 1.04, 1.08):
            period = int(round(model * scale))
            eng.set_option('rollout_pace', period)
            wall_us(eng, 20, T, 'uniform', traj)
            fixed[period] = min(wall_us(eng, 58, T, 'uniform', traj) for _ in range(3))
        eng.set_option('rollout_pace', None)
        best_us = min(fixed.values())
        print('store pacing, %s: closed loop %.2f us per launch at launches 233 .. 400, %.2f at launches 2001 .. 2174; fixed periods %s; ratios %.3f, %.3f'
              % (kind, loop_us, late_us, {k: round(v, 2) for k, v in fixed.items()}, loop_us / best_us, late_us / best_us))
        # int32 rows, against the best of the five fixed periods on the same engine: launches 233 .. 400 within 8 % (measured 0.99 .. 1.05
        # on five boxes, profiles/r06*_pace_quality.txt), launches 2001 .. within 5 %.  Packed rows: the loop's probe finds the limiter
        # not worth having and runs without it, 1.5 % behind 'no limiter' -- while the SHORTEST of the five fixed periods (0.92 x the
        # model) is 5 .. 7 % quicker than either.  The rule is frozen (round-5 review, item 6); the gap is stated, not tuned away.
        assert loop_us <= (1.10 if packed else 1.08) * best_us, dict(loop=loop_us, fixed=fixed, log_period=lg['period'][-8:], phase=lg['phase'][-8:])
        assert late_us <= (1.10 if packed else 1.05) * best_us, dict(late=late_us, fixed=fixed, log_period=lg['period'][-8:], phase=lg['phase'][-8:])
        assert early_us <= 1.15 * best_us, dict(early=early_us, best=best_us)  # launches 6 .. 32: on the way down from the model
