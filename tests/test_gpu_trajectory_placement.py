"""gu_reserve_trajectory's candidate search: what it keeps, what it reports, what later engines of the process may hold."""
import numpy as np
import pytest

from griduniverse_amd import Engine, GridSpec, _lib
from griduniverse_amd import _lib
from griduniverse_amd.engine import Engine
from griduniverse_amd.grid import GridSpec
from oracle import c_oracle as C
from tests import _golden as G

pytestmark = pytest.mark.gpu

def spec_of(meta):
    return GridSpec(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'])


def test_trajectory_buffer_is_chosen_among_candidates_and_kept_when_large_enough(gu_option):
    """gu_reserve_trajectory probes candidate allocations for buffers of 64 MB and more and keeps the one HBM writes fastest;
    a buffer that is already large enough is kept.  Results never depend on which allocation was taken."""
    meta, _ = G.load_traj('c3_maze32')
    N, T = 32768, 256  # 3 x 32 MB planes
    grid = C.Grid.from_lists(**meta)
    outs = []
    for cand in ('1', '5'):
        gu_option('traj_candidates', int(cand))
        st = C.State(N)
        with Engine(N, spec_of(meta), seed=2) as eng:
            assert np.array_equal(eng.reset(), C.reset(grid, 2, st))
            eng.reserve_trajectory(T)
            n, best, worst = eng.trajectory_placement()
            assert (n == 1 and best == 0.0) if cand == '1' else (1 <= n <= 5 and 0.0 < best <= worst)  # (a candidate that is fast in absolute terms ends the search)
            eng.reserve_trajectory(T // 2)  # large enough already: same buffer, same placement record
            assert eng.trajectory_placement() == (n, best, worst)
            eng.rollout(T // 2, 'uniform', True, True)
            got = eng.read_trajectory(0, T // 2)
            want = C.rollout(grid, 2, st, T // 2, True)
            assert all(np.array_equal(got[k], want[k]) for k in got)
            eng.reserve_trajectory(2 * T)   # grows: chosen again
            eng.rollout(2 * T, 'uniform', True, True)
            want = C.rollout(grid, 2, st, 2 * T, True)
            got = eng.read_trajectory(0, 2 * T)
            assert all(np.array_equal(got[k], want[k]) for k in got)
            outs.append(got['obs'][-1].copy())
    assert np.array_equal(outs[0], outs[1])
    with Engine(64, spec_of(meta)) as eng:  # small buffers are simply allocated
        eng.reserve_trajectory(16)
        assert eng.trajectory_placement()[0] == 1
    # buffers of 256 MiB and more: when the back-to-back candidates all look alike the search continues behind spacers
    gu_option('traj_candidates', 2)
    gu_option('traj_far_candidates', 3)
    gu_option('traj_stride_mib', 1536)
    gu_option('traj_probe_all', 1)  # (without it the search ends where two back-to-back candidates are alike or one is fast)
    N, T = 65536, 400
    with Engine(N, spec_of(meta), seed=2) as eng:
        eng.reset()
        eng.reserve_trajectory(T)
        n, best, worst = eng.trajectory_placement()
        assert 1 <= n <= 5 and 0.0 < best <= worst
        eng.rollout(T, 'uniform', True, True)
        got = eng.read_trajectory(T - 1, 1)
        st = C.State(2048)
        C.reset(grid, 2, st)
        want = C.rollout(grid, 2, st, T, True)
        assert all(np.array_equal(got[k][0, :2048], want[k][T - 1]) for k in got)


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
