"""Pins oracle/mc.py (and the stream-2 action sampling of the C oracle) to the goldens captured from the
reference's own monte_carlo_evaluation (run_episode patched to replay the fixture's episodes)."""
import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import gu_rng
from oracle import mc as omc
from tests import _golden as G


@pytest.mark.parametrize('name', G.mc_names())
def test_mc_restatement_bit_exact(name):
    meta, z = G.load_mc(name)
    S = meta['W'] * meta['H']
    for run in meta['runs'][::2] if name == 'maze8_uniform' else meta['runs']:
        eps = omc.episodes_from_trajectory(z['first_state'], z['obs'], z['reward'], z['done'])
        v, visits = omc.monte_carlo_evaluation(S, eps, every_visit=run['every_visit'], incremental_mean=run['incremental_mean'],
                                               stationary_env=run['stationary_env'], discount_factor=run['discount_factor'],
                                               threshold=run['threshold'], alpha=run['alpha'])
        assert v.tobytes() == z[run['key']].tobytes(), (name, run)


@pytest.mark.parametrize('name', G.mc_names())
def test_sampled_episodes_reproduce(name):
    """The fixture's episodes come back from the C oracle (policy sampling on RNG stream 2), and the scalar
    Python RNG restatement agrees with it action by action."""
    meta, z = G.load_mc(name)
    grid = C.Grid.from_lists(**meta)
    st = C.State(meta['N'])
    assert np.array_equal(C.reset(grid, meta['seed'], st), z['first_state'])
    traj = C.rollout(grid, meta['seed'], st, meta['T'], auto_reset=False, pi=z['policy'])
    for k in ('obs', 'reward', 'done'):
        assert np.array_equal(traj[k], z[k])
    for e in (0, 7, meta['N'] - 1):
        s = int(z['first_state'][e])
        for t in range(min(meta['T'], 40)):
            a = gu_rng.sampled_action(meta['seed'], e, t, z['policy'][s])
            nxt = C.look_step_ahead(grid, [s], [a])[0][0]
            assert nxt == z['obs'][t, e]
            s = int(nxt)
