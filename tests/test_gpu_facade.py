"""The N = 1 facade on the device: its table-driven scalar step against the step kernel, and in-place edits of the grid seen at the next step."""
import numpy as np
import pytest

from tests import _golden as G
import griduniverse_amd as gua

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('kat', G.load_json('kat.json'), ids=lambda k: k['name'])
def test_facade_table_step_equals_the_step_kernel(kat):
    """GridUniverseEnv.step() indexes the transition table the look-ahead kernel produced; step_on_device() launches the
    step kernel.  Both must give the reference's stream for its own ten KATs."""
    for pick, run in enumerate(kat['runs']):
        kw = dict(kat['kwargs'])
        if 'grid_shape' in kw:
            kw['grid_shape'] = tuple(kw['grid_shape'])
        if kat['level']:
            kw['custom_world_fp'] = G.level_path(kat['level'])
        for method in ('step', 'step_on_device'):
            env = gua.GridUniverseEnv(**kw)
            if kat['level']:
                env.current_state = env.starting_states[pick]
            got = []
            for a in kat['actions']:
                o, r, d, info = getattr(env, method)(a)
                assert type(o) is int and type(r) is np.int64 and type(d) is bool and info is env.info
                assert env.current_state == o and env.done == d
                got.append([o, int(r), d])
            assert got == run['steps'], method
            env.close()


def test_facade_sees_in_place_edits_at_the_next_step():
    """The reference consults goal_states / lava_states / wall_grid / reward_matrix on every step (env:157-174), so an edit
    takes effect at once.  The facade compiles the grid into the engine -- and drops the compiled form whenever one of those
    attributes is edited in place or replaced (round 5 needed env.invalidate() for it)."""
    from oracle.ref_env import OracleGridUniverseEnv
    rs = np.random.RandomState(4)
    for method in ('step', 'step_on_device'):
        env, ref = gua.GridUniverseEnv(grid_shape=(5, 4)), OracleGridUniverseEnv(grid_shape=(5, 4))
        edits = [lambda e: e.wall_grid.__setitem__(2, 1), lambda e: e.lava_states.append(7), lambda e: e.reward_matrix.__setitem__(7, -10),
                 lambda e: e.goal_states.remove(19), lambda e: setattr(e, 'goal_states', [18, 3]), lambda e: e.reward_matrix.__setitem__(3, 10),
                 lambda e: e.wall_grid.__setitem__(slice(12, 14), 1), lambda e: e.lava_states.clear(), lambda e: e.starting_states.extend([6, 11])]
        for k in range(len(edits) + 1):
            for _ in range(25):
                a = int(rs.randint(4))
                got, want = getattr(env, method)(a), ref.step(a)
                assert got[:3] == want[:3], (method, k, a, got, want)
                if got[2]:
                    env.current_state = ref.current_state = int(rs.choice(env.starting_states))
                    env.done = ref.done = False
            if k < len(edits):
                edits[k](env)
                edits[k](ref)
        assert list(env.goal_states) == [18, 3] and env.wall_grid[12] == 1 and env.starting_states == [0, 6, 11]
        env.close()
    # a write through a VIEW of an array is the one edit the env cannot see: invalidate() remains for it
    env = gua.GridUniverseEnv()
    assert env.step(1)[0] == 1
    view = env.wall_grid[2:]
    view[0] = 1
    assert env.step(1)[0] == 2
    env.invalidate()
    env.current_state = 1
    assert env.step(1)[0] == 1
    env.close()
