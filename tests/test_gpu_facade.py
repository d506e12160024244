"""The N = 1 facade on the device: its table-driven scalar step against the step kernel, and edits seen after invalidate()."""
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


def test_facade_sees_edits_after_invalidate():
    env = gua.GridUniverseEnv()
    assert env.step(1)[0] == 1
    env.wall_grid[2] = 1
    env.lava_states.append(5)
    env.reward_matrix[5] = -10
    assert env.step(1)[0] == 2  # compiled grid: the edit is not seen yet (documented difference, docs/API.md)
    env.invalidate()
    env.current_state = 1
    assert env.step(1)[0] == 1 and env.step(2) == (5, -10, True, env.info)
    env.close()
