"""Pins the scalar Python oracle (oracle/ref_env.py, oracle/maze.py) to golden
vectors captured from the real reference, incl. the reference's own ten KATs."""
import random

import numpy as np
import pytest

from oracle import gu_rng
from oracle.ref_env import OracleGridUniverseEnv, UnsupportedMode
from tests import _golden as G


def _triple(step):
    o, r, d, _ = step
    return [int(o), int(r), bool(d)]


def _env_from_spec(meta):
    """Instance with exactly the reference instance's grid (incl. quirky reward matrices)."""
    env = OracleGridUniverseEnv(grid_shape=(meta['W'], meta['H']), initial_state=list(meta['starts']),
                                goal_states=list(meta['goals']), lava_states=list(meta['lava']),
                                walls=list(meta['walls']))
    assert [int(r) for r in env.reward_matrix] == meta['reward']
    return env


@pytest.mark.parametrize('kat', G.load_json('kat.json'), ids=lambda k: k['name'])
def test_reference_kats(kat):
    for pick, run in enumerate(kat['runs']):
        kw = dict(kat['kwargs'])
        if 'grid_shape' in kw:
            kw['grid_shape'] = tuple(kw['grid_shape'])
        if kat['level']:
            kw['custom_world_fp'] = G.level_path(kat['level'])
        env = OracleGridUniverseEnv(**kw)
        if kat['level']:
            env.current_state = env.starting_states[pick]
        assert int(env.current_state) == run['first_state']
        got = [_triple(env.step(a)) for a in kat['actions']]
        assert got == run['steps']


def test_kat_assertions_of_the_reference_tests():
    """The literal assertions of tests/test_griduniverse.py:49-176."""
    env = OracleGridUniverseEnv(walls=[1])
    assert env.step(1)[0] == 0
    env = OracleGridUniverseEnv()
    dones = [env.step(a)[2] for a in [1, 1, 1, 2, 2, 2]]
    assert dones == [False] * 5 + [True]
    env = OracleGridUniverseEnv(grid_shape=(25, 30))
    dones = [env.step(a)[2] for a in [1] * 24 + [2] * 29]
    assert dones.index(True) == 52
    env = OracleGridUniverseEnv(lava_states=[1])
    _, r, d, _ = env.step(env.action_descriptor_to_int['RIGHT'])
    assert r == -10 and d
    env = OracleGridUniverseEnv()
    prev, same = env.reset(), []
    for i, a in enumerate([3, 0, 1, 1, 1, 1, 2, 2, 3, 2, 2, 3, 3, 3]):
        o = env.step(a)[0]
        if o == prev:
            same.append(i)
        prev = o
    assert same[:5] == [0, 1, 5, 10, 13]


@pytest.mark.parametrize('case', [c for c in G.load_json('errors.json') if 'kwargs' in c],
                         ids=lambda c: str(c['kwargs']))
def test_ctor_errors(case):
    kw = dict(case['kwargs'])
    if kw.get('grid_shape') == 'set':
        kw['grid_shape'] = set([2, 3])
    if case['error'] is None:
        OracleGridUniverseEnv(**kw)
        return
    with pytest.raises(Exception) as ei:
        OracleGridUniverseEnv(**kw)
    assert type(ei.value).__name__ == case['error']
    assert str(ei.value) == case['message']


@pytest.mark.parametrize('case', [c for c in G.load_json('errors.json') if 'lines' in c], ids=lambda c: c['name'])
def test_loader_errors(case):
    env = OracleGridUniverseEnv()
    with pytest.raises(Exception) as ei:
        env._load_lines(case['lines'])
    assert type(ei.value).__name__ == case['error'] and str(ei.value) == case['message']


def test_render_and_quirks():
    g = G.load_json('render_quirks.json')
    env = OracleGridUniverseEnv(walls=[1], lava_states=[2])
    assert env.render(mode='ansi').getvalue() == g['render_walls1_lava2']
    env = OracleGridUniverseEnv()
    frames = [env.render(mode='ansi').getvalue()]
    for a in [1, 2, 2, 1, 1, 2]:
        env.step(a)
        frames.append(env.render(mode='ansi').getvalue())
    assert frames == g['render_default_walk']
    env = OracleGridUniverseEnv(custom_world_fp=G.level_path('test_env.txt'))
    env.current_state = env.starting_states[0]
    assert env.render(mode='ansi').getvalue() == g['render_test_env']
    env = OracleGridUniverseEnv(grid_shape=(5, 3), goal_states=[14, 7], lava_states=[7, 3], walls=[6, 14])
    assert env.render(mode='ansi').getvalue() == g['render_5x3_overlaps']
    with pytest.raises(UnsupportedMode):
        env.render(mode='rgb_array')

    q = g['quirks']
    env = OracleGridUniverseEnv()
    env.current_state = 11
    assert [_triple(env.step(a)) for a in [2, 0, 3, 1]] == q['absorbing']
    env = OracleGridUniverseEnv(walls=[0])
    assert [_triple(env.step(a)) for a in [1, 3, 2, 0]] == q['start_on_wall']
    env = OracleGridUniverseEnv(goal_states=[5], walls=[5])
    seq = [_triple(env.step(a)) for a in [1, 2, 2, 0]]
    env.current_state = 5
    seq.append(_triple(env.step(1)))
    assert seq == q['goal_is_wall']
    env = OracleGridUniverseEnv(goal_states=[1, 15], lava_states=[1])
    assert [_triple(env.step(a)) for a in [1, 1]] == q['goal_and_lava']
    env = OracleGridUniverseEnv(goal_states=[-1])
    env.current_state = 14
    assert [int(r) for r in env.reward_matrix] == q['negative_goal']['reward']
    assert [_triple(env.step(a)) for a in [1, 1, 3]] == q['negative_goal']['steps']
    env = OracleGridUniverseEnv(custom_world_fp=G.level_path('test_env.txt'))
    assert dict(n=env.observation_space.n, world_size=env.world.size) == q['stale_observation_space']
    env = OracleGridUniverseEnv()
    assert [type(x).__name__ for x in env.step(1)] == q['types']
    env = OracleGridUniverseEnv(lava_states=[1])
    got = [[int(x) if k < 2 else bool(x) for k, x in enumerate(env.look_step_ahead(s, a, c))]
           for (s, a, c) in [(1, 1, True), (1, 1, False), (15, 3, True), (15, 3, False), (1, 2, False), (0, 1, False)]]
    assert got == q['care_about_terminal_false']
    t = q['lsa_table_6x5']
    env = _env_from_spec(t['spec'])
    for care in (True, False):
        got = [[[int(x) if k < 2 else bool(x) for k, x in enumerate(env.look_step_ahead(s, a, care))]
                for a in range(4)] for s in range(30)]
        assert got == t['table'][str(care)]


def test_step_action_domain():
    for c in [c for c in G.load_json('errors.json') if 'step_action' in c]:
        env = OracleGridUniverseEnv()
        env.current_state = c['from_state']
        if c['error']:
            with pytest.raises(Exception) as ei:
                env.step(c['step_action'])
            assert type(ei.value).__name__ == c['error']
        else:
            assert _triple(env.step(c['step_action'])) == c['result']


@pytest.mark.parametrize('key', sorted(G.load_json('mazes.json')))
def test_seeded_maze_generation(key):
    m = G.load_json('mazes.json')[key]
    random.seed(m['seed'])
    np.random.seed(m['seed'])
    env = OracleGridUniverseEnv(grid_shape=(m['W'], m['H']), random_maze=True)
    tail = [random.random(), float(np.random.random())]
    rows = []
    for y in range(m['H']):
        rows.append(''.join('#' if (y * m['W'] + x) in set(env.wall_indices) else
                            'x' if (y * m['W'] + x) in env.starting_states else
                            'G' if (y * m['W'] + x) in env.goal_states else 'o' for x in range(m['W'])))
    assert rows == m['rows']
    assert env.starting_states == m['start'] and env.goal_states == m['goal']
    assert len(env.wall_indices) == m['n_walls'] and env.initial_state == m['initial_state']
    assert tail == m['rng_tail'], 'RNG consumption order differs from the reference'


@pytest.mark.parametrize('fn', sorted(G.load_json('levels.json')))
def test_level_loader(fn):
    want = G.load_json('levels.json')[fn]
    random.seed(7)
    env = OracleGridUniverseEnv(custom_world_fp=G.level_path(fn))
    assert (env.x_max, env.y_max) == (want['W'], want['H'])
    assert env.starting_states == want['starts'] and env.goal_states == want['goals']
    assert env.lava_states == want['lava'] and env.wall_indices == want['walls']
    assert env.initial_state == want['initial_state_seed7']
    assert env.observation_space.n == want['observation_space_n']


def test_loader_strips_blanks_and_empty_lines(tmp_path):
    # env:248-249: every blank is removed, empty lines dropped (maze_101x101.txt is blank-separated)
    p = tmp_path / 'lvl.txt'
    p.write_text('x o  #\n\n o\tL G \n')
    env = OracleGridUniverseEnv(custom_world_fp=str(p))
    assert (env.x_max, env.y_max) == (3, 2)
    assert (env.starting_states, env.wall_indices, env.lava_states, env.goal_states) == ([0], [2], [4], [5])


SMALL = [n for n in G.traj_names() if n not in ('maze101',)]


@pytest.mark.parametrize('name', SMALL)
def test_trajectories_scalar_oracle(name):
    """Per-instance Python oracle driven exactly like the reference was (tests/golden/make_golden.py: rollout)."""
    meta, z = G.load_traj(name)
    env = _env_from_spec(meta)
    T, N = z['actions'].shape
    n_cols = min(N, 16)  # the C oracle covers every column; keep the Python loop short
    for j in range(n_cols):
        gid, episode = meta['env_id0'] + j, 0

        def do_reset():
            nonlocal episode
            env.reset()
            s = meta['starts'][gu_rng.start_index(meta['seed'], gid, episode, len(meta['starts']))]
            env.current_state = env.previous_state = env.initial_state = s
            episode += 1
            return s
        assert do_reset() == z['first_state'][j]
        done = False
        for t in range(T):
            if meta['auto_reset'] and done:
                do_reset()
            o, r, done, _ = env.step(int(z['actions'][t, j]))
            assert (o, r, done) == (z['obs'][t, j], z['reward'][t, j], bool(z['done'][t, j])), (name, j, t)
