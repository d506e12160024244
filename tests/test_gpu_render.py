"""Headless RGB frames rendered on the device.  Which texture a cell gets and the geometry of the policy arrows are the
reference viewer's and are checked against tests/golden/arrows.json (captured from its own method bodies); the flat colours,
the grid line and the agent square are build-defined and restated here."""
import numpy as np
import pytest

import griduniverse_amd as gua
from griduniverse_amd.engine import Engine
from griduniverse_amd.grid import GridSpec

pytestmark = pytest.mark.gpu

WALL, LAVA, GOAL, FLOOR, AGENT = (64, 64, 64), (220, 60, 30), (40, 180, 60), (220, 220, 220), (40, 90, 220)


def expected_frame(spec, pos, px):
    """Tiles by the viewer's texture rule (goal, else lava, else wall, else ground: oracle/render.py, pinned to arrows.json) in the
    build's palette, grid line, agent square."""
    from oracle import render as R
    S = spec.W * spec.H
    kinds = R.tile_kinds(S, np.flatnonzero(spec.goal).tolist(), np.flatnonzero(spec.lava).tolist(), np.flatnonzero(spec.wall).tolist())
    return R.tile_frame(spec.W, spec.H, kinds, px, agent=pos)


@pytest.mark.parametrize('px', [1, 4, 9, 16])
def test_frames_match_the_colour_rules(px):
    rs = np.random.RandomState(px)
    specs = [GridSpec(7, 5, [int(rs.randint(35))], [int(x) for x in rs.choice(35, 2, replace=False)],
                      [int(x) for x in rs.choice(35, 2, replace=False)], [int(x) for x in rs.choice(35, 8, replace=False)])
             for _ in range(3)]
    with Engine(12, specs[0], seed=1) as eng:
        eng.set_grids(specs)
        eng.reset()
        eng.rollout(5, 'uniform', True, trajectory=False)
        pos = eng.get_state()['pos']
        frames = eng.render_rgb(0, 12, px)
        assert frames.shape == (12, 5 * px, 7 * px, 3) and frames.dtype == np.uint8
        for e in range(12):
            assert np.array_equal(frames[e], expected_frame(specs[e // 4], pos[e], px)), e
        assert np.array_equal(eng.render_rgb(5, 2, px), frames[5:7])
        with pytest.raises(gua.GuError):
            eng.render_rgb(11, 2, px)


def test_facade_rgb_array_mode():
    env = gua.GridUniverseEnv(walls=[1], lava_states=[2])
    frame = env.render(mode='rgb_array')
    assert frame.shape == (64, 64, 3)
    assert tuple(frame[8, 8]) == AGENT and tuple(frame[8, 24]) == WALL and tuple(frame[8, 40]) == LAVA and tuple(frame[56, 56]) == GOAL
    env.step(2)
    frame = env.render(mode='rgb_array')
    assert tuple(frame[8, 8]) == FLOOR and tuple(frame[24, 8]) == AGENT


@pytest.mark.parametrize('px', [52, 16, 26, 3])
def test_policy_arrow_frames_equal_the_rasterised_reference_geometry(px):
    """tests/golden/arrows.json holds what the reference's own Viewer.render_policy_arrows and tile loop produced (head triangles,
    shafts, texture class per cell; make_golden.py lifts the method bodies and runs them without a window).  Those primitives,
    rasterised by the rule include/gu.h states (oracle/render.py), must equal gu_render_policy_rgb pixel for pixel: at the
    reference's own 52-pixel tile and scaled (16, 26 -- where shaft ends fall on pixel centres -- and 3)."""
    from oracle import render as R
    from tests import _golden as G
    for case in G.load_json('arrows.json')['cases']:
        W, H = case['W'], case['H']
        spec = GridSpec(W, H, case['starts'], case['goals'], case['lava'], case['walls'], case['reward'])
        pi = np.array(case['policy'], dtype=np.float64)
        with Engine(2, spec) as eng:
            eng.vi_set(np.zeros(W * H), pi)
            frame = eng.render_policy_rgb(px)
        want = R.policy_frame(W, H, case['tiles'], case['geoms'], px)
        assert frame.shape == want.shape == (H * px, W * px, 3)
        assert np.array_equal(frame, want), (case['name'], px, np.argwhere((frame != want).any(axis=2))[:5])
        if px == 52:  # arrows are really there
            assert (frame == 20).all(axis=2).sum() > 30 * len(case['geoms'])


def test_tiles_follow_the_viewers_texture_rule_where_cell_kinds_overlap():
    """goal+lava and goal+wall cells show the goal texture, lava+wall the lava one (rendering.py:119-133; arrows.json 'tiles'),
    in frames of the engine and of the facade; the agent square sits on its cell."""
    from oracle import render as R
    from tests import _golden as G
    case = next(c for c in G.load_json('arrows.json')['cases'] if c['name'] == 'quirk6x4')
    W, H = case['W'], case['H']
    spec = GridSpec(W, H, case['starts'], case['goals'], case['lava'], case['walls'], case['reward'])
    with Engine(3, spec, seed=4) as eng:
        eng.reset()
        eng.rollout(3, 'uniform', True, trajectory=False)
        pos = eng.get_state()['pos']
        frames = eng.render_rgb(0, 3, 8)
        for e in range(3):
            assert np.array_equal(frames[e], R.tile_frame(W, H, case['tiles'], 8, agent=int(pos[e]))), e
    env = gua.GridUniverseEnv(grid_shape=(W, H), goal_states=case['goals'], lava_states=case['lava'], walls=case['walls'])
    assert np.array_equal(env.render(mode='rgb_array'), R.tile_frame(W, H, case['tiles'], 16, agent=0))
    pi = np.array(case['policy'])
    assert np.array_equal(env.render_policy_arrows(pi, mode='rgb_array', cell_px=52), R.policy_frame(W, H, case['tiles'], case['geoms'], 52))
    env.close()


def test_policy_rows_outside_the_reference_domain():
    """NaN draws nothing (the reference raises on it), probabilities above 1 stay inside their tile (the reference's arrow would
    cross into the neighbour): documented divergences outside [0, 1]; the frame stays well-formed."""
    W, H = 3, 1
    spec = GridSpec(W, H, [0], [2], [], [])
    pi = np.array([[np.nan, 0.0, 0.0, 0.0], [0.0, 2.0, 0.0, -1.0], [0.25, 0.25, 0.25, 0.25]])
    with Engine(1, spec) as eng:
        eng.vi_set(np.zeros(3), pi)
        frame = eng.render_policy_rgb(52)
    dark = (frame == 20).all(axis=2)
    assert dark[:, :52].sum() == 0 and dark[:, 52:104].sum() > 0 and dark[:, 104:].sum() == 0  # (state 2 is terminal: no arrows)


@pytest.mark.parametrize('px', [52, 16])
def test_agent_trail_equals_the_reference_quads_and_the_stated_blend(px):
    """tests/golden/trail.json: the quads the reference's own trail loop (core/envs/rendering.py:287-311, lifted with ast) emitted
    after the reference env's own step() / reset() calls -- which tile, which alpha, in which order.  The engine with
    gu_trail_enable(500), driven by the same actions through gu_step / gu_reset, must keep the same last_n_states, and its frame
    must be those quads blended over the tiles by the integer rule include/gu.h states (oracle/render.blend_trail).  The same
    walk as ONE rollout on an uploaded action stream (int32 and packed rows), and an auto-reset walk step by step against one
    rollout, must leave the same rings."""
    import json
    import os

    from oracle import render as R
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'trail.json')))
    for case in gold['cases']:
        spec = GridSpec(case['W'], case['H'], case['starts'], case['goals'], case['lava'], case['walls'])
        S, N = case['W'] * case['H'], 3
        kinds = R.tile_kinds(S, case['goals'], case['lava'], case['walls'])
        choice = np.full(N, case['starts'].index(case['start_state']), np.int32)
        with Engine(N, spec, seed=1) as eng:
            with pytest.raises(gua.GuError):
                eng.trail_read(0, 1)  # off by default
            eng.trail_enable(500)
            eng.reset(start_choice=choice)
            for a in case['actions']:
                if a < 0:
                    eng.reset()
                else:
                    eng.step(np.full(N, a, np.int32))
            pos = eng.get_state()['pos']
            assert pos.tolist() == [case['current_state']] * N
            assert eng.trail_read(0, N) == [case['last_n_states']] * N, case['name']
            want = R.blend_trail(R.tile_frame(case['W'], case['H'], kinds, px, agent=case['current_state']), case['W'], px,
                                 [(s, a) for s, a in case['quads']])
            frames = eng.render_rgb(0, N, px)
            for e in range(N):
                assert np.array_equal(frames[e], want), (case['name'], e)
            # a rollout that keeps no rows cannot feed the trail; the fused sweep + step launches neither
            with pytest.raises(gua.GuError):
                eng.rollout(4, 'uniform', True, trajectory=False, stats=True)
            eng.trail_enable(0)
            assert np.array_equal(eng.render_rgb(0, 1, px)[0], R.tile_frame(case['W'], case['H'], kinds, px, agent=case['current_state']))
        if -1 in case['actions']:
            continue
        # the same walk as one rollout on the uploaded stream: int32 rows, then packed rows
        acts = np.repeat(np.asarray(case['actions'], np.int32)[:, None], N, axis=1)
        for traj in (True, 'packed'):
            with Engine(N, spec, seed=1) as eng:
                eng.trail_enable(500)
                eng.reset(start_choice=choice)
                eng.upload_actions(acts)
                eng.reserve_trajectory(len(acts))
                eng.rollout(len(acts), 'stream', False, trajectory=traj)
                assert eng.trail_read(0, N) == [case['last_n_states']] * N, (case['name'], traj)
    # auto-reset: the lazy reset of the step after a done empties the ring -- step by step against one rollout of the same launch
    spec = GridSpec(4, 4, [0], [15], [5], [])
    T, N = 300, 130
    rings = {}
    for how in ('steps', 'rollout'):
        with Engine(N, spec, seed=9) as eng:
            eng.trail_enable(40)
            eng.reset()
            eng.reserve_trajectory(T)
            if how == 'rollout':
                eng.rollout(T, 'uniform', True, trajectory=True)
            else:
                for _ in range(T):
                    eng.rollout(1, 'uniform', True, trajectory=True)
            rings[how] = eng.trail_read(0, N)
            assert all(len(r) <= 40 for r in rings[how]) and any(len(r) < 40 for r in rings[how]) and any(len(r) == 40 for r in rings[how])
    assert rings['steps'] == rings['rollout']


def test_the_trail_follows_set_state():
    """gu_set_state moves envs by hand and installs done flags: an env that was moved has no trail to continue (its ring is
    emptied), and the NEXT lazy reset is decided by the installed flag, not by the one behind the trail's last append (round 4
    kept a private flag that set_state never touched)."""
    spec = GridSpec(4, 4, [0], [15], [], [])
    N = 4
    with Engine(N, spec, seed=1) as eng:
        eng.trail_enable(500)
        eng.reset()
        for a in (1, 1, 2):  # right, right, down: 1, 2, 6
            eng.step(np.full(N, a, np.int32))
        assert eng.trail_read(0, N) == [[1, 2, 6]] * N
        # done flags installed by hand, positions untouched: the trail stays, the next auto-reset step empties it first
        eng.set_state(done=np.array([1, 0, 1, 0], np.int32))
        assert eng.trail_read(0, N) == [[1, 2, 6]] * N
        eng.step(np.full(N, 1, np.int32), auto_reset=True)  # done envs: back to the start cell 0, then right -> 1
        assert eng.trail_read(0, N) == [[1], [1, 2, 6, 7], [1], [1, 2, 6, 7]]
        # envs moved by hand: every ring starts over from there
        eng.set_state(pos=np.array([5, 5, 9, 9], np.int32), done=np.zeros(N, np.int32))
        assert eng.trail_read(0, N) == [[]] * N
        eng.step(np.full(N, 2, np.int32), auto_reset=True)  # down
        assert eng.trail_read(0, N) == [[9], [9], [13], [13]]
        # a done flag cleared by hand keeps the env where it is: no reset, the trail goes on
        eng.set_state(pos=np.array([14, 14, 14, 14], np.int32))
        eng.step(np.full(N, 1, np.int32), auto_reset=True)  # right -> 15, the goal: done
        eng.set_state(done=np.zeros(N, np.int32))
        eng.step(np.full(N, 3, np.int32), auto_reset=True)  # (the goal is absorbing for a step that is not a reset: stays on 15)
        assert eng.trail_read(0, N) == [[15, 15]] * N
