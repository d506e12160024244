"""Headless RGB frames rendered on the device (gu_render_rgb): the colour rules of csrc/gu_render.hip restated in
numpy and compared pixel for pixel, for a single grid, a multi-grid engine and the facade's render('rgb_array')."""
import numpy as np
import pytest

import griduniverse_amd as gua
from griduniverse_amd.engine import Engine
from griduniverse_amd.grid import GridSpec

pytestmark = pytest.mark.gpu

WALL, LAVA, GOAL, FLOOR, AGENT = (64, 64, 64), (220, 60, 30), (40, 180, 60), (220, 220, 220), (40, 90, 220)


def expected_frame(spec, pos, px):
    W, H = spec.W, spec.H
    img = np.zeros((H * px, W * px, 3), np.uint8)
    for s in range(W * H):
        colour = WALL if spec.wall[s] else LAVA if spec.reward[s] == -10 else GOAL if (spec.reward[s] == 10 or spec.goal[s] or spec.lava[s]) else FLOOR
        y, x = divmod(s, W)
        tile = np.empty((px, px, 3), np.uint8)
        tile[:] = colour
        if px >= 4:
            tile[0, :] = np.array(colour) * 3 // 4
            tile[:, 0] = np.array(colour) * 3 // 4
        if s == pos:
            lo, hi = px // 4, px - px // 4
            tile[lo:hi, lo:hi] = AGENT
        img[y * px:(y + 1) * px, x * px:(x + 1) * px] = tile
    return img


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


def expected_policy_frame(spec, pi, px):
    """csrc/gu_render.hip: gu_render_policy_kernel restated (integer arithmetic on doubled, tile-centred coordinates)."""
    W, H = spec.W, spec.H
    img = expected_frame(spec, -1, px)
    terminal = np.asarray(spec.goal, bool) | np.asarray(spec.lava, bool)
    shaft = max(1, px // 26)
    for s in range(W * H):
        if terminal[s] or spec.wall[s]:
            continue
        cy, cx = divmod(s, W)
        for iy in range(px):
            for ix in range(px):
                X, Y = 2 * ix + 1 - px, px - (2 * iy + 1)
                on = False
                for act in range(4):
                    p = pi[s, act]
                    if not p >= 0.1:
                        continue
                    L = int(min(np.rint(p * 20.0), 1000.0))
                    t = (Y, X, -Y, -X)[act]
                    u = abs(Y if act & 1 else X)
                    base, tip = 2 * L * px, 2 * (L + 5) * px
                    on |= t >= 0 and 52 * t <= base and u <= shaft
                    on |= base < 52 * t <= tip and 52 * u <= tip - 52 * t
                if on:
                    img[cy * px + iy, cx * px + ix] = (20, 20, 20)
    return img


@pytest.mark.parametrize('px', [3, 16, 52])
def test_policy_arrow_frames(px):
    rs = np.random.RandomState(px)
    W, H = 6, 4
    S = W * H
    spec = GridSpec(W, H, [0], [S - 1, 7], [9], [2, 3, 14])
    pi = rs.dirichlet(np.ones(4) * 0.6, S)
    pi[1] = [1, 0, 0, 0]
    pi[4] = [0, 0, 0, 1]
    pi[5] = [0.25, 0.25, 0.25, 0.25]
    pi[6] = [0.5, 0.5, 0, 0]
    pi[8] = [0.099, 0.101, 0.4, 0.4]
    pi[10] = [np.nan, 2.0, 0, -1]
    pi[11] = [0.125, 0.375, 0.3, 0.2]  # round-half-even: 2.5 -> 2, 7.5 -> 8
    with Engine(2, spec) as eng:
        eng.vi_set(np.zeros(S), pi)
        frame = eng.render_policy_rgb(px)
        assert frame.shape == (H * px, W * px, 3)
        assert np.array_equal(frame, expected_policy_frame(spec, pi, px))
        if px >= 16:  # an arrow is really there: the one-hot UP policy of state 1 paints above the centre, not below
            tile = frame[0:px, px:2 * px]
            dark = (tile == 20).all(axis=2)
            assert dark[:px // 2].sum() > 3 and dark[px // 2 + 1:].sum() == 0
    env = gua.GridUniverseEnv(grid_shape=(W, H), goal_states=[S - 1, 7], lava_states=[9], walls=[2, 3, 14])
    assert np.array_equal(env.render_policy_arrows(pi, mode='rgb_array', cell_px=px), expected_policy_frame(spec, pi, px))
    env.close()
