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
