"""oracle/render.py (the viewer's tile rule and policy-arrow geometry, restated) against tests/golden/arrows.json -- what the
reference's own method bodies produced (make_golden.py: capture_arrows)."""
import numpy as np

from oracle import render as R
from tests import _golden as G


def test_tile_kinds_and_arrow_geometry_equal_the_reference():
    data = G.load_json('arrows.json')
    assert data['tile_dim'] == R.TILE_DIM and data['arrow_base_length_full_prob'] == R.FULL_LENGTH
    assert (data['arrow_width'], data['arrow_height']) == (R.ARROW_WIDTH, R.ARROW_HEIGHT)
    assert len(data['cases']) >= 4
    n_arrows = 0
    for case in data['cases']:
        S = case['W'] * case['H']
        assert R.tile_kinds(S, case['goals'], case['lava'], case['walls']) == case['tiles'], case['name']
        assert R.arrow_geoms(case['policy'], S, case['goals'], case['lava'], case['walls']) == case['geoms'], case['name']
        n_arrows += len(case['geoms'])
    assert n_arrows > 200
    quirk = next(c for c in data['cases'] if c['name'] == 'quirk6x4')
    assert [quirk['tiles'][s] for s in (7, 13, 9)] == ['goal', 'goal', 'lava']  # goal+lava, goal+wall, lava+wall
    lengths = {(g['state'], tuple(g['end'])) for g in quirk['geoms']}
    assert (11, (26, 28)) in lengths and (11, (34, 26)) in lengths       # 0.125 * 20 = 2.5 -> 2, 0.375 * 20 = 7.5 -> 8
    assert not any(g['state'] == 8 and g['end'][1] > 26 for g in quirk['geoms'])  # p = 0.099 (UP) is not drawn
    assert sum(g['state'] == 10 for g in quirk['geoms']) == 4                       # p = 0.1 exactly is


def test_rasterisation_rule_at_the_reference_tile_size():
    """At 52 pixels a one-hot UP arrow: a 2-pixel shaft of 20 rows above the centre, then the 5-row head narrowing from 10 to 2."""
    geoms = R.arrow_geoms([[1.0, 0, 0, 0]], 1, [], [], [])
    m = R.rasterise(geoms, 52)
    rows = m.sum(axis=1)
    assert rows[26:].sum() == 0 and list(rows[6:26]) == [2] * 20 and list(rows[1:6]) == [2, 4, 6, 8, 10] and rows[0] == 0
    assert m[:, 25:27].sum() == 50 and m.sum() == 20 * 2 + 30
    for px in (3, 16, 26):
        assert R.rasterise(geoms, px)[:px // 2].sum() > 0 and R.rasterise(geoms, px)[(px + 1) // 2:].sum() == 0
    assert np.array_equal(R.rasterise(R.arrow_geoms([[0, 0, 1.0, 0]], 1, [], [], []), 52), m[::-1])
    assert np.array_equal(R.rasterise(R.arrow_geoms([[0, 1.0, 0, 0]], 1, [], [], []), 52), m.T[:, ::-1])


def test_trail_quads_equal_the_reference():
    """tests/golden/trail.json: what the reference's own trail loop (rendering.py:287-311, lifted with ast) emitted after the
    reference env's own step() / reset() calls.  The oracle env keeps the same last_n_states (cap 500, emptied by reset), and
    oracle/render.trail_quads restates the loop: same tiles, same alphas (float64, exactly), same order."""
    import json
    import os

    from oracle import render as R
    from oracle.ref_env import OracleGridUniverseEnv
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'trail.json')))
    assert (gold['alpha0'], gold['discount']) == (R.TRAIL_ALPHA0, R.TRAIL_DISCOUNT) and len(gold['cases']) >= 5
    for case in gold['cases']:
        env = OracleGridUniverseEnv(grid_shape=(case['W'], case['H']), initial_state=list(case['starts']), goal_states=list(case['goals']),
                                    lava_states=list(case['lava']), walls=list(case['walls']))
        env.reset()
        env.current_state = env.previous_state = env.initial_state = case['start_state']  # (a multi-start level draws it from the global RNG)
        for a in case['actions']:
            env.reset() if a < 0 else env.step(a)
        world = {(int(x), int(y)): s for s, (x, y) in enumerate(env.world)}
        kept = [world[(int(x), int(y))] for x, y in env.last_n_states]
        assert env.current_state == case['current_state'] and kept == case['last_n_states'], case['name']
        quads = R.trail_quads(kept, env.current_state)
        assert [[s, a] for s, a in quads] == case['quads'], case['name']
        assert [list(c) for c in R.TRAIL_CORNERS] == case['corner_colours']
    # the blend rule at its edges: alpha 0 leaves the tile, the corner pixels of a large tile approach the corner colours
    img = np.full((52, 52, 3), 200, np.uint8)
    assert np.array_equal(R.blend_trail(img.copy(), 1, 52, [(0, 0.0)]), img)
    full = R.blend_trail(np.zeros((52, 52, 3), np.uint8), 1, 52, [(0, 1.0)])
    assert tuple(full[51, 0]) == (253, 2, 2) and tuple(full[51, 51]) == (253, 253, 0) and tuple(full[0, 51]) == (2, 253, 2) and tuple(full[0, 0]) == (2, 2, 250)
