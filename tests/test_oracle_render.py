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
