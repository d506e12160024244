"""Structural invariants of recursive-backtracker mazes, checked (a) on mazes captured from the REFERENCE
generator and (b) on the CPU restatement of the on-device generator -- the two draw from different RNGs, so
they are compared by structure, not cell by cell."""
import numpy as np
import pytest

from oracle import c_oracle as C
from tests import _golden as G


def check_maze_structure(wall, start, goal):
    """wall: bool[H, W].  Rooms = open cells sharing the parity of some origin; every open cell is a room or a
    passage between two rooms two apart; the open cells form a tree (|open| = 2*|rooms| - 1, connected)."""
    H, W = wall.shape
    open_cells = np.argwhere(~wall)
    assert len(open_cells) >= 2 and start != goal and not wall.ravel()[start] and not wall.ravel()[goal]
    parities = {}
    for py in (0, 1):
        for px in (0, 1):
            rooms = [(y, x) for y, x in open_cells if y % 2 == py and x % 2 == px]
            others = [(y, x) for y, x in open_cells if not (y % 2 == py and x % 2 == px)]
            if all((y % 2 == py) != (x % 2 == px) for y, x in others) and len(open_cells) == 2 * len(rooms) - 1:
                parities[(py, px)] = rooms
    assert parities, 'open cells are not rooms of one parity class plus single passages'
    # connectivity by flood fill over 4-neighbours
    seen, todo = set(), [tuple(open_cells[0])]
    while todo:
        y, x = todo.pop()
        if (y, x) in seen:
            continue
        seen.add((y, x))
        for dy, dx in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            ny, nx = y + dy, x + dx
            if 0 <= ny < H and 0 <= nx < W and not wall[ny, nx] and (ny, nx) not in seen:
                todo.append((ny, nx))
    assert len(seen) == len(open_cells), 'maze is not connected'
    # every reachable room of the parity class was carved (the DFS exhausts them)
    (py, px), rooms = next(iter(parities.items()))
    assert len(rooms) == len(range(py, H, 2)) * len(range(px, W, 2))


@pytest.mark.parametrize('key', sorted(G.load_json('mazes.json')))
def test_reference_mazes_have_the_structure(key):
    m = G.load_json('mazes.json')[key]
    wall = np.array([[c == '#' for c in row] for row in m['rows']])
    check_maze_structure(wall, m['start'][0], m['goal'][0])
    assert wall.sum() == m['n_walls']


@pytest.mark.parametrize('W,H', [(8, 8), (11, 11), (7, 5), (5, 7), (32, 32), (64, 64), (21, 13), (4, 1), (1, 6)])
def test_restated_device_generator_has_the_structure(W, H):
    counts = set()
    for gid in range(12):
        wall, start, goal = C.generate_maze(77, gid, W, H)
        check_maze_structure(wall.reshape(H, W), start, goal)
        counts.add(int(wall.sum()))
    ref = {m['n_walls'] for k, m in G.load_json('mazes.json').items() if (m['W'], m['H']) == (W, H)}
    if ref and W % 2 == 0 and H % 2 == 0:
        assert counts == ref  # even sizes: the wall count does not depend on the origin (32x32 -> 513, 64x64 -> 2049)


def test_generator_is_keyed_by_seed_and_grid_id():
    a = C.generate_maze(1, 0, 16, 16)
    assert all(np.array_equal(x, y) for x, y in zip(a, C.generate_maze(1, 0, 16, 16)))
    assert not np.array_equal(a[0], C.generate_maze(1, 1, 16, 16)[0])
    assert not np.array_equal(a[0], C.generate_maze(2, 0, 16, 16)[0])
    with pytest.raises(ValueError):
        C.generate_maze(1, 0, 2, 2)
