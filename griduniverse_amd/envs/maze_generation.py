"""Random-maze level source (host side, runs once per grid -- not a kernel).

Same public names and result as the reference's `core/envs/maze_generation.py`
(`recursive_backtracker` :41-101, `create_random_maze` :104-149): a depth-first
"recursive backtracker" that carves corridors in strides of two cells from a random
origin, then one random start 'x' and one random goal 'G' among the open cells.

Seeded grids must be IDENTICAL to the reference's (bit-exact trajectories start from
bit-identical grids), so the two process-global RNGs are consumed in the reference's
order: numpy's legacy global RandomState for the origin (x, then y;
the reference's deprecated `random_integers(0, n-1)` is `randint(0, n)`), stdlib
`random.choice` once per carve among the unvisited stride-2 neighbours listed in the
order +x, -x, +y, -y, and stdlib `random.sample(open_cells, 2)` for 'x' and 'G'.
Unlike the reference this module is silent (no prints) and opens no matplotlib figure.

Note (reference behaviour, kept): only cells sharing the origin's parity are ever
carved, so even sizes such as 32x32 or 64x64 get no guaranteed border and the loop
ends when the backtracking stack empties, not when every cell has been visited.
"""
import random

import numpy as np

_STRIDE2 = ((2, 0), (-2, 0), (0, 2), (0, -2))  # neighbour order of maze_generation.py:61-68


def recursive_backtracker(width=20, height=20):
    """bool[height, width], True = wall."""
    is_wall = np.ones((height, width), dtype=bool)
    visited = np.zeros((height, width), dtype=bool)
    x = int(np.random.randint(0, width))
    y = int(np.random.randint(0, height))
    visited[y, x] = True
    unvisited = width * height - 1
    stack = []
    while unvisited > 0:
        options = [(x + dx, y + dy) for dx, dy in _STRIDE2
                   if 0 <= x + dx < width and 0 <= y + dy < height and not visited[y + dy, x + dx]]
        if options:
            nx, ny = random.choice(options)
            stack.append((x, y))
            is_wall[y, x] = is_wall[(y + ny) // 2, (x + nx) // 2] = is_wall[ny, nx] = False
            x, y = nx, ny
            visited[y, x] = True
            unvisited -= 1
        elif stack:
            x, y = stack.pop()
        else:
            break
    return is_wall


def create_random_maze(width, height):
    """List of rows, each a list of 'o' '#' 'x' 'G' characters."""
    is_wall = recursive_backtracker(width, height)
    rows = np.where(is_wall, '#', 'o').tolist()
    open_cells = np.flatnonzero(~is_wall.ravel()).tolist()
    start, goal = random.sample(open_cells, 2)
    rows[start // width][start % width] = 'x'
    rows[goal // width][goal % width] = 'G'
    return rows
