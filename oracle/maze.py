"""CPU restatement of the reference maze generator (test infrastructure).

Follows TheMTank/GridUniverse `core/envs/maze_generation.py`:
  recursive_backtracker  :41-101   depth-first carve in strides of two
  create_random_maze     :104-149  bool grid -> '#'/'o' rows, random 'x' and 'G'

What matters for parity is the ORDER in which the two process-global RNGs are
consumed: numpy's legacy global RandomState gives the DFS
origin (x first, then y; `random_integers(0, n-1)` == `randint(0, n)`), stdlib
`random.choice` picks each neighbour, stdlib `random.sample` places 'x' and 'G'.
The reference additionally opens a matplotlib figure and prints the maze; both
are side effects without influence on the result and are not restated.
Pinned by golden G3 (tests/golden/mazes.json) in tests/test_oracle_env.py.
"""
import random

import numpy as np


def carve(width, height):
    """Return bool[height, width], True = wall (maze_generation.py:41-101)."""
    wall = np.ones((height, width), dtype=bool)
    seen = np.zeros((height, width), dtype=bool)
    cx = int(np.random.randint(0, width))   # :53 first draw  -> x
    cy = int(np.random.randint(0, height))  # :53 second draw -> y
    seen[cy, cx] = True
    trail = []
    while not seen.all():  # :57 (never satisfied on even sizes; exit is the empty-trail break)
        nbrs = []  # candidate order fixed by :61-68: +x, -x, +y, -y
        if cx + 2 < width and not seen[cy, cx + 2]:
            nbrs.append((cx + 2, cy))
        if cx - 2 >= 0 and not seen[cy, cx - 2]:
            nbrs.append((cx - 2, cy))
        if cy + 2 < height and not seen[cy + 2, cx]:
            nbrs.append((cx, cy + 2))
        if cy - 2 >= 0 and not seen[cy - 2, cx]:
            nbrs.append((cx, cy - 2))
        if nbrs:
            nx, ny = random.choice(nbrs)  # :72
            trail.append((cx, cy))
            wall[(ny + cy) // 2, (nx + cx) // 2] = False  # :77-82 carve wall, neighbour, current
            wall[ny, nx] = False
            wall[cy, cx] = False
            cx, cy = nx, ny
            seen[cy, cx] = True
        elif trail:
            cx, cy = trail.pop()  # :90-93
        else:
            break  # :94-95
    return wall


def create_random_maze(width, height):
    """Rows of single characters with one 'x' and one 'G' (maze_generation.py:104-149)."""
    wall = carve(width, height)
    rows = [['#' if wall[y, x] else 'o' for x in range(width)] for y in range(height)]
    open_cells = [y * width + x for y in range(height) for x in range(width) if not wall[y, x]]
    start, goal = random.sample(open_cells, 2)  # :134
    rows[start // width][start % width] = 'x'
    rows[goal // width][goal % width] = 'G'
    return rows
