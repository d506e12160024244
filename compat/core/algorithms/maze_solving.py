from griduniverse_amd.algorithms.maze_solving import *  # noqa: F401,F403
