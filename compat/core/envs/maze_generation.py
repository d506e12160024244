from griduniverse_amd.envs.maze_generation import create_random_maze, recursive_backtracker  # noqa: F401
