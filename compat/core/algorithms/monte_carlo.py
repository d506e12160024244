from griduniverse_amd.algorithms.monte_carlo import *  # noqa: F401,F403
