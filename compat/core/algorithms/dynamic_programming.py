from griduniverse_amd.algorithms.dynamic_programming import *  # noqa: F401,F403
