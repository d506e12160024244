from griduniverse_amd.algorithms.utils import *  # noqa: F401,F403
