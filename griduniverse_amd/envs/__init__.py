"""Env classes.  `from griduniverse_amd.envs import GridUniverseEnv` mirrors the reference's package layout."""
from . import griduniverse_env as _module

GridUniverseEnv = _module.GridUniverseEnv
