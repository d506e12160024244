from griduniverse_amd.envs.griduniverse_env import GridUniverseEnv  # noqa: F401
