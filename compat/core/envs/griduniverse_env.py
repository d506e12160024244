from griduniverse_amd.envs.griduniverse_env import GridUniverseEnv, UnsupportedMode  # noqa: F401
