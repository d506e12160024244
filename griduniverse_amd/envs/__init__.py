from .griduniverse_env import GridUniverseEnv  # noqa: F401  (mirrors core/envs/__init__.py:1)
