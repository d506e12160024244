"""griduniverse_amd -- MI355X-native vectorised GridUniverse step/reset engine.

    GridUniverseEnv   drop-in for the reference's core.envs.griduniverse_env.GridUniverseEnv (N = 1)
    VecGridUniverse   N lock-stepped instances, array in / array out, fused rollout(T)
    Engine            numpy wrapper of one libgu handle (include/gu.h)
    GridSpec          grid lists -> row bit-planes

Compute runs only in libgu.so (hand-written HIP for gfx950); importing this package
does not load it, so host-side logic works without a GPU.
"""
from .envs.griduniverse_env import GridUniverseEnv, UnsupportedMode  # noqa: F401
from .grid import GridSpec  # noqa: F401
from ._lib import GuError  # noqa: F401

__version__ = '0.1.0'


def __getattr__(name):  # lazy: these import the ctypes binding
    if name == 'Engine':
        from .engine import Engine
        return Engine
    if name == 'VecGridUniverse':
        from .vec_env import VecGridUniverse
        return VecGridUniverse
    raise AttributeError(name)
