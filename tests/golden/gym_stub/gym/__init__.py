"""Minimal stand-in for the 2017-era `gym` package (golden-capture tooling only).

The reference (TheMTank/GridUniverse) imports gym only for method dispatch
(`step -> _step` etc.), `spaces.Discrete`, `utils.seeding.np_random`, and
`envs.registration.register`.  None of gym's arithmetic is on the hot path, so
this stub is sufficient to import and run the reference in this container in
order to capture golden vectors (tests/golden/make_golden.py).  It is NOT part of the
product and never travels with the package.
"""
from . import error, spaces  # noqa: F401
from .core import Env  # noqa: F401
