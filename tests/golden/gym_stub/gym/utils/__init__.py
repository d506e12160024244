from . import seeding  # noqa: F401


def reraise(*a, **k):
    raise
