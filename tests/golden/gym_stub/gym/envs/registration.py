_registry = {}


def register(id, entry_point=None, **kwargs):
    _registry[id] = (entry_point, kwargs)
