import importlib

from .. import error

registry = {}


def register(id, entry_point=None, **kwargs):
    registry[id] = (entry_point, kwargs)


def make(id, **kwargs):
    if id not in registry:
        raise error.UnregisteredEnv(id)
    entry_point, _ = registry[id]
    mod, cls = entry_point.split(":")
    return getattr(importlib.import_module(mod), cls)(**kwargs)
