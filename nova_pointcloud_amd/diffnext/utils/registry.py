"""Name -> factory registry for the architecture tables (reference diffnext/utils/registry.py:22-54)."""
import functools
from collections import OrderedDict


class Registry(object):
    """Maps architecture names to partially-applied builder functions."""

    def __init__(self, name):
        self.name, self.registry = name, OrderedDict()

    def has(self, key) -> bool:
        return key in self.registry

    def register(self, name, func=None, **kwargs):
        keys = list(name) if isinstance(name, (tuple, list)) else [name]

        def bind(fn):
            self.registry.update({k: functools.partial(fn, **kwargs) for k in keys})
            return fn

        return bind if func is None else bind(func)

    def get(self, name, default=None):
        if name is None:
            return None
        if name in self.registry:
            return self.registry[name]
        if default is not None:
            return default
        raise KeyError("`%s` is not registered in <%s>." % (name, self.name))  # message as the reference, :48

    def try_get(self, name):
        return self.registry.get(name, None)
