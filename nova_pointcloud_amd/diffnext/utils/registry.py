"""Architecture-name registry (the surface of reference diffnext/utils/registry.py:22-54).

`register(name, fn, **preset)` stores `fn` with some keyword arguments pre-bound; `get(name)` hands the bound
builder back and raises KeyError for unknown names (message format of the reference, :48).
"""
from collections import OrderedDict
from functools import partial


class Registry(object):
    def __init__(self, name):
        self.name = name
        self.registry = OrderedDict()  # name -> functools.partial

    def register(self, name, func=None, **kwargs):
        """Direct call when `func` is given, decorator otherwise. `name` may be a list of aliases."""
        aliases = [name] if not isinstance(name, (tuple, list)) else list(name)

        def _store(builder):
            for alias in aliases:
                self.registry[alias] = partial(builder, **kwargs)
            return builder

        if func is not None:
            return _store(func)
        return _store

    def has(self, key) -> bool:
        return key in self.registry

    def try_get(self, name):
        return self.registry.get(name)

    def get(self, name, default=None):
        if name is None:
            return None
        found = self.try_get(name)
        if found is None:
            if default is None:
                raise KeyError("`%s` is not registered in <%s>." % (name, self.name))
            return default
        return found
