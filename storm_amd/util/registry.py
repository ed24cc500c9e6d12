"""Name -> class registries (same behaviour as the reference's sgmse/util/registry.py:5-34:
double registration warns and replaces, unknown names raise ValueError)."""
import warnings
from typing import Callable


class Registry:
    def __init__(self, managed_thing: str):
        self.managed_thing = managed_thing
        self._registry = {}
        self._not_built = {}           # upstream names deliberately outside this engine's scope -> why

    def declare_out_of_scope(self, name: str, why: str):
        self._not_built[name] = why

    def register(self, name: str) -> Callable:
        def deco(cls):
            if name in self._registry:
                warnings.warn(f"{self.managed_thing} with name '{name}' doubly registered, old class will be replaced.")
            self._registry[name] = cls
            return cls
        return deco

    def get_by_name(self, name: str):
        if name in self._registry:
            return self._registry[name]
        if name in self._not_built:
            raise ValueError(f"{self.managed_thing} with name '{name}' is registered upstream but not built here: {self._not_built[name]}")
        raise ValueError(f"{self.managed_thing} with name '{name}' unknown.")

    def get_all_names(self):
        return list(self._registry.keys())
