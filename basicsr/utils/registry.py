"""Name -> class registries: the plugin boundary of the reference
(reference basicsr/utils/registry.py:4-92; same public behaviour: ``register`` as decorator or
call, ``get`` with the ``<name>_basicsr`` fallback and a KeyError, ``in``, iteration, ``keys``)."""
from __future__ import annotations


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map: dict = {}

    # -- registration ---------------------------------------------------------------------
    def _add(self, obj, suffix=None):
        key = obj.__name__ if not isinstance(suffix, str) else f"{obj.__name__}_{suffix}"
        if key in self._obj_map:
            raise AssertionError(f"An object named '{key}' was already registered in '{self._name}' registry!")
        self._obj_map[key] = obj
        return obj

    def register(self, obj=None, suffix=None):
        if obj is not None:
            self._add(obj, suffix)
            return None
        return lambda target: self._add(target, suffix)

    # -- lookup -----------------------------------------------------------------------------
    def get(self, name: str, suffix: str = "basicsr"):
        hit = self._obj_map.get(name)
        if hit is None:
            alt = f"{name}_{suffix}"
            hit = self._obj_map.get(alt)
            print(f"Name {name} is not found, use name: {alt}!")
        if hit is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return hit

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def keys(self):
        return self._obj_map.keys()


DATASET_REGISTRY = Registry("dataset")
ARCH_REGISTRY = Registry("arch")
MODEL_REGISTRY = Registry("model")
LOSS_REGISTRY = Registry("loss")
METRIC_REGISTRY = Registry("metric")
