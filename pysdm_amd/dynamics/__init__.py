"""Dynamics on and next to the collision path."""
from copy import deepcopy


def _instantiate(self, *, builder):
    copy = deepcopy(self)
    copy.register(builder=builder)
    return copy


def register_dynamic():
    """dynamics are deep-copied when a builder builds them, so that one instance can be handed
    to several builders (PySDM/dynamics/impl/register_dynamic.py:7-22)"""

    def decorator(cls):
        if hasattr(cls, "instantiate"):
            assert cls.instantiate is _instantiate
        else:
            setattr(cls, "instantiate", _instantiate)
        return cls

    return decorator
