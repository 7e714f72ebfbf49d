"""Dynamics on and next to the collision path."""
import copy


def builder_owned(cls):
    """Class decorator: gives a dynamic the `instantiate(builder=...)` hook a Builder calls when it
    builds.  The builder then works on its own deep copy, registered with it, so the object the
    user created can be handed to several builders (semantics of
    PySDM/dynamics/impl/register_dynamic.py; PySDM's own Builder calls the same hook, which is how
    this package's dynamics also run under it)."""

    def instantiate(self, *, builder):
        own = copy.deepcopy(self)
        own.register(builder=builder)
        return own

    cls.instantiate = instantiate
    return cls
