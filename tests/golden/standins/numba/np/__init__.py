from . import arrayobj  # noqa: F401
