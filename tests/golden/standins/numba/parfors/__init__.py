from . import parfor  # noqa: F401
