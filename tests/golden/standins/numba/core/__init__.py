from . import cgutils, errors, typing  # noqa: F401
