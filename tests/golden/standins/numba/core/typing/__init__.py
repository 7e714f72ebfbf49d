from . import arraydecl  # noqa: F401
