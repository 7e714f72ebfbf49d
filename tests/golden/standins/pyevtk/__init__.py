"""Stand-in for `pyevtk` (golden generation only); exporters are off the collision path."""
from . import hl, vtk  # noqa: F401
