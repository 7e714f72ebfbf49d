"""physics subset needed by the collision path"""
from . import constants
from .constants import si
