"""pysdm_amd -- MI355X (gfx950) implementation of the SDM collision / coalescence / breakup hot
path behind PySDM's backend interface: hand-written HIP kernels in libsdm_hip.so (C ABI in
include/sdm_hip.h) + the host-side mirror of the reference interface for that path.
"""
from .formulae import Formulae
from .particulator import Builder, Particulator

__all__ = ["Builder", "Particulator", "Formulae"]
__version__ = "0.1.0"
