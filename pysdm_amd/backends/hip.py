"""`HIP`: the MI355X backend object - PySDM's backend interface for the collision path with every
method executed by hand-written HIP kernels (libsdm_hip.so, C ABI in include/sdm_hip.h).

    from pysdm_amd.backends import HIP
    backend = HIP(formulae)        # needs a GPU; raises otherwise (no CPU fallback)

See pysdm_shaped.py for the contract and pysdm_amd/pysdm_plugin.py for plugging it into PySDM.
"""
from ..engine import HipEngine
from .pysdm_shaped import backend_class_for

HIP = backend_class_for(
    HipEngine.get, "HIP",
    doc="PySDM-shaped backend over libsdm_hip.so (HIP kernels for gfx950); "
        "HIP(formulae=None, double_precision=True, device_index=None)")
Storage = HIP.Storage
Random = HIP.Random
