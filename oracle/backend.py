"""TEST INFRASTRUCTURE, NOT PRODUCT CODE: the CPU oracle behind PySDM's backend interface.

`OracleBackend` is the very class `pysdm_amd.backends.HIP` is (pysdm_shaped.backend_class_for:
the reference's backend-method names on Storages), bound to the oracle engine instead of the HIP
engine, so every method forwards to oracle/sdm_oracle.c through the shared header.  It exists so
that tests can (i) pin the oracle against the goldens generated from the reference, (ii) exercise
the PySDM-facing layer without a GPU and (iii) plug it into the real PySDM front-end in the build
container (tests/test_reference_plugin.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.
"""
from pysdm_amd.backends.pysdm_shaped import backend_class_for

from .engine import OracleEngine, build  # noqa: F401  (build re-exported for __graft_entry__)

OracleBackend = backend_class_for(
    OracleEngine.get, "OracleBackend",
    doc="PySDM-shaped backend over the CPU oracle (numpy arrays, serial C)")
