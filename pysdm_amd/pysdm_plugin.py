"""Plugs the HIP backend into an unmodified PySDM installation.

PySDM picks its backend by object: `Builder(n_sd, backend=<instance>, environment=...)`
(PySDM/builder.py:25-38), insisting only that it is an instance of its own `BackendMethods`
(PySDM/particulator.py:22).  `HIP` of this package already carries every method, `Storage` and
`Random` the front-end calls, with PySDM's names and signatures; what remains is the base class:

    from pysdm_amd.pysdm_plugin import install
    HIP = install()                       # also reachable as PySDM.backends.HIP afterwards
    builder = Builder(n_sd, backend=HIP(Formulae(...)), environment=Box(...))

With PySDM's own `Coalescence`/`Collision`/`Breakup` dynamics the step then runs
method-by-method on the device.  The fused per-time-step route (one library call per step) comes
with this package's dynamics, which register with PySDM's Builder unchanged:

    from pysdm_amd.dynamics.collisions import Coalescence, Golovin

PySDM itself is imported lazily: this module loads (and fails loudly) only where PySDM exists.
"""
import importlib


def _pysdm_backend_methods():
    try:
        module = importlib.import_module("PySDM.backends.impl_common.backend_methods")
    except ImportError as error:
        raise ImportError("pysdm_amd.pysdm_plugin needs an importable PySDM") from error
    return module.BackendMethods


def as_pysdm_backend(backend_class):
    """`backend_class` (HIP; the tests pass the CPU oracle, its interface twin) as a class PySDM's
    Particulator accepts"""
    base = _pysdm_backend_methods()
    if issubclass(backend_class, base):
        return backend_class
    return type(backend_class.__name__, (backend_class, base), {
        "__doc__": backend_class.__doc__, "__module__": backend_class.__module__})


def install():
    """registers `PySDM.backends.HIP` next to CPU / GPU (PySDM/backends/__init__.py:75-83)"""
    from .backends.hip import HIP  # pylint: disable=import-outside-toplevel

    plugged = as_pysdm_backend(HIP)
    backends = importlib.import_module("PySDM.backends")
    setattr(backends, "HIP", plugged)
    return plugged
