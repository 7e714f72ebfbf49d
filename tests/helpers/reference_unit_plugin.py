"""pytest plug-in used by scripts/run_reference_unit_tests.py (build container only).

Points the reference's OWN unit tests at this package's backend class without touching a file of
the reference: before anything of `tests/unit_tests` is collected, the names the reference's tests
bind their backends from - `PySDM.backends.CPU / Numba`
(tests/unit_tests/conftest.py:4-17 `backend_class`, `backend_instance`; most files under
dynamics/collisions import `CPU` directly) - are bound to
`as_pysdm_backend(OracleBackend)`: the class `HIP` is (pysdm_amd/backends/pysdm_shaped.py), over the
CPU checker's implementation of include/sdm_hip.h (there is no GPU next to the reference).
Cases parametrised with the reference's own GPU class are deselected (not the subject).
Outcomes are collected per test id for the tracked report.
"""
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUTCOMES = {}      # nodeid -> (outcome, one line)
DESELECTED = []    # nodeids dropped as duplicates


def _bind():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import numpy as np  # pylint: disable=import-outside-toplevel

    # the reference's tests are written against NumPy 1.x (setup.py:25-38 pins 1.24 / 1.26 for its
    # CI): test_sdm_breakup.py:101,181,... build `Box(dv=np.NaN, dt=np.NaN)`.  NumPy 2 (this image)
    # dropped the alias; restored for the run, like the import-only stand-ins for numba / pint
    if not hasattr(np, "NaN"):
        np.NaN = np.nan
    backends = importlib.import_module("PySDM.backends")
    from oracle.backend import OracleBackend  # pylint: disable=import-outside-toplevel
    from pysdm_amd.pysdm_plugin import as_pysdm_backend  # pylint: disable=import-outside-toplevel

    plugged = as_pysdm_backend(OracleBackend)
    original = {name: getattr(backends, name) for name in ("CPU", "Numba", "GPU", "ThrustRTC")}
    # CPU (= Numba) becomes this package's class.  GPU (= ThrustRTC) stays the reference's own: its
    # tests switch cases off by identity (`if backend_class is ThrustRTC: pytest.skip("TODO #330")`,
    # dynamics/collisions/test_sdm_single_cell.py:270-271 and a dozen more) - bound to the same
    # class, those skips would swallow the very cases this run is about.  Everything parametrised
    # with the reference's GPU class (or an instance of it) is deselected below: not the subject.
    backends.CPU = backends.Numba = backends.HIP = plugged
    return plugged, original


PLUGGED, ORIGINAL = _bind()


def _is_reference_gpu(value):
    gpu = ORIGINAL["GPU"]
    return value is gpu or isinstance(value, gpu)


def pytest_collection_modifyitems(config, items):
    keep, drop = [], []
    for item in items:
        callspec = getattr(item, "callspec", None)
        if callspec is not None and any(_is_reference_gpu(v) for v in callspec.params.values()):
            drop.append(item)
        else:
            keep.append(item)
    if drop:
        DESELECTED.extend(i.nodeid for i in drop)
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def pytest_runtest_logreport(report):
    if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
        line = ""
        if report.outcome != "passed":
            text = str(report.longrepr)
            if hasattr(report.longrepr, "reprcrash") and report.longrepr.reprcrash is not None:
                line = report.longrepr.reprcrash.message.splitlines()[0]
            elif isinstance(report.longrepr, tuple):
                line = str(report.longrepr[-1])
            else:
                line = text.strip().splitlines()[-1] if text.strip() else ""
        outcome = report.outcome
        if hasattr(report, "wasxfail"):
            outcome = "xfailed" if report.outcome == "skipped" else "xpassed"
            line = report.wasxfail or line
        OUTCOMES[report.nodeid] = (outcome, line[:300])
