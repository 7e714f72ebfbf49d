"""pytest configuration: `gpu` marker; make sure both native libraries are built."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    import __graft_entry__  # pylint: disable=import-outside-toplevel

    __graft_entry__.build_if_missing()


@pytest.fixture(scope="session")
def oracle_engine():
    from oracle.engine import OracleEngine  # pylint: disable=import-outside-toplevel

    return OracleEngine.get()


@pytest.fixture(scope="session")
def oracle_backend_class():
    from oracle.backend import OracleBackend  # pylint: disable=import-outside-toplevel

    return OracleBackend


def _need_gpu():
    import torch  # pylint: disable=import-outside-toplevel

    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.fixture(scope="session")
def hip_engine():
    _need_gpu()
    from pysdm_amd.engine import HipEngine  # pylint: disable=import-outside-toplevel

    return HipEngine.get()


@pytest.fixture(scope="session")
def hip_backend_class():
    _need_gpu()
    from pysdm_amd.backends import HIP  # pylint: disable=import-outside-toplevel

    return HIP
