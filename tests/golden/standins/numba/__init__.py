"""Stand-in for `numba` used ONLY by tests/golden/gen_golden.py in the build container.

The reference (PySDM) is pure Python; its Numba backend bodies are plain Python functions that
the real numba would JIT.  numba is not installed here (and cannot be), so this shim makes
`njit` the identity decorator -- the same execution mode as the reference's own CI job that
runs with NUMBA_DISABLE_JIT=1.  Contains no reference code.
"""
import contextlib
import numpy as _np

from . import config, types, cuda, typed, extending  # noqa: F401
from .core import errors  # noqa: F401
from . import core, parfors  # noqa: F401

float64 = _np.float64
int64 = _np.int64
prange = range


def _identity_decorator(*args, **_kwargs):
    if len(args) >= 1 and callable(args[0]):
        func = args[0]
        try:
            func.py_func = func
        except AttributeError:
            pass
        return func

    def wrap(func):
        try:
            func.py_func = func
        except AttributeError:
            pass
        return func

    return wrap


njit = _identity_decorator
jit = _identity_decorator


def vectorize(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return _np.vectorize(args[0])

    def wrap(func):
        return _np.vectorize(func)

    return wrap


@contextlib.contextmanager
def objmode(*_args, **_kwargs):
    yield


def get_num_threads():
    return 1


def set_num_threads(_n):
    pass
