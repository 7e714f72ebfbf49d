"""TEST INFRASTRUCTURE, NOT PRODUCT CODE: the CPU oracle as an `Engine`.

oracle/sdm_oracle.c restates the reference's algorithm function by function (serial, strict IEEE,
each function citing the reference lines it follows); oracle/sdm_oracle_abi.c exports it through
the product's own header, include/sdm_hip.h, with host pointers.  `OracleEngine` binds that
library with the header-driven binding of pysdm_amd.abi and hands out numpy arrays, so the tests
run product and checker through identical host code and compare.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
pysdm_amd/ does.
"""
import ctypes
import os
import subprocess

import numpy as np

from pysdm_amd import abi
from pysdm_amd.engine import Engine

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(_HERE, name) for name in ("sdm_oracle_abi.c", "sdm_oracle.c")]
HEADER = abi.HEADER_PATH
# the transcendental functions are shared with the product (one implementation, same bits)
MATH = [os.path.join(os.path.dirname(_HERE), "pysdm_amd", "csrc", name)
        for name in ("sdm_math.h", "sdm_math_tables.h")]
LIB_PATH = os.path.join(_HERE, "libsdm_oracle.so")
LIB_PATH_OMP = os.path.join(_HERE, "libsdm_oracle_omp.so")
_FLAGS = ["-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
          "-fvisibility=hidden"]


def build(force=False):
    """compiles the serial checker and its OpenMP twin (same source; `prange` loops of the
    reference's Numba backend become `omp parallel for`) - used by bench.py's cpu_baseline"""
    newest = max(os.path.getmtime(p) for p in SOURCES + [HEADER] + MATH)
    for path, extra in ((LIB_PATH, []), (LIB_PATH_OMP, ["-fopenmp"])):
        if force or not os.path.exists(path) or os.path.getmtime(path) < newest:
            subprocess.check_call(["gcc", *_FLAGS, *extra, "-o", path, SOURCES[0], "-lm"])
    return LIB_PATH


class OracleEngine(Engine):
    name = "oracle"
    _instances = {}

    def __init__(self, threads=1):
        build()
        self.library = abi.Library(LIB_PATH if threads == 1 else LIB_PATH_OMP, "the CPU oracle")
        self.handle = abi.c_ptr()
        self.library.check(self.library.cdll.sdm_ctx_create(ctypes.byref(self.handle),
                                                            abi.c_int(0)))
        self.threads = threads
        if threads != 1:
            self.library.cdll.oracle_set_threads(ctypes.c_int(threads))

    @classmethod
    def get(cls, threads=1):
        if threads not in cls._instances:
            cls._instances[threads] = cls(threads)
        return cls._instances[threads]

    @staticmethod
    def empty(shape, dtype):
        return np.empty(shape, dtype=dtype)

    @staticmethod
    def upload(array):
        return np.array(array, copy=True, order="C")

    @staticmethod
    def download(array):
        return np.array(array, copy=True)
