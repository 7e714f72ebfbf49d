"""The C-ABI library loads and exports every symbol include/sdm_hip.h declares (no compute calls:
runs without a GPU); struct layouts of the ctypes mirror match the C header."""
import ctypes
import os
import subprocess
import tempfile

from pysdm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 38
    for name in names:
        assert hasattr(lib, name), name
    assert lib.sdm_abi_version() == 1


def test_struct_layouts_match_the_header():
    source = (
        '#include "include/sdm_hip.h"\n#include <stdio.h>\n'
        'int main(){printf("%zu %zu %zu\\n", sizeof(sdm_step_cfg), sizeof(sdm_step_state),'
        " sizeof(sdm_step_result));return 0;}\n"
    )
    with tempfile.TemporaryDirectory() as tmp:
        src, exe = os.path.join(tmp, "sz.c"), os.path.join(tmp, "sz")
        with open(src, "w", encoding="utf-8") as handle:
            handle.write(source)
        subprocess.check_call(["gcc", "-I", ROOT, src, "-o", exe], cwd=ROOT)
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(_lib.StepCfg), ctypes.sizeof(_lib.StepState),
                     ctypes.sizeof(_lib.StepResult)]


def test_error_reporting_without_a_gpu_call():
    lib = _lib.load()
    # a NULL ctx is rejected before anything touches the device
    assert lib.sdm_ctx_set_stream(None, None) == -1
    assert b"bad argument" in lib.sdm_last_error()
