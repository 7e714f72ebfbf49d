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


def test_hip_backend_and_oracle_backend_are_interface_twins(oracle_backend_class):
    """every backend method the front-end may call exists on both with the same parameters:
    what the oracle passes under the unmodified reference front-end (test_reference_plugin.py)
    carries over to HIP"""
    import inspect  # pylint: disable=import-outside-toplevel

    from pysdm_amd.backends.hip import HIP, Storage  # pylint: disable=import-outside-toplevel
    from pysdm_amd.backends.storage_base import StorageBase  # pylint: disable=import-outside-toplevel

    def public(cls):
        return {name: member for name, member in inspect.getmembers(cls, callable)
                if not name.startswith("_")}

    hip, oracle = public(HIP), public(oracle_backend_class)
    device_only = {"make_collision_step", "collision_step", "straub_consts", "synchronize",
                   "displacement_step"}
    assert set(hip) - device_only == set(oracle) - device_only
    for name in set(hip) - device_only:
        if inspect.isclass(hip[name]):
            continue
        params = [[p for p in inspect.signature(side[name]).parameters if p != "self"]
                  for side in (hip, oracle)]
        assert params[0] == params[1], name
    assert issubclass(Storage, StorageBase) and issubclass(oracle_backend_class.Storage,
                                                           StorageBase)
