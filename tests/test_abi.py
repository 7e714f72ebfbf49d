"""The C-ABI library loads and exports every symbol include/sdm_hip.h declares (no compute calls:
runs without a GPU); struct layouts of the ctypes mirror match the C header; the header-driven
binding understands every prototype; the oracle implements the same header."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest

from pysdm_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    library = abi.hip_library()
    names = abi.declared_symbols()
    assert len(names) >= 50
    for name in names:
        assert hasattr(library.cdll, name), name
    assert library.cdll.sdm_abi_version() == 1


def test_struct_layouts_match_the_header():
    source = (
        '#include "include/sdm_hip.h"\n#include <stdio.h>\n'
        'int main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(sdm_step_cfg), sizeof(sdm_step_state),'
        " sizeof(sdm_step_result), sizeof(sdm_disp_cfg), sizeof(sdm_disp_state),"
        " sizeof(sdm_disp_shard));return 0;}\n"
    )
    with tempfile.TemporaryDirectory() as tmp:
        src, exe = os.path.join(tmp, "sz.c"), os.path.join(tmp, "sz")
        with open(src, "w", encoding="utf-8") as handle:
            handle.write(source)
        subprocess.check_call(["gcc", "-I", ROOT, src, "-o", exe], cwd=ROOT)
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(t) for t in (abi.StepCfg, abi.StepState, abi.StepResult,
                                                abi.DispCfg, abi.DispState, abi.DispShard)]


def test_error_reporting_without_a_gpu_call():
    cdll = abi.hip_library().cdll
    # a NULL ctx is rejected before anything touches the device
    assert cdll.sdm_ctx_set_stream(None, None) == -1
    assert "bad argument" in abi.hip_library().last_error()


def test_every_prototype_is_understood_by_the_binding():
    table = abi.parse_header()
    kinds = {param.kind for _, params in table.values() for param in params}
    assert kinds == {"ctx", "scalar", "pointer", "host_array"}
    first = {name: params[0].kind for name, (_, params) in table.items() if params}
    not_ctx_first = {n for n, kind in first.items() if kind != "ctx"}
    assert not_ctx_first == {"sdm_ctx_create", "sdm_phase_name", "sdm_comm_unique_id"}
    cfg = dict((p.name, p) for p in table["sdm_collision_step"][1])
    assert cfg["cfg"].base == "sdm_step_cfg" and cfg["flags"].kind == "scalar"


def test_the_oracle_implements_the_same_header(oracle_engine):
    """one header, two libraries: product (device pointers) and checker (host pointers)"""
    for name in abi.declared_symbols():
        assert hasattr(oracle_engine.library.cdll, name), name
    with pytest.raises(TypeError):  # operand checks of the binding, before any call
        oracle_engine.call("sdm_identity_index", np.zeros(4), 4)
    with pytest.raises(ValueError):
        oracle_engine.call("sdm_identity_index", np.zeros((4, 2), dtype=np.int64)[:, 0], 4)


def test_pysdm_shaped_backends_are_one_class(oracle_backend_class):
    """`HIP` and the oracle's backend come from the same factory: every method PySDM may call
    exists on both with the same parameters, by construction"""
    import inspect  # pylint: disable=import-outside-toplevel

    from pysdm_amd.backends.hip import HIP  # pylint: disable=import-outside-toplevel

    def public(cls):
        return {name: inspect.signature(member) for name, member
                in inspect.getmembers(cls, inspect.isfunction) if not name.startswith("_")}

    assert public(HIP) == public(oracle_backend_class)
    assert {"shuffle_local", "find_pairs", "compute_gamma", "collision_coalescence_breakup",
            "moments", "calculate_displacement"} <= set(public(HIP))


def test_random_sector_calibration_checksum(oracle_engine):
    """sdm_calib_random_sectors (measurement entry behind bench.py's random-sector ceiling): the
    checksum is the sum over the records the header says are read - checked here against an
    independent numpy evaluation of the same hash (the GPU test compares the kernel with it)"""
    import ctypes  # pylint: disable=import-outside-toplevel

    table, reads, reps = 1000, 5000, 3
    ms, checksum = ctypes.c_double(), ctypes.c_uint64()
    oracle_engine.call("sdm_calib_random_sectors", table, reads, reps, ms, checksum)
    assert checksum.value == expected_calibration_checksum(table, reads, reps)


def expected_calibration_checksum(table, reads, reps):
    total = np.uint64(0)
    with np.errstate(over="ignore"):
        for r in range(reps):
            x = np.arange(reads, dtype=np.uint64) ^ np.uint64(((r + 1) * 0x100000001B3) % 2**64)
            x = x + np.uint64(0x9E3779B97F4A7C15)
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            x = x ^ (x >> np.uint64(31))
            at = x % np.uint64(table)
            total += np.sum(at + (np.uint64(2) * at + np.uint64(1)), dtype=np.uint64)
    return int(total)
