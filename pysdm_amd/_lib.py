"""ctypes binding of libsdm_hip.so (C ABI: include/sdm_hip.h).  No fallback: a missing or
incomplete library is an ImportError, a failing call is a RuntimeError with the library's message.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDM_HIP_LIB: another build of the same library (tuning variants); still no fallback
LIB_PATH = os.environ.get("SDM_HIP_LIB") or os.path.join(_HERE, "libsdm_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sdm_hip.h")

c_i64 = ctypes.c_int64
c_f64 = ctypes.c_double
c_int = ctypes.c_int
c_ptr = ctypes.c_void_p
c_u64 = ctypes.c_uint64


class StepCfg(ctypes.Structure):  # == sdm_step_cfg
    _fields_ = [
        ("n_sd", c_i64), ("n_cell", c_i64), ("n_attr", c_i64),
        ("dt", c_f64), ("dv", c_f64), ("dt_min", c_f64), ("dt_max", c_f64),
        ("adaptive", ctypes.c_int32), ("substeps", ctypes.c_int32),
        ("croupier_local", ctypes.c_int32), ("optimized_random", ctypes.c_int32),
        ("enable_breakup", ctypes.c_int32), ("handle_all_breakups", ctypes.c_int32),
        ("kernel", ctypes.c_int32), ("ec", ctypes.c_int32), ("frag", ctypes.c_int32),
        ("mass_attr", ctypes.c_int32),
        ("kernel_param", c_f64 * 2), ("ec_param", c_f64 * 2), ("eb_const", c_f64),
        ("frag_param", c_f64 * 2), ("frag_vmin", c_f64), ("frag_nfmax", c_f64),
        ("rho_w", c_f64), ("sgm_w", c_f64), ("straub_consts", c_f64 * 6),
        ("berry_params", c_f64 * 13), ("berry_unit", c_f64),
        ("kernel_berry_params", c_f64 * 13), ("kernel_berry_unit", c_f64),
        ("max_multiplicity", c_i64), ("rng_state_inc", c_u64 * 4),
        ("gk_table_len", c_i64), ("gk_factor", c_f64),
    ]


class StepState(ctypes.Structure):  # == sdm_step_state
    _fields_ = [
        ("idx", c_ptr), ("tmp_idx", c_ptr), ("multiplicity", c_ptr), ("attributes", c_ptr),
        ("cell_id", c_ptr), ("cell_idx", c_ptr), ("cell_start", c_ptr), ("dt_left", c_ptr),
        ("stats_dt_min", c_ptr), ("stats_n_substep", c_ptr), ("collision_rate", c_ptr),
        ("collision_rate_deficit", c_ptr), ("coalescence_rate", c_ptr), ("breakup_rate", c_ptr),
        ("breakup_rate_deficit", c_ptr), ("gk_a", c_ptr), ("gk_b", c_ptr), ("ctl", c_ptr),
        ("nm", c_ptr), ("known_valid", c_i64), ("rng_offset", c_u64),
        ("rng_offset_breakup", c_u64),
    ]


class StepResult(ctypes.Structure):  # == sdm_step_result
    _fields_ = [
        ("n_substeps", c_i64), ("n_pairs", c_i64), ("valid_n_sd", c_i64), ("idx_swapped", c_i64),
        ("rng_offset", c_u64), ("rng_offset_breakup", c_u64), ("ctl", c_i64 * 8),
    ]


class DispCfg(ctypes.Structure):  # == sdm_disp_cfg
    _fields_ = [
        ("n_sd", c_i64), ("n_dims", ctypes.c_int32), ("scheme", ctypes.c_int32),
        ("enable_sedimentation", ctypes.c_int32), ("n_substeps", ctypes.c_int32),
        ("grid", c_i64 * 3), ("strides", c_i64 * 3), ("dt_over_dz", c_f64), ("level", c_f64),
    ]


class DispState(ctypes.Structure):  # == sdm_disp_state
    _fields_ = [
        ("courant", c_ptr * 3), ("displacement", c_ptr), ("position_in_cell", c_ptr),
        ("cell_origin", c_ptr), ("cell_id", c_ptr), ("fall_velocity", c_ptr),
        ("water_mass", c_ptr), ("multiplicity", c_ptr), ("idx", c_ptr), ("ctl", c_ptr),
    ]


def declared_symbols():
    """every function the header declares"""
    with open(HEADER_PATH, encoding="utf-8") as header:
        text = header.read()
    return sorted(set(re.findall(r"^(?:int|const char \*)\s*(sdm_[a-z0-9_]+)\(", text, re.M)))


_lib = None


def load():
    global _lib  # pylint: disable=global-statement
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (pysdm_amd has no CPU fallback)"
        )
    lib = ctypes.CDLL(LIB_PATH)
    missing = [name for name in declared_symbols() if not hasattr(lib, name)]
    if missing:
        raise ImportError(f"{LIB_PATH} lacks symbols declared in sdm_hip.h: {missing}")
    lib.sdm_last_error.restype = ctypes.c_char_p
    for name in declared_symbols():
        if name != "sdm_last_error":
            getattr(lib, name).restype = c_int
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise RuntimeError(f"libsdm_hip: error {code}: {load().sdm_last_error().decode()}")
