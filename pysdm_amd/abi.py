"""Binding of include/sdm_hip.h generated from the header itself.

The header is the single description of the boundary; this module parses its prototypes once and
builds, for any shared library that implements them, a table of callables that
  * accept device arrays (torch tensors), host arrays (numpy) or ctypes objects for pointer
    parameters and Python numbers / sequences for scalar and fixed-size host-array parameters,
  * check on the host, before anything is launched, that every array is contiguous and of the
    element type the C parameter names (double <-> float64, int64_t <-> int64, uint8_t <-> bool /
    uint8) - a kernel must never see an operand it was not written for,
  * turn a non-zero return code into a RuntimeError carrying `sdm_last_error()`.
The product library is pysdm_amd/libsdm_hip.so (HIP kernels, device pointers).  The CPU oracle
(oracle/, test infrastructure) implements the same header for host pointers and is bound by the
same code.  There is no fallback between the two: a missing library or symbol is an ImportError.
"""
import ctypes
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sdm_hip.h")
# SDM_HIP_LIB: another build of the same library (tuning variants); still no fallback
HIP_LIB_PATH = os.environ.get("SDM_HIP_LIB") or os.path.join(_HERE, "libsdm_hip.so")

c_i64, c_f64, c_int, c_ptr, c_u64 = (ctypes.c_int64, ctypes.c_double, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_uint64)


# ---- the structs of the fused entry points (layout checked against the header in tests/test_abi.py)
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
        ("cell_owned", c_ptr), ("exchange", c_ptr), ("exchange_user", c_ptr),
        ("xchg_cells", c_ptr), ("xchg_idx", c_ptr),
        ("shard_rank", ctypes.c_int32), ("shard_world", ctypes.c_int32),
        ("cell_id_by_id", c_ptr),
    ]


# sdm_exchange_fn
ExchangeFn = ctypes.CFUNCTYPE(c_int, c_ptr, c_int, c_ptr, c_i64)
XCHG_SUM_F64, XCHG_SUM_I64, XCHG_MIN_F64 = 1, 2, 3
COMM_ID_BYTES = 128


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


class DispShard(ctypes.Structure):  # == sdm_disp_shard
    _fields_ = [
        ("cell_owned", c_ptr), ("n_cell", c_i64), ("exchange", c_ptr), ("exchange_user", c_ptr),
        ("shard_rank", ctypes.c_int32), ("shard_world", ctypes.c_int32),
        ("xchg_counts", c_ptr), ("xchg_words", c_ptr), ("word_capacity", c_i64),
        ("cell_id_by_id", c_ptr), ("role", c_ptr), ("role_ready", ctypes.c_int32),
        ("multiplicity", c_ptr), ("attributes", c_ptr), ("n_attr", ctypes.c_int32),
        ("n_moved", c_i64), ("n_left", c_i64), ("n_arrived", c_i64), ("n_words", c_i64),
        ("n_removed", c_i64),
    ]


# ---- header parsing -------------------------------------------------------------------------
_SCALARS = {"int": c_int, "int64_t": c_i64, "uint64_t": c_u64, "double": c_f64,
            "int32_t": ctypes.c_int32}
_ELEMENT = {"double": (np.float64,), "int64_t": (np.int64,), "uint8_t": (np.uint8, np.bool_),
            "uint64_t": (np.uint64,), "int32_t": (np.int32,)}


class Param:  # pylint: disable=too-few-public-methods
    """one C parameter: kind in {ctx, scalar, pointer, host_array}"""

    def __init__(self, text):
        text = " ".join(text.split())
        match = re.match(r"^(.*?)(\w+)(\[\d*\])?$", text)
        self.name = match.group(2)
        ctype = match.group(1).strip()
        self.array_len = None
        if match.group(3):
            digits = match.group(3)[1:-1]
            self.array_len = int(digits) if digits else -1
        self.base = ctype.replace("const", "").replace("*", "").strip()
        stars = ctype.count("*")
        if self.base == "sdm_ctx":
            self.kind = "ctx" if stars == 1 else "pointer"
        elif self.array_len is not None:
            self.kind = "host_array"
        elif stars:
            self.kind = "pointer"
        else:
            self.kind = "scalar"

    def convert(self, value, symbol):
        if self.kind == "scalar":
            return _SCALARS[self.base](value)
        if self.kind == "host_array":
            return _host_array(value, self, symbol)
        return _pointer(value, self, symbol)


def _host_array(value, param, symbol):
    if isinstance(value, ctypes.Array):
        return value
    values = [v for v in value]
    if param.array_len not in (None, -1) and len(values) != param.array_len:
        raise ValueError(f"{symbol}: `{param.name}` takes {param.array_len} values, "
                         f"got {len(values)}")
    return (_SCALARS[param.base] * len(values))(*values)


def _pointer(value, param, symbol):
    if value is None:
        return c_ptr(0)
    if isinstance(value, (ctypes._SimpleCData, ctypes.Structure, ctypes.Array)):  # pylint: disable=protected-access
        return ctypes.byref(value)
    if isinstance(value, type(ctypes.byref(c_int()))) or isinstance(value, c_ptr):
        return value
    if isinstance(value, int):
        return c_ptr(value)
    if isinstance(value, (list, tuple)):  # a small host array (e.g. coefficients)
        return (_SCALARS[param.base] * len(value))(*value)
    allowed = _ELEMENT.get(param.base)
    if isinstance(value, np.ndarray):
        if not value.flags["C_CONTIGUOUS"]:
            raise ValueError(f"{symbol}: `{param.name}` is not contiguous")
        if allowed and value.dtype.type not in allowed:
            raise TypeError(f"{symbol}: `{param.name}` is {param.base}*, got {value.dtype}")
        return c_ptr(value.ctypes.data)
    if hasattr(value, "data_ptr"):  # torch tensor
        if not value.is_contiguous():
            raise ValueError(f"{symbol}: `{param.name}` is not contiguous")
        if allowed:
            names = {np.dtype(t).name for t in allowed} | ({"bool"} if np.bool_ in allowed
                                                          else set())
            if str(value.dtype).rsplit(".", maxsplit=1)[-1] not in names:
                raise TypeError(f"{symbol}: `{param.name}` is {param.base}*, got {value.dtype}")
        return c_ptr(value.data_ptr())
    raise TypeError(f"{symbol}: cannot pass {type(value).__name__} as `{param.name}`")


def parse_header(path=HEADER_PATH):
    """{symbol: (return kind, [Param, ...])} for every function the header declares"""
    with open(path, encoding="utf-8") as header:
        text = header.read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    table = {}
    for ret, name, params in re.findall(
            r"\b(int|const char \*)\s*(sdm_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        params = params.strip()
        plist = [] if params in ("", "void") else [Param(p) for p in params.split(",")]
        table[name] = ("str" if "char" in ret else "int", plist)
    return table


def declared_symbols():
    return sorted(parse_header())


class Library:
    """a shared library implementing include/sdm_hip.h, with checked, converting callables"""

    def __init__(self, path, what):
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing ({what}); build it with "
                              f"`python -c 'import __graft_entry__ as g; g.build()'`")
        self.path = path
        self.cdll = ctypes.CDLL(path)
        self.signatures = parse_header()
        missing = [name for name in self.signatures if not hasattr(self.cdll, name)]
        if missing:
            raise ImportError(f"{path} lacks symbols declared in sdm_hip.h: {missing}")
        for name, (ret, _) in self.signatures.items():
            getattr(self.cdll, name).restype = ctypes.c_char_p if ret == "str" else c_int

    def last_error(self):
        return self.cdll.sdm_last_error().decode()

    def check(self, code):
        if code != 0:
            raise RuntimeError(f"{os.path.basename(self.path)}: error {code}: "
                               f"{self.last_error()}")

    def invoke(self, symbol, ctx_handle, args):
        """calls `symbol(ctx, *args)`; array / dtype checks first, return code checked after"""
        _, params = self.signatures[symbol]
        expected = [p for p in params if p.kind != "ctx"]
        if len(args) != len(expected):
            raise TypeError(f"{symbol} takes {len(expected)} arguments "
                            f"({', '.join(p.name for p in expected)}), got {len(args)}")
        converted, it = [], iter(args)
        for param in params:
            converted.append(ctx_handle if param.kind == "ctx"
                             else param.convert(next(it), symbol))
        self.check(getattr(self.cdll, symbol)(*converted))


_hip_library = None


def hip_library():
    """libsdm_hip.so (raises ImportError if it is not built: there is no CPU fallback)"""
    global _hip_library  # pylint: disable=global-statement
    if _hip_library is None:
        _hip_library = Library(HIP_LIB_PATH, "the HIP kernels of pysdm_amd")
    return _hip_library


def pcg64_state_inc(seed):
    """{state_hi, state_lo, inc_hi, inc_lo} of numpy.random.PCG64(seed): NumPy defines the stream
    the reference draws from (PySDM/backends/impl_numba/random.py:16); both libraries reproduce
    it from these four words with jump-ahead"""
    state = np.random.PCG64(seed).state["state"]
    mask = (1 << 64) - 1
    return (state["state"] >> 64, state["state"] & mask, state["inc"] >> 64, state["inc"] & mask)
